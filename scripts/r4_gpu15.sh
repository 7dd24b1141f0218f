#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_grid.py tests/test_gpu_fuzz.py tests/test_gpu_refine.py -q -x --timeout=900 > gpurun_out/g15_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g15_pytest.log)
tail -3 gpurun_out/g15_pytest.log
bash scripts/env_ab.sh g15 SLUAMD_NO_FULL_INV64=1
