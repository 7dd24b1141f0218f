#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_bench_config.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g12_pytest.log)
tail -3 gpurun_out/g12_pytest.log
bash scripts/env_ab.sh g12 SLUAMD_KSPLIT=1 SLUAMD_KSPLIT=2 SLUAMD_KSPLIT=8 SLUAMD_KSPLIT=1
