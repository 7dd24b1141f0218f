#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py -q --timeout=300 -x > gpurun_out/tight_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/tight_pytest.log)
grep -E "passed|failed|FAILED|rc " gpurun_out/tight_pytest.log | tail -8
bash scripts/ab.sh tight ab/libsluamd_notight.so 2>&1 | tee gpurun_out/tight_ab.txt
