#!/bin/bash
# round 6: the predicate-free loader for clean sources (k_schur `clean`) -- parity subset, then a same-box A/B against the build without it
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_grid.py -q --timeout=300 -x > gpurun_out/clean_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/clean_pytest.log)
grep -E "passed|failed|FAILED|rc " gpurun_out/clean_pytest.log | tail -8
bash scripts/ab.sh clean ab/libsluamd_noclean.so 2>&1 | tee gpurun_out/clean_ab.txt
