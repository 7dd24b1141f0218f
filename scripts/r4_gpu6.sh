#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_grid.py tests/test_gpu_refine.py tests/test_gpu_bench_config.py tests/test_gpu_fuzz.py -q -x --timeout=900 > gpurun_out/g6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g6_pytest.log)
tail -3 gpurun_out/g6_pytest.log
bash scripts/ab.sh g6 $PWD/ab/libsluamd_unpacked.so
