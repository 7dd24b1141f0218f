#!/bin/bash
mkdir -p gpurun_out
(SLUAMD_FUSE_GROUP_MIN_NODES=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g24_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g24_pytest.log)
tail -2 gpurun_out/g24_pytest.log
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py tests/test_gpu_grid.py -q -x --timeout=600 > gpurun_out/g24_pytest2.log 2>&1; echo "pytest rc $?" >> gpurun_out/g24_pytest2.log)
tail -2 gpurun_out/g24_pytest2.log
bash scripts/ab.sh g24 $PWD/ab/libsluamd_rec32.so
