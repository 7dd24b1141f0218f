#!/bin/bash
mkdir -p gpurun_out
(SLUAMD_FUSE_SMALL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g18_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g18_pytest.log)
tail -3 gpurun_out/g18_pytest.log
bash scripts/env_ab.sh g18 "SLUAMD_FUSE_SMALL=1" "SLUAMD_FUSE_SMALL=1 SLUAMD_FUSE_MIN_PCT=60"
