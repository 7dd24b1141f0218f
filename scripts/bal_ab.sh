#!/bin/bash
# round 6: balanced bulk launches (LevelSched::x_off) -- parity subset, then a same-box A/B of the planner switch
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q --timeout=300 -x > gpurun_out/bal_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/bal_pytest.log)
grep -E "passed|failed|FAILED|rc " gpurun_out/bal_pytest.log | tail -8
bash scripts/env_ab.sh bal "SLUAMD_BALANCE_MIN_TILES=0" "SLUAMD_BALANCE_OVH=8" "SLUAMD_BALANCE_OVH=1" "SLUAMD_BALANCE_MIN_TILES=256" "SLUAMD_BALANCE_MIN_TILES=0" 2>&1 | tee gpurun_out/bal_ab.txt
