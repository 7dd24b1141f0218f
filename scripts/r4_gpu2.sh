#!/bin/bash
# round 4, GPU call 2: full GPU suite on the XY changes, one-GPU proxy of the grid work inflation (xy_bench), default bench
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/g2_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g2_pytest.log)
tail -4 gpurun_out/g2_pytest.log
for cfg in "100 1 1 1" "100 2 2 1" "100 2 2 2"; do
  timeout 400 python scripts/xy_bench.py $cfg 2 > gpurun_out/g2_xy_$(echo $cfg | tr ' ' '_').txt 2>&1; tail -12 gpurun_out/g2_xy_$(echo $cfg | tr ' ' '_').txt
done
SLUAMD_NO_FUSE=1 timeout 400 python scripts/xy_bench.py 100 2 2 2 2 > gpurun_out/g2_xy_100_2_2_2_nofuse.txt 2>&1; tail -3 gpurun_out/g2_xy_100_2_2_2_nofuse.txt
SLUAMD_NO_LEVEL_SPLIT=1 timeout 400 python scripts/xy_bench.py 100 2 2 2 2 > gpurun_out/g2_xy_100_2_2_2_nosplit.txt 2>&1; tail -3 gpurun_out/g2_xy_100_2_2_2_nosplit.txt
timeout 400 python scripts/grid_footprint_check.py 100 2 2 2 > gpurun_out/g2_footprint_100.txt 2>&1; tail -10 gpurun_out/g2_footprint_100.txt
timeout 600 python scripts/xy_bench.py 150 2 2 2 1 > gpurun_out/g2_xy_150_2_2_2.txt 2>&1; tail -12 gpurun_out/g2_xy_150_2_2_2.txt
