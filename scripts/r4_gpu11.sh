#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/g11_kernel_stats_lookahead.txt 2>&1
python scripts/timeline.py $db 3 > gpurun_out/g11_timeline.txt 2>&1
tail -45 gpurun_out/g11_timeline.txt
