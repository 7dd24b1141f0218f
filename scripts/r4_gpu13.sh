#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/timeline.py $db 3 > gpurun_out/g13_timeline.txt 2>&1
awk '$2>285.0 && $2<287.5' gpurun_out/g13_timeline.txt | head -50
bash scripts/env_ab.sh g13 SLUAMD_TRSM_RS32=1
