#!/bin/bash
# Development aid (GPU box): kernel timeline of ONE pzgstrf3d of the complex16 configuration -> gpurun_out/<tag>_ztimeline.txt
tag=${1:-z}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kzt
env "$@" rocprofv3 --kernel-trace -d /tmp/kzt -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 2 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/kzt.json 2> /tmp/kzt.err
cd $R
db=$(find /tmp/kzt -name "*.db" | head -1)
python scripts/timeline.py $db 2 > gpurun_out/${tag}_ztimeline.txt 2>&1
tail -30 gpurun_out/${tag}_ztimeline.txt
