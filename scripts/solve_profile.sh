#!/bin/bash
# Development aid (GPU box): kernel timeline of ONE pdgstrs3d of the default bench workload -> gpurun_out/<tag>_solve_timeline.txt
tag=${1:-solve}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kss
env "$@" rocprofv3 --kernel-trace -d /tmp/kss -o run -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/kss.json 2> /tmp/kss.err
cd $R
db=$(find /tmp/kss -name "*.db" | head -1)
python scripts/solve_timeline.py $db 2 > gpurun_out/${tag}_solve_timeline.txt 2>&1
head -12 gpurun_out/${tag}_solve_timeline.txt
