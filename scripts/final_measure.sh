#!/bin/bash
# End-of-round measurement on the GPU box (from the repo root): PMC traffic of the Schur kernels, rocprof kernel statistics of the
# default bench command (look-ahead) and of the serial profile, the default bench line.  Outputs -> gpurun_out/ (copy to profiles/).
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
bash scripts/collect_pmc.sh $tag > gpurun_out/${tag}_collect_pmc.log 2>&1
cp gpurun_out/${tag}_pmc_schur.json profiles/r06_pmc_schur.json    # bench.py reads this one (same box, same kernels)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/${tag}_kernel_stats_lookahead_final.txt 2>&1
python scripts/timeline.py $db 3 > gpurun_out/${tag}_timeline_final.txt 2>&1
python scripts/solve_timeline.py $db 2 > gpurun_out/${tag}_solve_timeline_final.txt 2>&1
cd /tmp && rm -rf /tmp/ks2
SLUAMD_NO_LOOKAHEAD=1 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks2.json 2> /tmp/ks2.err
cd $R
db2=$(find /tmp/ks2 -name "*.db" | head -1)
python scripts/rocpd_stats.py $db2 > gpurun_out/${tag}_kernel_stats_serial_final.txt 2>&1
python bench.py > gpurun_out/${tag}_bench100_final.json 2> gpurun_out/${tag}_bench100_final.err
tail -c 600 gpurun_out/${tag}_bench100_final.json
# complex16 workload (BASELINE.json configs[4]): bench line + kernel statistics
cd /tmp && rm -rf /tmp/ksz
rocprofv3 --kernel-trace --stats -d /tmp/ksz -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/ksz.json 2> /tmp/ksz.err
cd $R
python scripts/rocpd_stats.py $(find /tmp/ksz -name "*.db" | head -1) > gpurun_out/${tag}_zgrid2d_kernel_stats_final.txt 2>&1
python bench.py --workload zgrid2d --n 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_zgrid2d_1000.json 2> gpurun_out/${tag}_zgrid2d_1000.err
