mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/g20_kernel_stats.txt 2>&1
python scripts/timeline.py $db 3 > gpurun_out/g20_timeline.txt 2>&1
grep -A70 "^# span" gpurun_out/g20_timeline.txt | head -80
