mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
SLUAMD_SOLVE_GROUPS=1 rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/g24_kernel_stats.txt 2>&1
python scripts/solve_timeline.py $db 2 > gpurun_out/g24_solve_timeline.txt 2>&1
grep -E "gemm_batched|grp_gather|sweep_join|kernel " gpurun_out/g24_kernel_stats.txt
head -14 gpurun_out/g24_solve_timeline.txt
