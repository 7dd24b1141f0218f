mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ksz
rocprofv3 --kernel-trace --stats -d /tmp/ksz -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/ksz.json 2> /tmp/ksz.err
cd $R
python scripts/solve_timeline.py $(find /tmp/ksz -name "*.db" | head -1) 2 > gpurun_out/g15_zsolve_timeline.txt 2>&1
head -12 gpurun_out/g15_zsolve_timeline.txt
