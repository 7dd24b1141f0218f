# round 5, GPU call 4: kernel statistics + timeline of the complex16 workload, one-wave vs four-wave diagonal LU
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for z in 0 100000; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ksz
  SLUAMD_ZLU4_MAX_NODES=$z rocprofv3 --kernel-trace --stats -d /tmp/ksz -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 2 --no-cpu-baseline > /tmp/ksz.json 2> /tmp/ksz.err
  cd $R
  db=$(find /tmp/ksz -name "*.db" | head -1)
  python scripts/rocpd_stats.py $db > gpurun_out/g4_zstats_$z.txt 2>&1
  python scripts/timeline.py $db 3 > gpurun_out/g4_ztimeline_$z.txt 2>&1
  head -12 gpurun_out/g4_zstats_$z.txt
done
