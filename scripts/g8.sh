# round 5, GPU call 8: HBM traffic of the Schur kernel per DAG level (serial schedule, two PMC passes + the library's launch table)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcl_$c
  SLUAMD_NO_LOOKAHEAD=1 SLUAMD_PROFILE_DUMP=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcl_$c -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/pmcl_$c.json 2> /tmp/pmcl_$c.err
  grep "^SCHUR level" /tmp/pmcl_$c.err > $R/gpurun_out/g8_dump_$c.txt
done
cd $R
python scripts/level_flops.py 100 gpurun_out/g8_level_flops.json > /dev/null 2>&1
f=$(find /tmp/pmcl_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmcl_WRITE_SIZE -name "*.db" | head -1)
python scripts/pmc_by_level.py $f $w gpurun_out/g8_dump_FETCH_SIZE.txt gpurun_out/g8_level_flops.json > gpurun_out/g8_pmc_by_level.txt 2> gpurun_out/g8_pmc_by_level.err
tail -12 gpurun_out/g8_pmc_by_level.txt; tail -3 gpurun_out/g8_pmc_by_level.err
