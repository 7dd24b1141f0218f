#!/bin/bash
# Regenerates the PMC evidence bench.py's roofline.traffic is built from (run on the GPU box from the repo root):
#   two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + one MFMA-busy pass of the default bench workload,
#   kernel-trace only (no sys/hip/hsa traces), summaries into gpurun_out/ -> copy to profiles/.
set -u
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d /tmp/pmc_mfma -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/pmc_mfma.log 2>&1
cd $R
f=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1); m=$(find /tmp/pmc_mfma -name "*.db" | head -1)
python scripts/make_pmc_json.py $f $w 4 "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (one pass each) -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-scaling-point (4 factorisations: 2 warm-up, 1 timed, 1 profiled), scripts/collect_pmc.sh" > gpurun_out/${tag}_pmc_schur.json
python scripts/rocpd_pmc.py $f $w > gpurun_out/${tag}_bench100_pmc_traffic.txt
python scripts/rocpd_pmc.py $m > gpurun_out/${tag}_bench100_pmc_mfma.txt
cat gpurun_out/${tag}_pmc_schur.json | head -12
