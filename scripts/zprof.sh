#!/bin/bash
# Development aid (GPU box): per-kernel statistics + per-launch listing of the sweeps of the complex16 configuration for one or more builds of the library.
# usage: bash scripts/zprof.sh <tag> [lib paths...]   ("" = in-tree)
tag=${1:-zprof}; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for lib in "" "$@"; do
  cd /tmp; rm -rf /tmp/zp$i
  [ -n "$lib" ] && lib=$R/$lib
  SLUAMD_LIB=$lib rocprofv3 --kernel-trace -d /tmp/zp$i -o run -- python $R/bench.py --workload zgrid2d --n 1000 --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/zp$i.json 2> /tmp/zp$i.err
  cd $R
  db=$(find /tmp/zp$i -name "*.db" | head -1)
  [ -z "$db" ] && { echo "no db"; tail -5 /tmp/zp$i.err; ls -R /tmp/zp$i | head; }
  python scripts/rocpd_stats.py $db > gpurun_out/${tag}_stats$i.txt
  for k in kz_fwd_fused kz_bwd_fused kz_fwd_update kz_bwd_update kz_solve_diag_wave; do python scripts/rocpd_stats.py $db --launches $k > gpurun_out/${tag}_launches${i}_$k.txt; done
  echo "== lib $i ($lib)"; grep "kz_.*fused\|kz_.*update\|kz_solve" gpurun_out/${tag}_stats$i.txt
  i=$((i+1))
done
