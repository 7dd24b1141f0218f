#!/bin/bash
# round 4, GPU call 3: leaf-level options (substitution panel solves + deferred inverses; 32 x 32 one-wave Schur tiles): parity, A/B on one box
mkdir -p gpurun_out
(SLUAMD_TINY_TILES=1 SLUAMD_TRSM_LEAF=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout=600 > gpurun_out/g3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g3_pytest.log)
tail -3 gpurun_out/g3_pytest.log
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g3_$name.json 2> gpurun_out/g3_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/g3_$name.json"))
    print("%-60s factor_ms %.1f solve_ms %.2f schur_ms %.1f panel_ms %.1f frac %.3f res %.1e" % ("$*", j["factor_ms"], j["solve_ms"], j["roofline"]["schur_ms"], j["roofline"]["panel_ms"], j["roofline"]["frac"], j["residual"]))
except Exception as e:
    print("$* failed", e); print(open("gpurun_out/g3_$name.err").read()[-400:])
PY
}
run base A=1
run leaf SLUAMD_TRSM_LEAF=1
run tiny SLUAMD_TINY_TILES=1
run both SLUAMD_TRSM_LEAF=1 SLUAMD_TINY_TILES=1
run base2 A=1
run both2 SLUAMD_TRSM_LEAF=1 SLUAMD_TINY_TILES=1
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
SLUAMD_TRSM_LEAF=1 SLUAMD_TINY_TILES=1 rocprofv3 --kernel-trace --stats -d /tmp/ks -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > /tmp/ks.json 2> /tmp/ks.err
cd $R
db=$(find /tmp/ks -name "*.db" | head -1)
python scripts/rocpd_stats.py $db > gpurun_out/g3_kernel_stats_both.txt 2>&1
python scripts/timeline.py $db 3 > gpurun_out/g3_timeline_both.txt 2>&1
head -30 gpurun_out/g3_kernel_stats_both.txt
