#!/bin/bash
# Development aid (CPU): does tests/test_stream_order.py catch a forgotten dependency?  Deletes one hipStreamWaitEvent of the
# look-ahead schedule at a time from sluamd_factor.cpp, rebuilds the emulation library and expects the test file to FAIL.
# The sources are restored on exit.  (All mutations are caught; the unmodified code passes every seed.)
set -u
cd "$(dirname "$0")/.."
src=superlu_dist_amd/csrc/sluamd_factor.cpp
cp $src /tmp/sluamd_factor_orig.cpp
trap 'cp /tmp/sluamd_factor_orig.cpp $src; make -C oracle >/dev/null 2>&1' EXIT
muts=(
  'hipStreamWaitEvent(ps, e_u1, 0);'
  'hipStreamWaitEvent(us, e_p, 0); hipStreamWaitEvent(u2s, e_p, 0); hipStreamWaitEvent(s, e_p, 0);'
  'if (e_bulk_prev2) hipStreamWaitEvent(ps, e_bulk_prev2, 0);'
  'if (e_u2_prev) hipStreamWaitEvent(ps, e_u2_prev, 0);'
  'if (xy && e_bulk_prev) hipStreamWaitEvent(ps, e_bulk_prev, 0);'
  'if (e) hipStreamWaitEvent(ps, e, 0);'
  # split panel solves (round 5): the other strips wait for the level's diagonal blocks; the part-1 tiles for the urgent strips; U2 / the bulk for both parts
  'hipStreamWaitEvent(us, e_pa, 0);'
  'hipStreamWaitEvent(us, e_split_urgent, 0);'
  'hipStreamWaitEvent(u2s, e_split_urgent, 0); hipStreamWaitEvent(u2s, e_rest, 0);'
  'hipStreamWaitEvent(s, e_split_urgent, 0); hipStreamWaitEvent(s, e_rest, 0);'
  # merged chain groups (round 5, SLUAMD_SOLVE_GROUPS=1): the group stream waits for the members' panels / inverses; the factorisation for the group stream
  'hipEventRecord(e_g, s); hipStreamWaitEvent(H->gstream, e_g, 0);'
  'if (!H->lvl_groups.empty() && H->gstream) wait_on(s, H->gstream);'
)
# ONLY_NEW=1: just the e_u1 wait (both branches) and the split-panel waits
if [ -n "${ONLY_NEW:-}" ]; then muts=("${muts[@]:0:1}" "${muts[@]:6}"); fi
bad=0
for m in "${muts[@]}"; do
  cp /tmp/sluamd_factor_orig.cpp $src
  python - "$m" <<'PY'
import sys
p = 'superlu_dist_amd/csrc/sluamd_factor.cpp'
s = open(p).read(); m = sys.argv[1]
assert s.count(m) >= 1, (m, s.count(m))      # (a wait that exists in the split and the unsplit branch of the schedule goes in both)
open(p, 'w').write(s.replace(m, '/* mutated */'))
PY
  make -C oracle >/dev/null 2>&1
  out=$(timeout 900 python -m pytest tests/test_stream_order.py -q -x --timeout=120 -k 'not under_the_scheduler' 2>&1 | grep -E ' (passed|failed)' | tail -1)   # a hang is a failure (per-test timeout), the first failure is enough
  echo "without '$m': $out"
  case "$out" in *failed*) ;; *) bad=1; echo "  NOT CAUGHT";; esac
done
exit $bad
