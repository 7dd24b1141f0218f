#!/bin/bash
# Development aid (CPU): does tests/test_stream_order.py catch a forgotten dependency?  Deletes one hipStreamWaitEvent of the
# look-ahead schedule at a time from sluamd_factor.cpp, rebuilds the emulation library and expects the test file to FAIL.
# The source is restored on exit.  (Round 2: all five mutations are caught; the unmodified schedule passes every seed.)
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
)
bad=0
for m in "${muts[@]}"; do
  cp /tmp/sluamd_factor_orig.cpp $src
  python - "$m" <<'PY'
import sys
p = 'superlu_dist_amd/csrc/sluamd_factor.cpp'
s = open(p).read(); m = sys.argv[1]
assert s.count(m) == 1, (m, s.count(m))
open(p, 'w').write(s.replace(m, '/* mutated */'))
PY
  make -C oracle >/dev/null 2>&1
  out=$(timeout 900 python -m pytest tests/test_stream_order.py -q -k 'not under_the_scheduler' 2>&1 | grep -E ' (passed|failed)' | tail -1)
  echo "without '$m': $out"
  case "$out" in *failed*) ;; *) bad=1; echo "  NOT CAUGHT";; esac
done
exit $bad
