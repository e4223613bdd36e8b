#!/bin/bash
# interleaved same-box A/B of one environment knob on cfg4 (a box slows by a few per cent as it warms up: sequential blocks of
# runs are biased towards whatever ran first): ROUNDS x (A, B) with A = "$1" and B = "$2" as VAR=value strings (or "-" for none)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=${M:-10}; IT=40; [ $M -gt 16 ] && IT=60
run () {
  env $1 python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters $IT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('model') or {}
print('%-28s window mean %.3f ms (%.1f it/s)  median-based %.1f  from x0 %.1f  fx %.12g' % ('$1', m.get('window_ms_per_iteration',0), 1e3/max(m.get('window_ms_per_iteration',1e9),1e-9), d['steady_it_per_s'], d['it_per_s'], d['fx']))"
}
A=$1; B=$2; [ "$A" = "-" ] && A="LBFGSX_NOP=1"; [ "$B" = "-" ] && B="LBFGSX_NOP=1"
run "$A" > /dev/null   # warm the box
for r in $(seq 1 ${ROUNDS:-4}); do run "$A"; run "$B"; done
