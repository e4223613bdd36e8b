#!/bin/bash
# cfg4's problem (n = 1e7) at several history lengths: it/s from x0 and steady, the roofline view of the steady state and
# (PROF=1) the kernels that take the time.  Output: gpurun_out/r4/by_m_<tag>.txt
TAG=${1:-head}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/by_m_$TAG.txt
: > $OUT
for m in ${MS:-10 12 16 20}; do
  it=40; [ $m -gt 16 ] && it=60
  for rep in 1 2; do
    env "$@" python scripts/bench_lbfgsb.py --n 1e7 --m $m --iters $it 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stats']
print('m=%2d iters=%d: from x0 %7.1f it/s  steady %7.1f it/s  frac %.3f  q %.2f  sweeps %d  carried %s  fx %.17g' % (d['m'], d['niter'], d['it_per_s'], d['steady_it_per_s'], d['roofline']['frac'], d['q_sweeps_per_iteration'], st['submin_sweeps'], st.get('gram_carried'), d['fx']))" >> $OUT
  done
  if [ "${PROF:-0}" = 1 ]; then
    rm -rf /tmp/pg; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m $m --iters $it > /tmp/pg.json 2>/dev/null
    python3 - >> $OUT <<PY
import csv
rows = list(csv.DictReader(open("/tmp/pg/b_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("  kernels of the profiled run (m=$m), %.1f ms in all:" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("    %5.1f %%  calls %5s  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
  fi
done
cat $OUT
