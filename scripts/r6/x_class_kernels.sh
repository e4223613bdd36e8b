#!/bin/bash
# per-kernel averages of cfg4 (m = $1) with the tree's library and with variants/liblbfgsx_$2.so (rocprofv3 --kernel-trace --stats), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=${1:-10}; V=${2:-x201}; IT=40; [ $M = 20 ] && IT=60
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for v in base $V; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  rm -rf /tmp/px_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$v -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters $IT > /dev/null 2>&1
  echo "== $v"; f=$(find /tmp/px_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-70s calls %4s avg %8.1f us  %5s %%" % (r["Name"][:70].replace("void lbfgsx::",""), r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
