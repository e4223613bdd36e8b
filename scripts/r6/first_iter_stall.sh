#!/bin/bash
# hunts the intermittent ~12 ms stall in the first iteration of cfg4 from x0 (seen in 2 of ~40 runs): repeats bench.py's cfg4 leg
# with the host trace on and keeps the trace of a run whose first iteration took more than 32 ms
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-25}); do
  rm -f /tmp/ht.txt
  r=$(LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/r6/cfg4_leg.py --m 10 --iters 40 2>/dev/null | tail -1)
  ms=$(echo "$r" | python -c "import sys,json; print(json.loads(sys.stdin.read())['first_iteration_ms'])")
  echo "run $i first_iteration_ms $ms"
  if python -c "import sys; sys.exit(0 if float('$ms') > 32 else 1)"; then cp /tmp/ht.txt gpurun_out/stall_trace_$i.txt; echo "  kept gpurun_out/stall_trace_$i.txt"; fi
done
cp /tmp/ht.txt gpurun_out/stall_trace_normal.txt
