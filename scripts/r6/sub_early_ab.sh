#!/bin/bash
# interleaved A/B on one box: the first solve of every subspace minimisation in its sweep form (LBFGSX_SUB_EARLY=1) against the
# form chosen by what the previous iteration needed (0); bench.py's cfg4 leg at m = 20 and m = 10
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in 20 10; do
    it=40; [ $m = 20 ] && it=60
    for v in 1 0; do
      echo -n "LBFGSX_SUB_EARLY=$v m=$m  "
      LBFGSX_SUB_EARLY=$v python scripts/r6/cfg4_leg.py --m $m --iters $it 2>/dev/null | tail -1
    done
  done
done
