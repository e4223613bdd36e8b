#!/bin/bash
# interleaved A/B on one box: the L u U list pass also when one of the two sets is empty (LBFGSX_LU_ONE_SIDED, SubspaceMin.h),
# bench.py's cfg4 leg at m = 10 and m = 20 (steady it/s, from x0, first iteration ms)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for m in 20 10; do
    it=40; [ $m = 20 ] && it=60
    for v in 1 0; do
      echo -n "LBFGSX_LU_ONE_SIDED=$v m=$m  "
      LBFGSX_LU_ONE_SIDED=$v python scripts/r6/cfg4_leg.py --m $m --iters $it 2>/dev/null | tail -1
    done
  done
done
