#!/bin/bash
# the sweeps' solves at 2c = 40 (m = 20): a row over two lanes of 20 columns (tree) against four lanes of 10 as the other passes keep
# (variants/liblbfgsx_x104.so = the tree before the change), interleaved on one box: bench.py's cfg4 leg at m = 20
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3 4; do
for v in base x104; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=20  "; python scripts/r6/cfg4_leg.py --m 20 --iters 60 2>/dev/null | tail -1
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
