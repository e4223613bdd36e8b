#!/bin/bash
# the sweeps' solves at 2c = 40 compiled for two waves per SIMD (256 registers, 16-26 spilled: tree) against one wave per SIMD
# (284 registers, no scratch: variants/liblbfgsx_pre.so), interleaved on one box: bench.py's cfg4 leg at m = 20
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base pre; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=20  "; python scripts/r6/cfg4_leg.py --m 20 --iters 60 2>/dev/null | tail -1
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
