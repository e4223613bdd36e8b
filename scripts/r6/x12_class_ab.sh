#!/bin/bash
# 2c = 9..12 (m = 5, 6 -- the reference's default m): a class that fits exactly, two lanes of 6 columns (tree), against the 16-slot class
# (variants/liblbfgsx_pre.so), interleaved on one box: bench.py's cfg4 leg at m = 6 and m = 5
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for m in 6 5; do
for v in base pre; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=$m  "; python scripts/r6/cfg4_leg.py --m $m --iters 40 2>/dev/null | tail -1
done; done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
python -m pytest tests/test_lbfgsb_gpu.py tests/test_param_space_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
