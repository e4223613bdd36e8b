#!/bin/bash
# 2c = 41..48 (m = 21..24): classes that fit exactly -- four lanes of 12 columns, the solves two lanes of 24 (tree) -- against the
# 60-slot classes with a fifth of their slots padded (variants/liblbfgsx_pre.so), interleaved on one box: bench.py's cfg4 leg at m = 24
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base pre; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=24  "; python scripts/r6/cfg4_leg.py --m 24 --iters 60 2>/dev/null | tail -1
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
python -m pytest tests/test_lbfgsb_gpu.py -x -q -m gpu -k "long_histories" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
