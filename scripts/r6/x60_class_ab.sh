#!/bin/bash
# the sweeps' solves at 2c = 41..60 (m = 24 and 30): two lanes of 30 columns (tree) against four lanes of 15 (variants/liblbfgsx_pre.so),
# interleaved on one box: bench.py's cfg4 leg; then the L-BFGS-B tests with histories of that length
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2; do
for m in 30 24; do
for v in base pre; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=$m  "; python scripts/r6/cfg4_leg.py --m $m --iters 80 2>/dev/null | tail -1
done; done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
python -m pytest tests/test_lbfgsb_gpu.py -x -q -m gpu -k "long_histories or 36 or 40" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
