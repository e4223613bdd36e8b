#!/bin/bash
# the sweeps' solves at 2c = 32 (m = 16): a row over two lanes of 16 columns (tree) against four lanes of 8 (variants/liblbfgsx_pre.so),
# interleaved on one box: bench.py's cfg4 leg at m = 16; then the L-BFGS-B tests that run 2c = 25..32
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base pre; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=16  "; python scripts/r6/cfg4_leg.py --m 16 --iters 40 2>/dev/null | tail -1
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
python -m pytest tests/test_lbfgsb_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
