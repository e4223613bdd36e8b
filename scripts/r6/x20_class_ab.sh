#!/bin/bash
# the sweeps' solves at 2c = 20 (m = 10): a whole row per lane (20 columns, variants/liblbfgsx_x201.so) against two lanes of 10 (tree),
# interleaved on one box: bench.py's cfg4 leg at m = 10; then the L-BFGS-B tests on the variant
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base x201; do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  echo -n "$v m=10  "; python scripts/r6/cfg4_leg.py --m 10 --iters 40 2>/dev/null | tail -1
done; done
cp variants/liblbfgsx_x201.so lbfgspp_amd/liblbfgsx.so
python -m pytest tests/test_lbfgsb_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
