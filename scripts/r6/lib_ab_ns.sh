#!/bin/bash
# two library builds against each other on the north-star (and cfg2), interleaved on one box: variants/liblbfgsx_<tag>.so vs the tree's
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base $(ls variants | sed 's/liblbfgsx_//; s/.so//'); do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  python bench.py --no-cpu --no-batched --no-legs --steps 20 $NS_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v rep$rep value %.3f apply_Hv_ms %.4f avg_launch_ms %.5f' % (d['value'], r['apply_Hv_ms'], r['avg_launch_ms']))"
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
