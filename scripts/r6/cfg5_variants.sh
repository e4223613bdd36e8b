#!/bin/bash
# A/B of library variants (variants/liblbfgsx_<tag>.so, built in the container) on ONE box, interleaved: the cfg5 leg's line per variant
cd $GRAFT_REPO_ROOT
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in 1 2 3; do
for v in base $(ls variants | sed 's/liblbfgsx_//; s/.so//'); do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  python bench.py --workload cfg5-batched --steps 50 --no-cpu --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v rep$rep value %.0f wall %.4f kernel %.4f fev %d' % (d['value'], c['wall_ms_per_step'], c['kernel_ms_per_step'], c['fevals_total']))"
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
