#!/bin/bash
# as r6/cfg5_variants.sh for another problem size: bash scripts/r6/cfg5_variants_n.sh N P [reps]
cd $GRAFT_REPO_ROOT
N=$1; P=$2; R=${3:-2}
cp lbfgspp_amd/liblbfgsx.so /tmp/liblbfgsx_base.so
for rep in $(seq 1 $R); do
for v in base $(ls variants | sed 's/liblbfgsx_//; s/.so//'); do
  if [ $v = base ]; then cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so; else cp variants/liblbfgsx_$v.so lbfgspp_amd/liblbfgsx.so; fi
  python bench.py --workload cfg5-batched --steps 50 --no-cpu --verbose --batched-n $N --problems-per-gpu $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('n=$N $v rep$rep value %.0f wall %.4f kernel %.4f fev %d' % (d['value'], c['wall_ms_per_step'], c['kernel_ms_per_step'], c['fevals_total']))"
done; done
cp /tmp/liblbfgsx_base.so lbfgspp_amd/liblbfgsx.so
