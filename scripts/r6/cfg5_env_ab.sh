#!/bin/bash
# cfg5 leg with an environment switch off / on, interleaved on one box: bash scripts/r6/cfg5_env_ab.sh VAR [reps]
cd $GRAFT_REPO_ROOT
V=$1; R=${2:-3}
for rep in $(seq 1 $R); do for val in 0 1; do
  env $V=$val python bench.py --workload cfg5-batched --steps 50 --no-cpu --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$V=$val rep$rep value %.0f wall %.4f kernel %.4f launches/step %.2f fev %d' % (d['value'], c['wall_ms_per_step'], c['kernel_ms_per_step'], c['launches_per_step'], c['fevals_total']))"
done; done
