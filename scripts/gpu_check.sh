#!/bin/bash
# one gpurun call: the GPU suite, then the default bench line (what the driver runs), both logged under gpurun_out/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/check; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3
python - <<'P'
import json
d=json.load(open('gpurun_out/check/bench.json'))
print('north-star %.2f it/s frac %.3f'%(d['value'], d['roofline']['frac']))
for k in ('cfg2','cfg3','cfg4_lbfgsb','cfg5_batched'):
    g=d.get(k)
    if g: print(k, '%.1f'%g['value'], g['unit'], 'frac %.3f'%g['roofline']['frac'], (g.get('from_x0') or {}).get('value'))
c4=d.get('cfg4_lbfgsb',{}).get('config',{})
print({k:c4.get(k) for k in ('q','n_ord','n_sorted','gcp_crossings','launches_per_iteration','host_syncs_per_iteration','copies_per_iteration','phase_ms_per_iteration')})
print('cpu', d.get('cpu_baseline'))
P
