#!/bin/bash
# the whole -m gpu suite, smoke(), then the default bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=15 > gpurun_out/r4/pytest_gpu_full.log 2>&1
tail -30 gpurun_out/r4/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time timeout 900 python bench.py ) > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
tail -5 gpurun_out/r4/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus')}, d.get('roofline',{}).get('frac'))
print('parity', d.get('parity'))
for k,v in d.get('legs',{}).items():
    print(k, v.get('value'), v.get('value_median'), v.get('roofline',{}).get('frac'), v.get('from_x0',{}).get('value') if isinstance(v.get('from_x0'),dict) else None)
PY
