#!/bin/bash
# the whole -m gpu suite, smoke(), then the default bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --durations=12 > gpurun_out/r5/pytest_gpu_full.log 2>&1
tail -25 gpurun_out/r5/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time timeout 900 python bench.py --full-json gpurun_out/r5/bench_default_full.json ) > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err
tail -5 gpurun_out/r5/bench_default.err
python3 - <<'PY'
import json
raw=open('gpurun_out/r5/bench_default.json').read().strip().splitlines()[-1]
d=json.loads(raw)
print('line bytes', len(raw), 'last key', list(d)[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus')}, d.get('roofline',{}).get('frac'))
print('parity', d.get('parity'))
print(json.dumps(d['legs_digest'], indent=0))
PY
