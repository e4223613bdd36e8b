#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_cfg4; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/run.json 2> $O/run.err
python scripts/trace_cfg4.py $O/t > $O/summary.txt 2>&1
find $O -name "*.csv" -delete
python -c "
import json; d=json.load(open('$O/run.json')); print({k:d['stats'][k] for k in d['stats']}); print(d['per_iter_ms'])"
tail -24 $O/summary.txt
