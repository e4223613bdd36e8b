#!/bin/bash
# kernel timeline of the default cfg4 configuration (steady iteration, early iterations, gaps)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_cfg4; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/run.json 2> $O/run.err
python scripts/trace_cfg4.py $O/t > $O/summary.txt 2>&1
find $O -name "*.csv" -delete
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 > $O/plain.json
tail -95 $O/summary.txt
