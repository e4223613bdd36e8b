#!/bin/bash
# cfg5 leg: tests of the batch, the leg's line, and a rocprofv3 kernel trace of the same command (gpurun_out/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_batched_gpu.py -x -q 2>&1 | tail -6
python bench.py --workload cfg5-batched --steps 50 --verbose > gpurun_out/r6_cfg5.json 2> gpurun_out/r6_cfg5.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_cfg5.json"))
c=d["config"]
print("cfg5", d["value"], "frac", d["roofline"]["frac"], "wall", c["wall_ms_per_step"], "kernel", c["kernel_ms_per_step"], "launches/step", c["launches_per_step"], "fev", c["fevals_total"])
PY
rm -rf gpurun_out/prof_cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o cfg5 -- python bench.py --workload cfg5-batched --steps 50 --no-cpu > gpurun_out/prof_cfg5.log 2>&1; tail -3 gpurun_out/prof_cfg5.log; find gpurun_out/prof_cfg5 | head
f=$(find gpurun_out/prof_cfg5 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r6_batched_kernel_stats.csv
head -12 gpurun_out/r6_batched_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_cfg5
