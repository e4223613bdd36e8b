#!/bin/bash
# end-of-round evidence that needs the GPU: the Gram table, the example programs run to the end (bracketing) / within a
# budget (comparison), the 40-iteration cfg4 drift curve, the single-process protocol lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r3; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_GRAM" ]; then MS="10 12 14 15" GS="dd i8" bash scripts/experiments/prof_gram_m.sh > $O/gram_dd_vs_i8.txt 2>&1; fi
( cd tests/cpp/bin
  echo "== example-rosenbrock-bracketing (8 dimensions x 1024 random starts), GPU build against include/ vs reference-header build"
  T0=$SECONDS; ./example-rosenbrock-bracketing.gpu > /tmp/br_gpu.txt; echo "exit $?; gpu build: $((SECONDS - T0)) s"
  ./example-rosenbrock-bracketing.ref > /tmp/br_ref.txt
  if cmp -s /tmp/br_gpu.txt /tmp/br_ref.txt; then echo "outputs identical ($(grep -c 'Test passed' /tmp/br_gpu.txt) x 'Test passed!')"; else echo "OUTPUTS DIFFER"; diff /tmp/br_gpu.txt /tmp/br_ref.txt | head; fi
  echo "== example-rosenbrock-comparison (12 dimensions x 1024 starts x 4 line searches), ${CMP_BUDGET:-240} s budget for the GPU build"
  timeout ${CMP_BUDGET:-240} ./example-rosenbrock-comparison.gpu > /tmp/cmp_gpu.txt; echo "exit $? (124 = budget reached)"
  ./example-rosenbrock-comparison.ref > /tmp/cmp_ref.txt
  python3 - <<'P'
import re
def blocks(t):
    return {int(m.group(1)): m.group(2) for m in re.finditer(r"^n = (\d+)\n  Average #calls:\n((?:  LineSearch.*\n){4})", t, re.M)}
g, r = blocks(open("/tmp/cmp_gpu.txt").read()), blocks(open("/tmp/cmp_ref.txt").read())
print("dimensions completed on the GPU: %s; identical to the reference's blocks: %s" % (sorted(g), all(g[k] == r[k] for k in g)))
for k in sorted(g):
    print("n = %d\n%s" % (k, g[k]), end="")
P
  for e in example-quadratic example-rosenbrock example-rosenbrock-box; do echo "== $e (GPU build)"; ./$e.gpu | head -4; done
) > $O/reference_examples.txt 2>&1
LBFGSX_BENCH_DEVICES=0,0 python bench.py --single-process --gpus 2 --no-cpu --no-legs --n 4e7 --problems-per-gpu 512 > $O/bench_single_process_2x.json 2> /dev/null
if [ -z "$SKIP_DRIFT" ]; then python scripts/drift_curves.py cfg4 --n 1e7 --iters 40 --devmin default > $O/drift_cfg4_1e7_40it.json 2> $O/drift.err; python -c "
import json; d=json.load(open('$O/drift_cfg4_1e7_40it.json')); print(str(d)[:600])"; fi
head -14 $O/reference_examples.txt
