#!/bin/bash
# A/B of k_trial's cache policy / vectors in flight on the north-star workload (same box, alternating order)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for pol in 0 1 2 3 4 7; do
  echo -n "LBFGSX_TRIAL_POLICY=$pol  "
  LBFGSX_TRIAL_POLICY=$pol python bench.py --no-cpu --no-batched --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f it/s  %.3f ms/it  step %.4f ms'%(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
