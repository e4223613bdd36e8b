# A/B of the solve+sweep fusion on cfg4 (LBFGSX_SWEEP_SOLVE_FUSE=0 restores the separate passes), then the kernel table
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  LBFGSX_SWEEP_SOLVE_FUSE=$v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse=$v it/s from x0 %.1f steady %.1f sweeps %d fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d['stats']['submin_sweeps'], d.get('fx', 0)))"
done
bash scripts/experiments/prof_cfg4.sh
