cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  LBFGSX_COMPACT_KEEP=$v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keep=$v it/s from x0 %.1f steady %.1f sweeps %d carried %s fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d['stats']['submin_sweeps'], d['stats'].get('gram_carried'), d.get('fx', 0)))"
done
bash scripts/experiments/prof_cfg4.sh 2>&1 | head -14
