cd $GRAFT_REPO_ROOT
for v in 8 1 8 1 4; do
  LBFGSX_GCP_PIECES=$v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pieces=$v it/s from x0 %.1f steady %.1f first it %.2f ms fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d['per_iter_ms'][0], d.get('fx', 0)))"
done
