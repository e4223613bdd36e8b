cd $GRAFT_REPO_ROOT
for v in 156250 16384 156250 16384 65536 32768; do
  LBFGSX_DELTA_CAP=$v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cap=$v it/s from x0 %.1f steady %.1f carried %s first %.1f' % (d['it_per_s'], d['steady_it_per_s'], d['stats'].get('gram_carried'), d['per_iter_ms'][0]))"
done
