cd $GRAFT_REPO_ROOT
run() { env "$@" python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*: from x0 %.1f steady %.1f sweeps %d fused %d fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d['stats']['submin_sweeps'], d['stats']['submin_fused_sweeps'], d.get('fx', 0))); print('   ', [round(v,1) for v in d['per_iter_ms'][:20]])"; }
for v in 16384 65536 262144 1048576; do run LBFGSX_LU_MAX=$v; done
run LBFGSX_LU_MAX=16384
