cd $GRAFT_REPO_ROOT
for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=2"; do
  env $v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v] it/s from x0 %.1f steady %.1f fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d.get('fx', 0)))"
done
for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  env $v python bench.py --no-cpu --no-batched --no-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v] north-star %.2f it/s' % d['value'])"
done
