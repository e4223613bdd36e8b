cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 10 20 40; do it=40; [ $m -gt 16 ] && it=60; [ $m -gt 30 ] && it=100
python scripts/bench_lbfgsb.py --n 1e7 --m $m --iters $it 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stats']
print('m=%d steady %.1f x0 %.1f sweeps %d calls %d carried %d identities %d' % (d['m'], d['steady_it_per_s'], d['it_per_s'], st['submin_sweeps'], st['submin_calls'], st['gram_carried'], st['rhs_identities']))"
done
