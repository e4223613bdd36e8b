# A/B of the traversal-direction alternation and the q cache policy on the headline workload
for rep in 1 2; do
for zz in 1 0; do for qp in 0 1 2 3; do
  r=$(LBFGSX_ZIGZAG=$zz LBFGSX_Q_POLICY=$qp python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['roofline']['achieved']), round(d['roofline']['apply_Hv_ms'],3))")
  echo "northstar zigzag=$zz q_policy=$qp : $r"
done; done; done
