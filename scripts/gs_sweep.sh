#!/bin/bash
# A/B of the Gram-space kernel variants on the north-star workload (prints avg launch ms of post / combine)
for v in 0 1 2; do
  LBFGSX_GS_VARIANT=$v timeout 300 python -m pytest tests/test_gram_space_gpu.py -q -x 2>&1 | tail -1
  for rep in 1 2; do
  LBFGSX_GS_VARIANT=$v python bench.py --recursion gram --no-cpu --steps 10 --warmup 12 "$@" 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant $v : %.2f it/s  post %.3f ms (%.0f GB/s)  combine %.3f ms (%.0f GB/s)' % (d['value'], r['avg_launch_ms'], r['achieved'], r['combine']['avg_launch_ms'], r['combine']['achieved']))"
  done
done
