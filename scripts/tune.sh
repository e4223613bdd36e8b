#!/bin/bash
# sweep launch geometry of the two-loop kernels on the north-star workload (run on the GPU box)
mkdir -p gpurun_out
out=gpurun_out/tune.txt; : > $out
for ch in 0 1; do for nt in 0 1; do for u in 4 8; do for g in 256 512 768 1024 1536; do
  r=$(LBFGSX_CHUNKED=$ch LBFGSX_NT=$nt LBFGSX_UNROLL=$u LBFGSX_GRID_CAP=$g timeout 120 python bench.py --no-cpu --steps 10 --warmup 11 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('it/s %.2f  launch_ms %.4f  achieved %.0f GB/s  hv_ms %.3f copy %.0f triad %.0f' % (d['value'], r['avg_launch_ms'], r['achieved'], r['apply_Hv_ms'], r['stream_copy_GBs'], r['stream_triad_GBs']))")
  echo "chunked=$ch nt=$nt unroll=$u grid=$g : $r" | tee -a $out
done; done; done; done
