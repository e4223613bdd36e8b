#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats, then PMC passes (separate runs, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r1}
O=gpurun_out/prof_$R; mkdir -p $O
CMD="python bench.py --no-cpu --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
# opt-in Gram-space recursion: same three passes
GCMD="python bench.py --recursion gram --no-cpu --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gram_trace -o bench -- $GCMD > $O/gram_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gram_pmc_fetch -o bench -- $GCMD > $O/gram_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gram_pmc_write -o bench -- $GCMD > $O/gram_pmc_write.log 2>&1
# ... and with the f32 history
HCMD="python bench.py --recursion gram-f32h --no-cpu --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32h_trace -o bench -- $HCMD > $O/f32h_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f32h_pmc_fetch -o bench -- $HCMD > $O/f32h_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/f32h_pmc_write -o bench -- $HCMD > $O/f32h_pmc_write.log 2>&1
# secondary workloads: kernel-trace stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbfgsb -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/lbfgsb.log 2>&1
BCMD="python bench.py --workload cfg5-batched --steps 50 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched_trace -o bench -- $BCMD > $O/batched_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/batched_pmc_fetch -o bench -- $BCMD > $O/batched_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/batched_pmc_write -o bench -- $BCMD > $O/batched_pmc_write.log 2>&1
# plain (un-profiled) numbers
python bench.py > $O/bench_northstar.json 2> /dev/null
python bench.py --recursion gram --no-cpu > $O/bench_northstar_gram.json 2> /dev/null
python bench.py --recursion gram --m 20 --steps 10 --warmup 22 --no-cpu > $O/bench_cfg3_m20_gram.json 2> /dev/null
python bench.py --recursion gram-f32h --no-cpu > $O/bench_northstar_gram_f32h.json 2> /dev/null
python bench.py --workload sharded --no-cpu > $O/bench_sharded_n1.json 2> /dev/null
python bench.py --m 20 --steps 10 --warmup 22 --no-cpu > $O/bench_cfg3_m20.json 2> /dev/null
python bench.py --objective quadratic --n 10000000 --no-cpu > $O/bench_cfg2_quad1e7.json 2> /dev/null
python bench.py --workload cfg5-batched --steps 50 > $O/bench_cfg5_batched.json 2> /dev/null
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 --cpu-n 2e5 > $O/bench_cfg4_lbfgsb.json 2> /dev/null
LBFGSX_GCP_DEVICE_MIN=4096 python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/bench_cfg4_lbfgsb_devmin4096.json 2> /dev/null
find $O -type f | wc -l
