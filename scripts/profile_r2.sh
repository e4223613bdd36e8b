#!/bin/bash
# round-2 rocprofv3 evidence for profiles/: kernel-trace stats, then PMC passes (separate runs, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r2}
O=gpurun_out/prof_$R; mkdir -p $O
CMD="python bench.py --no-cpu --no-batched --no-legs --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
# cfg4 (L-BFGS-B): kernel trace + the two PMC passes
BCMD="python scripts/bench_lbfgsb.py --n 1e7 --iters 40"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbfgsb -o b -- $BCMD > $O/lbfgsb.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lbfgsb_pmc_fetch -o b -- $BCMD > $O/lbfgsb_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lbfgsb_pmc_write -o b -- $BCMD > $O/lbfgsb_pmc_write.log 2>&1
# cfg5 batch
KCMD="python bench.py --workload cfg5-batched --steps 50 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched_trace -o bench -- $KCMD > $O/batched_trace.log 2>&1
# plain (un-profiled) numbers, same box
python bench.py > $O/bench_northstar.json 2> /dev/null
LBFGSX_FUSE_POST=0 python bench.py --no-cpu --no-batched --no-legs > $O/bench_northstar_unfused.json 2> /dev/null
python bench.py --m 20 --steps 10 --warmup 22 --no-cpu --no-batched --no-legs > $O/bench_cfg3_m20.json 2> /dev/null
python bench.py --objective quadratic --n 10000000 --no-cpu --no-batched --no-legs > $O/bench_cfg2_quad1e7.json 2> /dev/null
python bench.py --workload cfg5-batched --steps 50 > $O/bench_cfg5_batched.json 2> /dev/null
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 --cpu-n 2e5 > $O/bench_cfg4_lbfgsb.json 2> /dev/null
LBFGSX_GRAM=i8 python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/bench_cfg4_lbfgsb_i8.json 2> /dev/null
LBFGSX_GCP_CHAIN=scan python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/bench_cfg4_lbfgsb_scan.json 2> /dev/null
LBFGSX_GRAM=i8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbfgsb_i8 -o b -- $BCMD > $O/lbfgsb_i8.log 2>&1
LBFGSX_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --no-cpu --steps 5 > $O/bench_two_ranks_one_device.json 2> /dev/null
# keep only what the summary needs (the raw traces are large)
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
find $O -type f | wc -l; du -sh $O
python scripts/summarize_profile.py $R
