#!/bin/bash
# round-3 rocprofv3 evidence for profiles/: kernel-trace stats, then PMC passes (separate runs, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r3}
O=gpurun_out/prof_$R; rm -rf $O; mkdir -p $O
# the headline as the driver runs it, all legs and both CPU baselines (un-profiled numbers of this box)
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
CMD="python bench.py --no-cpu --no-batched --no-legs --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
# cfg4 (L-BFGS-B): kernel trace + the two PMC passes + the per-iteration timeline
BCMD="python scripts/bench_lbfgsb.py --n 1e7 --iters 40"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lbfgsb -o b -- $BCMD > $O/lbfgsb.log 2>&1
python scripts/trace_cfg4.py $O/lbfgsb > $O/cfg4_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lbfgsb_pmc_fetch -o b -- $BCMD > $O/lbfgsb_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/lbfgsb_pmc_write -o b -- $BCMD > $O/lbfgsb_pmc_write.log 2>&1
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 --cpu-n 2e5 > $O/bench_cfg4_lbfgsb.json 2> /dev/null
# the matrix-core Gram question (opt-in) has its own script: scripts/experiments/prof_gram_m.sh (final_r3.sh)
# cfg5 batch
KCMD="python bench.py --workload cfg5-batched --steps 50 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched_trace -o bench -- $KCMD > $O/batched_trace.log 2>&1
# row-sharded mode: one rank through RCCL; two ranks of one process on the one device (host-memory sums)
python bench.py --workload sharded --no-cpu > $O/bench_sharded_n1.json 2> /dev/null
LBFGSX_BENCH_DEVICES=0,0 python bench.py --workload sharded --single-process --gpus 2 --no-cpu --n 4e7 > $O/bench_sharded_one_process_2x.json 2> /dev/null
LBFGSX_BENCH_DEVICES=0,0 python bench.py --single-process --gpus 2 --no-cpu --no-legs --n 4e7 --problems-per-gpu 512 > $O/bench_single_process_2x.json 2> /dev/null
LBFGSX_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --no-cpu --no-legs --steps 5 --n 4e7 > $O/bench_two_ranks_one_device.json 2> /dev/null
# keep only what the summary needs (the raw traces are large)
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
find $O -type f | wc -l; du -sh $O
python scripts/summarize_profile.py $R 2>&1 | tail -40
cat $O/bench_default.time
