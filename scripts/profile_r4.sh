#!/bin/bash
# round-4 rocprofv3 evidence for profiles/: kernel-trace stats, then the PMC passes (separate runs, as the guide prescribes),
# for the headline AND for every leg bench.py quotes a roofline for (cfg2, cfg3, cfg4 m=10, cfg4 m=20, cfg5)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r4}
O=gpurun_out/prof_$R; rm -rf $O; mkdir -p $O
prof3 () {  # prof3 <prefix> <basename> <command...>: kernel trace + FETCH_SIZE pass + WRITE_SIZE pass
  local p=$1 b=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$p -o $b -- "$@" > $O/$p.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${p}_pmc_fetch -o $b -- "$@" > $O/${p}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${p}_pmc_write -o $b -- "$@" > $O/${p}_pmc_write.log 2>&1
}
CMD="python bench.py --no-cpu --no-batched --no-legs --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
prof3 cfg2 bench python bench.py --no-cpu --no-batched --no-legs --objective quadratic --n 10000000
prof3 cfg3 bench python bench.py --no-cpu --no-batched --no-legs --m 20 --steps 10 --warmup 22
# cfg4 (L-BFGS-B), m = 10 and m = 20: kernel trace (+ the per-iteration timeline) and the two PMC passes
prof3 cfg4_m10 b python scripts/bench_lbfgsb.py --n 1e7 --iters 40
cp -r $O/cfg4_m10 $O/lbfgsb; cp -r $O/cfg4_m10_pmc_fetch $O/lbfgsb_pmc_fetch; cp -r $O/cfg4_m10_pmc_write $O/lbfgsb_pmc_write
python scripts/trace_cfg4.py $O/cfg4_m10 > $O/cfg4_timeline.txt 2>&1
prof3 cfg4_m20 b python scripts/bench_lbfgsb.py --n 1e7 --m 20 --iters 60
TL_ITERS=60 python scripts/trace_cfg4.py $O/cfg4_m20 > $O/cfg4_m20_timeline.txt 2>&1
python scripts/bench_lbfgsb.py --n 1e7 --iters 40 --cpu-n 2e5 > $O/bench_cfg4_lbfgsb.json 2> /dev/null
# cfg5 batch
prof3 batched bench python bench.py --workload cfg5-batched --steps 50 --no-cpu
mv $O/batched $O/batched_trace
# row-sharded mode: one rank through RCCL; two ranks of one process on the one device (host-memory sums)
python bench.py --workload sharded --no-cpu > $O/bench_sharded_n1.json 2> /dev/null
LBFGSX_BENCH_DEVICES=0,0 python bench.py --workload sharded --single-process --gpus 2 --no-cpu --n 4e7 > $O/bench_sharded_one_process_2x.json 2> /dev/null
# keep only what the summaries need (the raw traces are large)
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete
find $O -path "*pmc*" -name "*kernel_trace.csv" -delete
find $O -name "*kernel_trace.csv" ! -path "*cfg4*" -delete
python scripts/summarize_profile.py $R 2>&1 | tail -30
python scripts/r4/pmc_legs.py $O $R 2>&1 | tail -12
for t in cfg2 cfg3 cfg4_m20; do cp $O/$t/*kernel_stats.csv profiles/${R}_${t}_kernel_stats.csv 2>/dev/null; done
cp $O/cfg4_m20_timeline.txt profiles/${R}_cfg4_m20_timeline.txt 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
du -sh $O
# the benchmark's own 40 iterations of cfg4 against the reference (about 5 minutes of one host core): DRIFT=1
[ "${DRIFT:-0}" = 1 ] && python scripts/drift_curves.py cfg4 --n 1e7 --iters 40 --devmin default > $O/drift_cfg4_1e7_40it.json 2> $O/drift.err
[ "${DRIFT:-0}" = 1 ] && python3 -c "
import json; d=json.load(open('$O/drift_cfg4_1e7_40it.json')); r=d['runs'][0]
print('drift 40 it: counts', r['same_counts'], 'max per-eval', max(r['max_dx_per_evaluation']), 'final', r['max_dx_final_all_coordinates'])"
