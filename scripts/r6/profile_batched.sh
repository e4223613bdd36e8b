#!/bin/bash
# the cfg5 leg's rocprofv3 passes alone (kernel trace + FETCH_SIZE + WRITE_SIZE, separate runs), into gpurun_out/prof_r6/batched_*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6; mkdir -p $O; rm -rf $O/batched_trace $O/batched_pmc_fetch $O/batched_pmc_write $O/batched
CMD="python bench.py --workload cfg5-batched --steps 50 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched -o bench -- $CMD > $O/batched.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/batched_pmc_fetch -o bench -- $CMD > $O/batched_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/batched_pmc_write -o bench -- $CMD > $O/batched_pmc_write.log 2>&1
mv $O/batched $O/batched_trace
find $O/batched_trace $O/batched_pmc_fetch $O/batched_pmc_write -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
head -4 $O/batched_trace/*kernel_stats.csv | cut -c1-160
