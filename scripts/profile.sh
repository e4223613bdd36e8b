#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats, then PMC passes (separate runs, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r1}
O=gpurun_out/prof_$R; mkdir -p $O
CMD="python bench.py --no-cpu --steps 10 --warmup 11"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
find $O -type f | head -30
