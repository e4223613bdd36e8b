#!/bin/bash
# round-5 rocprofv3 evidence for profiles/: the round-4 recipe (kernel-trace stats, then the FETCH_SIZE and WRITE_SIZE passes as
# separate runs, for the headline and every leg) under the r5 prefix, plus the timeline with the benchmark's first iteration
bash scripts/profile_r4.sh r5 2>&1 | tail -45
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_r5/cfg4_timeline.txt profiles/r5_cfg4_timeline.txt 2>/dev/null
ls profiles | grep r5_
# only gpurun_out/ travels back from the box: the summaries written under profiles/ go there too
mkdir -p gpurun_out/prof_r5/summaries; cp profiles/r5_* gpurun_out/prof_r5/summaries/ 2>/dev/null
