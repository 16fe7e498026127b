#!/bin/bash
# kernel-trace timeline of one resident prove of benchmark/1600k with the environment given as arguments (VAR=value ...)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LOOP_CONSTRAINTS=${LOOP_CONSTRAINTS:-1600000}
for a in "$@"; do export "$a"; done
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl_one -- python3 $R/scratch/prove_loop.py 12 > /dev/null 2>&1
python3 $R/profiles/timeline_rocpd.py $R/gpurun_out/prof_tl_one -14 20000 2>&1 | cut -c1-150
rm -rf $R/gpurun_out/prof_tl_one
