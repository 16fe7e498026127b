#!/bin/bash
# kernel + memory-copy trace of file-to-file proves (the reference's timed region) → one-prove timeline with the witness upload
# usage: file_timeline.sh <tag> [env assignments ...]     output: gpurun_out/<tag>_timeline_files_1600k.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_$TAG -- python3 $R/scratch/paths_loop.py 6 > $R/gpurun_out/prof_${TAG}.log 2>&1
cd $R
python profiles/timeline_rocpd.py gpurun_out/prof_$TAG 8 > gpurun_out/${TAG}_timeline_files_1600k.txt 2>&1
rm -rf gpurun_out/prof_$TAG
cat gpurun_out/${TAG}_timeline_files_1600k.txt
