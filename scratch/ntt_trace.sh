#!/bin/bash
# per-kernel durations of the QAP front end inside resident proves (kernel trace), fused middle pass on / off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LOOP_CONSTRAINTS=${LOOP_CONSTRAINTS:-1600000}
for v in mid nomid; do
  [ $v = nomid ] && export ICICLE_SNARK_NTT_MID=0 || unset ICICLE_SNARK_NTT_MID
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ntt_$v -o t -- python3 $R/scratch/prove_loop.py 12 > /dev/null 2>&1
  echo "== $v"
  python3 $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/prof_ntt_$v -name '*_results.db' | head -1) 2 2>&1 | grep -E "ntt_|spmv|kernel  " | cut -c1-170
  python3 $R/profiles/timeline_rocpd.py $R/gpurun_out/prof_ntt_$v -14 20000 2>&1 | grep -E "ntt_|spmv|sort2|one prove" | cut -c1-120
  rm -rf $R/gpurun_out/prof_ntt_$v
done
