#!/bin/bash
# solo kernel record of the G2 accumulation, classic walk vs pairwise tree (scratch/g2_solo.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in tree classic; do
  [ $v = classic ] && export ICICLE_SNARK_G2_TREE=0 || unset ICICLE_SNARK_G2_TREE
  echo "== $v"
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_g2_$v -o g2 -- python3 $R/scratch/g2_solo.py ${1:-1600002} 6 2>&1 | grep -E "^call|equal|Error|error" | tail -5
  python3 $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/prof_g2_$v -name '*_results.db' | head -1) 2>&1 | grep -E "tree_|msm_accumulate|zeta|kernel  |total kernel" | cut -c1-160
  rm -rf $R/gpurun_out/prof_g2_$v
done
