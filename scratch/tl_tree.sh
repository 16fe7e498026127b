#!/bin/bash
# kernel-trace timeline of one resident 1600k prove, tree vs classic G2 accumulation
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LOOP_CONSTRAINTS=1600000
for v in tree classic; do
  [ $v = classic ] && export ICICLE_SNARK_G2_TREE=0 || unset ICICLE_SNARK_G2_TREE
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl_$v -- python3 $R/scratch/prove_loop.py 12 > /dev/null 2>&1
  echo "== $v: one resident prove"
  python3 $R/profiles/timeline_rocpd.py $R/gpurun_out/prof_tl_$v -14 20000 2>&1 | cut -c1-150
  rm -rf $R/gpurun_out/prof_tl_$v
done
