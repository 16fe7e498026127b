#!/bin/bash
# interleaved A/B (lib vs lib_b) of the three entry points at 1600k + a kernel-trace timeline of one resident prove per library
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
export LOOP_CONSTRAINTS=1600000
for r in 1 2 3; do
  cp /tmp/lib_a.so $L; echo "-- lib   : $(python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
  cp /tmp/lib_b.so $L; echo "-- lib_b : $(python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
done
for v in a b; do
  cp /tmp/lib_$v.so $L
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab_$v -- python3 $GRAFT_REPO_ROOT/scratch/prove_loop.py 12 > /dev/null 2>&1 )
  echo "== lib_$v: one resident prove"
  python profiles/timeline_rocpd.py gpurun_out/prof_ab_$v -14 2>&1 | grep -E "sort2|^#" | cut -c1-110
  rm -rf gpurun_out/prof_ab_$v
done
cp /tmp/lib_a.so $L
