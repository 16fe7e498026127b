#!/bin/bash
# A/B of two prebuilt libraries on one box, interleaved, all three entry points: icicle-snark_amd/lib (shipped) against lib_b
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
for r in 1 2 3; do
  cp /tmp/lib_a.so $L; echo "-- lib   : $(python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
  cp /tmp/lib_b.so $L; echo "-- lib_b : $(python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
done
done
cp /tmp/lib_a.so $L
