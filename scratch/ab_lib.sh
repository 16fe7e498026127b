#!/bin/bash
# A/B of two prebuilt libraries on one box, interleaved: icicle-snark_amd/lib (shipped) against icicle-snark_amd/lib_b
# usage: ab_lib.sh [constraints ...]   (default 1600000)
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
for r in 1 2 3; do
  cp /tmp/lib_a.so $L; echo "-- $n lib   : $(run)"
  cp /tmp/lib_b.so $L; echo "-- $n lib_b : $(run)"
done
done
cp /tmp/lib_a.so $L
