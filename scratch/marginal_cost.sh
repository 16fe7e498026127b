#!/bin/bash
# Marginal cost of the stages of one prove, by duplication (work-bound overlapped schedule: a stage's solo duration says little
# about what it costs the prove).  icicle-snark_amd/lib_b must be a build with -DICICLE_SNARK_EXPERIMENTS
#   make -j8 OBJDIR=build/obj_b LIBDIR=icicle-snark_amd/lib_b EXTRA=-DICICLE_SNARK_EXPERIMENTS icicle-snark_amd/lib_b/libicicle_snark_hip.so
# ISNARK_DUP bit 0 runs the first reduction level of every G1 set twice, bit 1 the same for G2, bit 2 the G1 accumulations, bit 3 the G2
# accumulation, bit 4 H's digit sort, bit 5 the witness digit sort, bit 6 the large-bucket kernels (all idempotent; DUP_MASKS="0 16 32 64"
# selects).  lib_c (optional): another variant, run interleaved.
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
[ -f icicle-snark_amd/lib_c/libicicle_snark_hip.so ] && cp icicle-snark_amd/lib_c/libicicle_snark_hip.so /tmp/lib_c.so
trap 'cp /tmp/lib_a.so $L' EXIT
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
for r in 1 2; do
  cp /tmp/lib_a.so $L; echo "-- $n shipped : $(run)"
  cp /tmp/lib_b.so $L
  for m in ${DUP_MASKS:-0 1 2 4 8}; do echo "-- $n dup=$m  : $(ISNARK_DUP=$m run)"; done
  if [ -f /tmp/lib_c.so ]; then cp /tmp/lib_c.so $L; echo "-- $n lib_c   : $(run)"; fi
done
done
