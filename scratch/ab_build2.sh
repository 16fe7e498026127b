#!/bin/bash
# A/B of a compile-time variant on one box (interleaving impossible: two builds): shipped, variant, shipped, variant
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
echo "-- shipped"; run; run
touch icicle-snark_amd/csrc/msm_impl.h; make -j16 EXTRA="$1" 2>&1 | grep -E " error |Error"
echo "-- $1"; run; run
touch icicle-snark_amd/csrc/msm_impl.h; make -j16 2>&1 | grep -E " error |Error"
echo "-- shipped again"; run
touch icicle-snark_amd/csrc/msm_impl.h; make -j16 EXTRA="$1" 2>&1 | grep -E " error |Error"
echo "-- $1 again"; run
