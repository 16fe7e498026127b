#!/bin/bash
# Interleaved A/B/C of prebuilt libraries on one box: icicle-snark_amd/lib (shipped), lib_b, lib_c (when present).
# usage: ab_lib3.sh [check] [constraints ...]   ("check": run the squaring-chain parity tests with each variant first)
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so
VARS="a"
for v in b c; do [ -f icicle-snark_amd/lib_$v/libicicle_snark_hip.so ] && cp icicle-snark_amd/lib_$v/libicicle_snark_hip.so /tmp/lib_$v.so && VARS="$VARS $v"; done
trap 'cp /tmp/lib_a.so $L' EXIT
if [ "$1" = check ]; then
  shift
  for v in $VARS; do [ $v = a ] && continue; cp /tmp/lib_$v.so $L; echo "== parity with lib_$v: $(python -m pytest tests/test_gpu_prove.py -q -x -k 'squaring_chain or golden' 2>&1 | tail -1)"; done
fi
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
for r in 1 2 3; do
  for v in $VARS; do cp /tmp/lib_$v.so $L; echo "-- $n lib_$v : $(run)"; done
done
done
