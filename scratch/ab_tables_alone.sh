#!/bin/bash
# interleaved A/B of lib (current) vs lib_b (older commit) for the table build alone
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
for r in 1 2; do
  for v in a b; do cp /tmp/lib_$v.so $L; echo "== lib_$v"; python3 scratch/tables_alone.py 2>/dev/null | grep rep; done
done
cp /tmp/lib_a.so $L
