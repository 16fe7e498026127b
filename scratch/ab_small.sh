#!/bin/bash
# interleaved A/B of lib vs lib_b at the small sizes + rank 0 of 8 (scratch/shard_rank_time.py) + the aliased device group
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
for n in 100000 200000 400000 1600000; do
export LOOP_CONSTRAINTS=$n
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
for r in 1 2; do
  cp /tmp/lib_a.so $L; echo "-- $n lib   : $(run)"
  cp /tmp/lib_b.so $L; echo "-- $n lib_b : $(run)"
done
done
export LOOP_CONSTRAINTS=1600000
for v in a b; do cp /tmp/lib_$v.so $L; echo "== lib_$v"; python scratch/shard_rank_time.py 2>/dev/null | tail -4; python scratch/group_alias_time.py 2>/dev/null | tail -4; done
cp /tmp/lib_a.so $L
