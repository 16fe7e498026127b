# lib (shipped) vs lib_b, interleaved: small sizes + one rank of an 8-way shard
bash scratch/ab_lib.sh 100000 200000 400000 2>&1 | grep -- "--"
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so
for r in 1 2; do
  cp /tmp/lib_a.so $L; echo "-- rank 0/8 lib  : $(python scratch/shard_rank_time.py 8 0 2>/dev/null | tr '\n' ' ')"
  cp icicle-snark_amd/lib_b/libicicle_snark_hip.so $L; echo "-- rank 0/8 lib_b: $(python scratch/shard_rank_time.py 8 0 2>/dev/null | tr '\n' ' ')"
done
cp /tmp/lib_a.so $L
