#!/bin/bash
# A/B of icicle-snark_amd/lib vs lib_b on a stand-in workload (default aadhaar_standin), interleaved
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
python scratch/standin_loop.py 3 > /dev/null 2>&1
for r in 1 2 3; do
  cp /tmp/lib_a.so $L; echo "-- lib   : $(python scratch/standin_loop.py 40 2>/dev/null | head -1)"
  cp /tmp/lib_b.so $L; echo "-- lib_b : $(python scratch/standin_loop.py 40 2>/dev/null | head -1)"
done
cp /tmp/lib_a.so $L
