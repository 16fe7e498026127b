#!/bin/bash
# interleaved comparison of prebuilt libraries (icicle-snark_amd/lib_ab/*.so) on a stand-in workload (LOOP_WORKLOAD, default aadhaar_standin)
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_shipped.so
python scratch/standin_loop.py 3 > /dev/null 2>&1
for r in 1 2; do for v in icicle-snark_amd/lib_ab/*.so; do cp $v $L; echo "-- $(basename $v .so): $(python scratch/standin_loop.py 40 2>/dev/null | head -1)"; done; done
cp /tmp/lib_shipped.so $L
