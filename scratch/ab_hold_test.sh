#!/bin/bash
# the regression test of the deferred-table hold, against the library WITHOUT the hold (lib_b) and with it
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
for v in b a; do cp /tmp/lib_$v.so $L; echo "== lib_$v"; timeout 600 python -m pytest tests/test_gpu_prove.py -x -q -k "waits_for_the_cold_upload" 2>&1 | tail -4; done
cp /tmp/lib_a.so $L
