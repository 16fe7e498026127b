#!/bin/bash
# A/B of icicle-snark_amd/lib against lib_b through bench.py lines (file-to-file, resident, phases, scatter), interleaved
L=icicle-snark_amd/lib/libicicle_snark_hip.so
cp $L /tmp/lib_a.so; cp icicle-snark_amd/lib_b/libicicle_snark_hip.so /tmp/lib_b.so
one() { python bench.py --steps 10 --warmup 2 --no-pmc --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$1 file', round(d['ms_per_step'],2), 'host', round(c['prove_ms_host_witness'],2), 'resident', round(c['prove_ms_hbm_resident'],2), 'qap', round(c['phase_ms']['qap_ntt'],2), 'msm', round(c['phase_ms']['msm'],2), 'witness sort', round(r['scatter']['launch_ms'],3))"; }
for r in 1 2 3; do
  cp /tmp/lib_a.so $L; one lib
  cp /tmp/lib_b.so $L; one lib_b
done
cp /tmp/lib_a.so $L
