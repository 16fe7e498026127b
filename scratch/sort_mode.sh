#!/bin/bash
# bench lines with the stream roles dealt by measured hardware pipe (ICICLE_SNARK_PIPE_ROLES: 0 as the streams come, 1 everything apart,
# 2 the sort stream on the front end's pipe), interleaved on one box
for i in 1 2 3 4 5; do for v in ${PIPE_VARIANTS:-2 0 1}; do ICICLE_SNARK_PIPE_ROLES=$v python bench.py --steps 10 --warmup 2 --no-pmc --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('pipe_roles=$v file', round(d['ms_per_step'],2), 'host', round(c['prove_ms_host_witness'],2), 'resident-witness', round(c['prove_ms_hbm_resident'],2), 'qap', round(c['phase_ms']['qap_ntt'],2), 'msm', round(c['phase_ms']['msm'],2), 'witness sort', round(r['scatter']['launch_ms'],2))"; done; done
