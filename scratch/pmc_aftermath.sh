#!/bin/bash
# does the default bench.py run (three rocprofv3 PMC child passes BEFORE the parent touches the GPU) time a slower prove than the same run without them?
one() { python bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*', 'file', round(d['ms_per_step'],2), 'host', round(c['prove_ms_host_witness'],2), 'resident', round(c['prove_ms_hbm_resident'],2), 'qap', round(c['phase_ms']['qap_ntt'],2), 'msm', round(c['phase_ms']['msm'],2), 'loadavg', c['cold_path'].get('host_loadavg_1m'), 'tables', round(c['cold_path']['cold_tables_build_ms']))"; }
for i in 1 2 3; do
  one --no-dropin --no-cpu-baseline
  one --no-pmc --no-dropin --no-cpu-baseline
done
uptime
