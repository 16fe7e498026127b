#!/bin/bash
# solo time + PMC traffic of the witness digit sort, both implementations (ICICLE_SNARK_SORT_SOLO serialises it before the QAP)
for v in 1 0; do
  ICICLE_SNARK_SORT_SOLO=1 ICICLE_SNARK_LDS_SORT=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['scatter']
print('LDS_SORT=$v solo sort ms', round(s['launch_ms'],3), 'achieved GB/s', round(s['achieved']), 'traffic MB', None if s['traffic'] is None else round(s['traffic']/1e6), 'alg MB', round(s['algorithmic_bytes']/1e6))
for k,v in (d['roofline'].get('pmc_detail') or {}).items():
    if k != 'acc_h': print('   ', k, v)"
done
