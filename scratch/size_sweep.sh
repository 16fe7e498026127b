#!/bin/bash
# profiles/r01_size_sweep.txt: one bench line per BASELINE size
echo "# python bench.py --constraints N --steps 10 --warmup 2 --no-cpu-baseline, one MI355X, table mode, witness resident in HBM"
echo "# N         ms/prove  Mconstraints/s  qap_ntt_ms  msm_ms  ms with host witness  cold cache build ms  digit bits c  digits W"
for n in 100000 200000 400000 800000 1600000 3200000; do
python bench.py --constraints $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; g=d['roofline']['geometry']
print(f\"{c['constraints']:<11d} {d['ms_per_step']:<9.2f} {d['value']/1e6:<15.1f} {c['phase_ms']['qap_ntt']:<11.2f} {c['phase_ms']['msm']:<7.2f} {c['prove_ms_with_witness_over_pcie']:<21.2f} {c['cold_cache_build_ms']:<20.0f} {g['c']:<13d} {g['W']}\")"
done
