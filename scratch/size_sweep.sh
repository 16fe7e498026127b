#!/bin/bash
# profiles/r02_size_sweep.txt: one bench line per BASELINE size (timed region = groth16_prove on files, warm cache)
echo "# python bench.py --workload N --steps 10 --warmup 2 --no-pmc --no-dropin, one MI355X, table mode"
echo "# N         ms file-to-file  Mconstraints/s  ms host witness  ms resident  qap_ntt_ms  msm_ms  cold prove ms (first key / process warm)  key usable after ms  tables alone ms  digit bits c  digits W  CPU oracle s (threads)"
for n in 100000 200000 400000 800000 1600000 3200000 6400000; do
python bench.py --workload $n --steps 10 --warmup 2 --no-pmc --no-dropin 2>/dev/null | python3 -c "
import json,sys,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; g=d['roofline']['geometry']; cb=d.get('cpu_baseline') or {}
cpu = '%.2f (%d)' % (c['constraints'] / cb['value'], cb['cores']) if cb else '-'
print(f\"{c['constraints']:<11d} {d['ms_per_step']:<16.2f} {d['value']/1e6:<15.1f} {c['prove_ms_host_witness']:<16.2f} {c['prove_ms_hbm_resident']:<12.2f} {c['phase_ms']['qap_ntt']:<11.2f} {c['phase_ms']['msm']:<7.2f} {c['cold_path']['cold_prove_ms_files']:.1f} / {c['cold_path']['cold_prove_ms_files_process_warm']:<28.1f} {c['cold_cache_build_ms']:<20.0f} {c['cold_path']['cold_tables_build_ms']:<16.0f} {g['c']:<13d} {g['W']:<9d} {cpu}\")"
done
for w in aadhaar_standin keyless_standin; do
python bench.py --workload $w --steps 10 --warmup 2 --no-pmc --no-dropin 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; g=d['roofline']['geometry']; cb=d.get('cpu_baseline') or {}
cpu = '%.2f (%d)' % (c['constraints'] / cb['value'], cb['cores']) if cb else '-'
print(f\"$w ({c['constraints']} constraints, synthetic stand-in): file-to-file {d['ms_per_step']:.2f} ms, host witness {c['prove_ms_host_witness']:.2f}, resident {c['prove_ms_hbm_resident']:.2f}, qap {c['phase_ms']['qap_ntt']:.2f}, msm {c['phase_ms']['msm']:.2f}, CPU oracle {cpu}\")"
done
