#!/bin/bash
# round-2 evidence run: default bench line, stand-in workloads, kernel trace + timeline, sort solo/PMC for both sort paths
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_1600k.json 2> gpurun_out/r02_bench_1600k.err
python bench.py --workload aadhaar_standin --no-pmc --no-dropin > gpurun_out/r02_bench_aadhaar_standin.json 2> gpurun_out/r02_bench_aadhaar.err
python bench.py --workload keyless_standin --no-pmc --no-dropin > gpurun_out/r02_bench_keyless_standin.json 2> gpurun_out/r02_bench_keyless.err
python bench.py --workload 3200k --no-pmc --no-dropin --no-cpu-baseline > gpurun_out/r02_bench_3200k.json 2> gpurun_out/r02_bench_3200k.err
python bench.py --workload 100k --no-pmc --no-dropin > gpurun_out/r02_bench_100k.json 2> gpurun_out/r02_bench_100k.err
bash scratch/sort_solo.sh > gpurun_out/r02_pmc_sort_solo.txt 2>&1
bash scratch/profile_r02.sh r02 > /dev/null 2>&1
tail -c 3000 gpurun_out/r02_bench_1600k.json
for f in aadhaar_standin keyless_standin 3200k 100k; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r02_bench_$f.json').read().strip().splitlines()[-1]); c=d['config']
print('$f', 'ms/step', round(d['ms_per_step'],3), 'host', c['prove_ms_host_witness'], 'resident', c['prove_ms_hbm_resident'], c['phase_ms'], d.get('cpu_baseline',{}).get('sample'))"; done
cat gpurun_out/r02_pmc_sort_solo.txt | grep -v "^    "
