#!/bin/bash
# A/B of a compile-time variant on one box: bench with the shipped build, rebuild msm_g1 with $1, bench again, twice
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['config']['phase_ms'], 'H acc', round(d['roofline']['launch_ms'],3), 'bit-heavy', d['config']['prove_ms_bit_heavy_witness_standin'])"; }
echo "-- shipped"; run; run
touch icicle-snark_amd/csrc/msm_impl.h icicle-snark_amd/csrc/msm_g1.hip; make -j16 EXTRA="$1" 2>&1 | grep -E "error|Error"
echo "-- $1"; run; run; run
python scratch/msm_only.py g1 21 | tail -1
