#!/bin/bash
# large-bucket threshold sweep: factor × average with a floor; stand-ins and benchmark/1600k
for cfg in "10 512" "3 64" "2 32" "4 128" "3 32"; do
  set -- $cfg
  echo "== factor $1 floor $2"
  ICICLE_SNARK_LARGE_FACTOR=$1 ICICLE_SNARK_LARGE_FLOOR=$2 python scratch/standin_loop.py 2>/dev/null | tail -6
  ICICLE_SNARK_LARGE_FACTOR=$1 ICICLE_SNARK_LARGE_FLOOR=$2 ICICLE_SNARK_SPARSE_B=0 python scratch/standin_loop.py 2>/dev/null | tail -6 | head -1
  ICICLE_SNARK_LARGE_FACTOR=$1 ICICLE_SNARK_LARGE_FLOOR=$2 LOOP_WORKLOAD=keyless_standin python scratch/standin_loop.py 2>/dev/null | tail -6 | head -1
  ICICLE_SNARK_LARGE_FACTOR=$1 ICICLE_SNARK_LARGE_FLOOR=$2 python scratch/prove_loop.py 30 2>/dev/null | tail -1
done
