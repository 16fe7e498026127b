#!/bin/bash
# A/B of one environment variable on the shipped library, interleaved: ab_env.sh VAR A B [constraints ...]
VAR=$1; A=$2; B=$3; shift 3
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
run() { python scratch/prove_loop.py 40 2>/dev/null | tail -1; }
for r in 1 2 3; do
  echo "-- $n $VAR=$A : $(env $VAR=$A python scratch/prove_loop.py 40 2>/dev/null | tail -1)"
  echo "-- $n $VAR=$B : $(env $VAR=$B python scratch/prove_loop.py 40 2>/dev/null | tail -1)"
done
done
