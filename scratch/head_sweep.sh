#!/bin/bash
# share of the witness sorted and accumulated under its own upload (ICICLE_SNARK_HEAD_PCT), interleaved, one box
PCTS=${PCTS:-"0 10 15 20 25 30"}
for n in ${@:-1600000}; do
export LOOP_CONSTRAINTS=$n
for r in 1 2; do
for p in $PCTS; do
  echo "-- head $p%: $( ( [ $p = auto ] || export ICICLE_SNARK_HEAD_PCT=$p; python scratch/paths_loop.py 30 2>/dev/null | tail -1 ) )"
done
done
done
