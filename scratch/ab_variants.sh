#!/bin/bash
# interleaved comparison of environment variants on the shipped library: ab_variants.sh "<env a>" "<env b>" ... (each a quoted list of VAR=value, "-" = none)
export LOOP_CONSTRAINTS=${LOOP_CONSTRAINTS:-1600000}
python scratch/prove_loop.py 3 > /dev/null 2>&1
for r in 1 2 ${ROUNDS3:+3}; do
  for v in "$@"; do
    [ "$v" = "-" ] && e="" || e="$v"
    echo "-- [$v] : $(env $e python scratch/prove_loop.py 30 2>/dev/null | tail -1)"
  done
done
