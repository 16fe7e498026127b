#!/bin/bash
# interleaved comparison of several "VAR=val VAR2=val2" settings: scratch/ab_combo.sh rounds "A=1 B=2" "A=0 B=2" ...
R=$1; shift
for i in $(seq $R); do
  for combo in "$@"; do
    echo -n "[$combo]  "; env $combo python scratch/prove_loop.py 40 2>/dev/null | tail -1
  done
done
