#!/bin/bash
# interleaved A/B of one environment knob: scratch/ab_env.sh VAR valA valB [rounds]
VAR=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  echo -n "$VAR=$A  "; env $VAR=$A python scratch/prove_loop.py 40 2>/dev/null | tail -1
  echo -n "$VAR=$B  "; env $VAR=$B python scratch/prove_loop.py 40 2>/dev/null | tail -1
done
