#!/bin/bash
# interleaved A/B of the G2 accumulation inside the 1600k prove: pairwise tree (default) vs classic walk (ICICLE_SNARK_G2_TREE=0)
# usage: ab_tree.sh [pairs] [extra env for the tree leg, e.g. ICICLE_SNARK_TREE_ROUNDS=4]
export LOOP_CONSTRAINTS=${LOOP_CONSTRAINTS:-1600000}
python scratch/prove_loop.py 3 > /dev/null 2>&1   # inputs cached in /tmp
for r in $(seq 1 ${1:-3}); do
  echo "-- tree    : $(env $2 python scratch/prove_loop.py 30 2>/dev/null | tail -1)"
  echo "-- classic : $(ICICLE_SNARK_G2_TREE=0 python scratch/prove_loop.py 30 2>/dev/null | tail -1)"
done
