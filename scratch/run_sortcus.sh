export LOOP_CONSTRAINTS=1600000
for r in 1 2; do
for k in 0 2 4 8; do
  echo "-- sort on every ${k}th CU: $(ICICLE_SNARK_SORT_CUS=$k python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
done
done
