export LOOP_CONSTRAINTS=1600000
for r in 1 2 3; do
for k in 3 4 5 2; do
  echo "-- $k upload lanes: $(ICICLE_SNARK_UPLOAD_LANES=$k python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
done
done
