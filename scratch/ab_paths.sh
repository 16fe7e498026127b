#!/bin/bash
# interleaved comparison of environment variants through the three entry points (files / host buffer / resident): ab_paths.sh "<env a>" "<env b>" ...
python scratch/paths_loop.py 3 > /dev/null 2>&1
for r in 1 2 3; do
  for v in "$@"; do
    [ "$v" = "-" ] && e="" || e="$v"
    echo "-- [$v] : $(env $e python scratch/paths_loop.py 30 2>/dev/null | tail -1)"
  done
done
