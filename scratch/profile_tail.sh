cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof12 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof12_bench.json 2>/dev/null
cd $R
sed 's/(e - s) > 30_000/(e - s) > 0/' profiles/timeline_rocpd.py > /tmp/tl.py
python /tmp/tl.py gpurun_out/prof12 6 > gpurun_out/prof12_timeline.txt 2>&1
rm -rf gpurun_out/prof12
awk '$2 > 11.5' gpurun_out/prof12_timeline.txt | cut -c1-110
