cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof10 -- python3 $R/bench.py --constraints 100000 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof10_bench.json 2>/dev/null
cd $R
sed 's/(e - s) > 30_000/(e - s) > 4_000/' profiles/timeline_rocpd.py > /tmp/tl.py
python /tmp/tl.py gpurun_out/prof10 5 > gpurun_out/prof10_timeline.txt 2>&1
rm -rf gpurun_out/prof10
cat gpurun_out/prof10_timeline.txt
