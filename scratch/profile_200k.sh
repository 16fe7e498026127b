cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof9 -- python3 $R/bench.py --constraints 200000 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python profiles/timeline_rocpd.py gpurun_out/prof9 5 > gpurun_out/prof9_timeline.txt 2>&1
rm -rf gpurun_out/prof9
sed -i 's/(e - s) > 30_000/(e - s) > 30_000/' profiles/timeline_rocpd.py
cat gpurun_out/prof9_timeline.txt
