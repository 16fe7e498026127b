cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof10 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python profiles/timeline_rocpd.py gpurun_out/prof10 11 > gpurun_out/prof10_timeline.txt 2>&1
rm -rf gpurun_out/prof10
cat gpurun_out/prof10_timeline.txt
