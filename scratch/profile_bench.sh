cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof7 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof7_bench.json 2>/dev/null
cd $R
python profiles/summarize_rocpd.py $(ls gpurun_out/prof7/*/*_results.db | head -1) > gpurun_out/prof7_summary.txt 2>&1
python profiles/timeline_rocpd.py gpurun_out/prof7 6 > gpurun_out/prof7_timeline.txt 2>&1
rm -rf gpurun_out/prof7
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r1e.json 2> gpurun_out/bench_r1e.err
cat gpurun_out/bench_r1e.json
