# kernel trace + stats of the default bench command, one-prove timeline; outputs under gpurun_out/ (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-dropin > $R/gpurun_out/prof_${TAG}_bench.json 2>/dev/null
cd $R
python profiles/summarize_rocpd.py $(ls gpurun_out/prof_$TAG/*/*_results.db | head -1) > gpurun_out/${TAG}_kernel_trace_bench_1600k.txt 2>&1
python profiles/timeline_rocpd.py gpurun_out/prof_$TAG 6 > gpurun_out/${TAG}_timeline_one_prove_1600k.txt 2>&1
rm -rf gpurun_out/prof_$TAG
cat gpurun_out/${TAG}_timeline_one_prove_1600k.txt
head -40 gpurun_out/${TAG}_kernel_trace_bench_1600k.txt
