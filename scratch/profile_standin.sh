# kernel trace + stats of the stand-in bench command; output → gpurun_out/r02_kernel_trace_bench_aadhaar_standin.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_standin -- python3 $R/bench.py --workload aadhaar_standin --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-dropin > $R/gpurun_out/prof_standin_bench.json 2>/dev/null
cd $R
(echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload aadhaar_standin --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-dropin  (includes the key synthesis and the one-time cache build)"; python profiles/summarize_rocpd.py $(ls gpurun_out/prof_standin/*/*_results.db | head -1)) > gpurun_out/r02_kernel_trace_bench_aadhaar_standin.txt 2>&1
rm -rf gpurun_out/prof_standin
head -30 gpurun_out/r02_kernel_trace_bench_aadhaar_standin.txt | cut -c1-160
