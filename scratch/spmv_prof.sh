cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof8 -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
cd $R; python profiles/summarize_rocpd.py $(ls gpurun_out/prof8/*/*_results.db | head -1) | grep -i "spmv\|ntt_pass\|coset\|final"; rm -rf gpurun_out/prof8
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"config\"][\"phase_ms\"], d[\"roofline\"][\"launch_ms\"])"
