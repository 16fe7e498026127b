ICICLE_SNARK_TRACE_HOST=1 python scratch/file_trace.py 2>&1 | grep "head .* of the witness" | head -8
for r in 1 2 3; do python scratch/paths_loop.py 30 2>/dev/null | tail -1; done
python bench.py --no-dropin --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['prove_ms_hbm_resident'])"
