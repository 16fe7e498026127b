#!/bin/bash
# round-6 evidence run (outputs under gpurun_out/r06/, copied to profiles/ by hand):
#  default bench line (PMC + CPU baseline + cold-path cycles + two proves in flight), size sweep incl. 6.4 M constraints,
#  kernel trace of bench.py + one-prove timeline, PMC table of every kernel of a prove, ten default-ish bench runs for the spread of the
#  scatter pass (sort modes), evict-and-prove cycles, two proves in flight
O=gpurun_out/r06
mkdir -p $O
python bench.py > $O/r06_bench_1600k.json 2> $O/r06_bench_1600k.err
tail -c 600 $O/r06_bench_1600k.json; echo
bash scratch/size_sweep.sh > $O/r06_size_sweep.txt 2>&1
cat $O/r06_size_sweep.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-dropin > /dev/null 2>&1 )
python profiles/summarize_rocpd.py $(ls $O/prof/*/*_results.db | head -1) > $O/r06_kernel_trace_bench_1600k.txt 2>&1
python profiles/timeline_rocpd.py $O/prof -9 > $O/r06_timeline_one_prove_1600k.txt 2>&1
rm -rf $O/prof
head -64 $O/r06_timeline_one_prove_1600k.txt | cut -c1-140
bash scratch/pmc_r05.sh > /dev/null 2>&1
cp gpurun_out/r05_pmc_kernels.txt $O/r06_pmc_kernels_1600k.txt
# spread of the scatter pass inside the prove over ten bench runs on this box (file-to-file ms, scatter launch ms, fraction of HBM peak)
for i in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --steps 10 --warmup 2 --no-pmc --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline'].get('scatter',{})
print(f\"run $i: file-to-file {d['ms_per_step']:.3f} ms, resident {d['config']['prove_ms_hbm_resident']:.3f}, qap {d['config']['phase_ms']['qap_ntt']:.3f}, msm {d['config']['phase_ms']['msm']:.3f}, scatter launch {s.get('launch_ms',0):.3f} ms (head {s.get('launch_ms_head',0):.3f} + tail {s.get('launch_ms_tail',0):.3f}) = {s.get('frac',0):.4f} of HBM peak\")"
done > $O/r06_sort_modes.txt
cat $O/r06_sort_modes.txt
python scratch/cold_cycles.py 5 none 2>&1 | grep -E "cycle|load alone" > $O/r06_cold_cycles.txt
cat $O/r06_cold_cycles.txt
python scratch/two_in_flight.py 20 2>&1 | grep "ms per prove" > $O/r06_two_in_flight.txt
cat $O/r06_two_in_flight.txt
