#!/bin/bash
# round-5 evidence run (outputs under gpurun_out/r05/, copied to profiles/ by hand):
#  default bench line (PMC + CPU baseline + cold path), size sweep incl. 6.4 M constraints, cold path at 1600k / 3200k,
#  kernel trace of bench.py + one-prove timeline, PMC table of every kernel of a prove, per-rank time of 2 / 4 / 8-way shards at
#  1600k AND 3200k, the in-process device group with every shard on this GPU, bench.py --gpus 2 without a launcher
O=gpurun_out/r05
mkdir -p $O
python bench.py > $O/r05_bench_1600k.json 2> $O/r05_bench_1600k.err
tail -c 600 $O/r05_bench_1600k.json; echo
bash scratch/size_sweep.sh > $O/r05_size_sweep.txt 2>&1
cat $O/r05_size_sweep.txt
(ICICLE_SNARK_TRACE_COLD=1 python3 scratch/cold_prove.py 1600000; ICICLE_SNARK_TRACE_COLD=1 python3 scratch/cold_prove.py 3200000) 2>&1 | grep -v "^\[host\]" > $O/r05_cold_path.txt
grep "round\|DEFER" $O/r05_cold_path.txt | cut -c1-260
# kernel trace + timeline of a file-to-file prove inside bench.py's timed loop
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-dropin > /dev/null 2>&1 )
python profiles/summarize_rocpd.py $(ls $O/prof/*/*_results.db | head -1) > $O/r05_kernel_trace_bench_1600k.txt 2>&1
python profiles/timeline_rocpd.py $O/prof -9 > $O/r05_timeline_one_prove_1600k.txt 2>&1
rm -rf $O/prof
head -64 $O/r05_timeline_one_prove_1600k.txt | cut -c1-140
bash scratch/pmc_r05.sh > /dev/null 2>&1
cp gpurun_out/r05_pmc_kernels.txt $O/r05_pmc_kernels_1600k.txt
for n in 1600000 3200000; do
  (LOOP_CONSTRAINTS=$n python scratch/shard_rank_time.py 8 0; LOOP_CONSTRAINTS=$n python scratch/shard_rank_time.py 4 0; LOOP_CONSTRAINTS=$n python scratch/shard_rank_time.py 2 0) 2>/dev/null | sed "s/^/benchmark\/$n: /" >> $O/r05_shard_rank_time.txt
  LOOP_CONSTRAINTS=$n python scratch/group_alias_time.py 2>/dev/null >> $O/r05_group_alias_time.txt
done
cat $O/r05_shard_rank_time.txt $O/r05_group_alias_time.txt
ICICLE_SNARK_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 5 > $O/r05_bench_gpus2_one_gpu.json 2> $O/r05_bench_gpus2_one_gpu.err
tail -c 300 $O/r05_bench_gpus2_one_gpu.json
