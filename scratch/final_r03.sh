#!/bin/bash
# round-3 evidence run: default bench line (PMC + CPU baseline), size sweep, kernel trace + one-prove timelines, per-rank time of an
# 8-way shard, the in-process device group with every shard on this GPU.  Outputs under gpurun_out/ (copied to profiles/ by hand).
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_1600k.json 2> gpurun_out/r03_bench_1600k.err
tail -c 2500 gpurun_out/r03_bench_1600k.json
bash scratch/size_sweep.sh > gpurun_out/r03_size_sweep.txt 2>&1
cat gpurun_out/r03_size_sweep.txt
bash scratch/profile_r02.sh r03 > /dev/null 2>&1
head -45 gpurun_out/r03_timeline_one_prove_1600k.txt
head -30 gpurun_out/r03_kernel_trace_bench_1600k.txt | cut -c1-150
(python scratch/shard_rank_time.py 8 0; python scratch/shard_rank_time.py 4 0; python scratch/shard_rank_time.py 2 0) 2>/dev/null > gpurun_out/r03_shard_rank_time.txt
cat gpurun_out/r03_shard_rank_time.txt
python scratch/group_alias_time.py > gpurun_out/r03_group_alias_time.txt 2>/dev/null
cat gpurun_out/r03_group_alias_time.txt
