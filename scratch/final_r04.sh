#!/bin/bash
# round-4 evidence run: default bench line (PMC + CPU baseline), size sweep, kernel trace of bench.py + one-prove timelines (resident
# loop of bench.py and the file-to-file region with its memory copies), per-rank time of an 8-way shard, the in-process device group
# with every shard on this GPU, bench.py --gpus 2 without a launcher.  Outputs under gpurun_out/ (copied to profiles/ by hand).
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_1600k.json 2> gpurun_out/r04_bench_1600k.err
tail -c 1500 gpurun_out/r04_bench_1600k.json
bash scratch/size_sweep.sh > gpurun_out/r04_size_sweep.txt 2>&1
cat gpurun_out/r04_size_sweep.txt
bash scratch/profile_r02.sh r04 > /dev/null 2>&1
head -30 gpurun_out/r04_kernel_trace_bench_1600k.txt | cut -c1-150
bash scratch/file_timeline.sh r04 2>&1 | cut -c1-150 | head -70
(python scratch/shard_rank_time.py 8 0; python scratch/shard_rank_time.py 4 0; python scratch/shard_rank_time.py 2 0) 2>/dev/null > gpurun_out/r04_shard_rank_time.txt
cat gpurun_out/r04_shard_rank_time.txt
python scratch/group_alias_time.py > gpurun_out/r04_group_alias_time.txt 2>/dev/null
cat gpurun_out/r04_group_alias_time.txt
ICICLE_SNARK_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 5 > gpurun_out/r04_bench_gpus2_one_gpu.json 2> gpurun_out/r04_bench_gpus2_one_gpu.err
