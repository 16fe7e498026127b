python -m pytest tests -m gpu -x -q 2>&1 | tail -4
PCTS="0 auto" bash scratch/head_sweep.sh 1600000 2>&1 | tee gpurun_out/r04_head_sweep_e.txt
for wl in aadhaar_standin; do for p in 0 auto; do echo "-- $wl head $p: $( ( [ $p = auto ] || export ICICLE_SNARK_HEAD_PCT=$p; LOOP_WORKLOAD=$wl python scratch/standin_loop.py 20 2>/dev/null | head -2 | tr '\n' ' ') )"; done; done 2>&1 | tee -a gpurun_out/r04_head_sweep_e.txt
bash scratch/pmc_r04.sh 2>&1 | cut -c1-260
python bench.py > gpurun_out/r04_bench_1600k_a.json 2> gpurun_out/r04_bench_1600k_a.err; tail -c 3000 gpurun_out/r04_bench_1600k_a.json
ICICLE_SNARK_BENCH_DEVICES=0,0 python bench.py --gpus 2 --steps 5 > gpurun_out/r04_bench_gpus2_alias.json 2> gpurun_out/r04_bench_gpus2_alias.err; tail -c 2500 gpurun_out/r04_bench_gpus2_alias.json; tail -5 gpurun_out/r04_bench_gpus2_alias.err
