cd $GRAFT_REPO_ROOT
export ICICLE_SNARK_BENCH_DEVICE=0 ICICLE_SNARK_BENCH_EXCHANGE=gloo ICICLE_SNARK_BENCH_DEVICES=0,0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --constraints 400000 2> gpurun_out/bench2.err | tail -1 > gpurun_out/bench2.json
python3 -c "
import json
d=json.loads(open('gpurun_out/bench2.json').read())
c=d['config']
print(d['n_gpus'], d['ms_per_step'], d['value'], d['scaling'])
print(c['host']); print(c['timed_region'][:200]); print(c['device_group']); print('rank-per-gpu', c['prove_ms_rank_per_gpu'], c['rank_per_gpu_exchange'])
print(d['roofline']['launch_ms'], d['roofline']['geometry'])
"
tail -5 gpurun_out/bench2.err
