# one-step timeline of one rank of an 8-way shard of benchmark/1600k (rocprofv3 kernel trace of scratch/shard_loop.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scratch/shard_loop.py ${1:-8} 3 > /dev/null 2>&1
rocprofv3 --kernel-trace -d /tmp/prof_shard -- python3 $R/scratch/shard_loop.py ${1:-8} 12 > /tmp/shard.log 2>&1
cd $R
tail -1 /tmp/shard.log
python profiles/timeline_rocpd.py /tmp/prof_shard 8 0
ICICLE_SNARK_TRACE_HOST=1 python3 scratch/shard_loop.py ${1:-8} 3 2>&1 | grep "\[host\]" | tail -9
python3 scratch/shard_loop.py ${1:-8} 40 | tail -1
