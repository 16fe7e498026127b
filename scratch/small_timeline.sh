# one-prove timeline of a small benchmark size (default 100000 constraints), EVERY dispatch: rocprofv3 kernel trace of scratch/prove_loop.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LOOP_CONSTRAINTS=${1:-100000}
rocprofv3 --kernel-trace -d /tmp/prof_small -- python3 $R/scratch/prove_loop.py 12 > /tmp/small.log 2>&1
cd $R
tail -1 /tmp/small.log
python profiles/timeline_rocpd.py /tmp/prof_small 8 0
ICICLE_SNARK_TRACE_HOST=1 python3 scratch/prove_loop.py 3 2>&1 | grep "\[host\]" | tail -14
python3 scratch/prove_loop.py 60 | tail -1
