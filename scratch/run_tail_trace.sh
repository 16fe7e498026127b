cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof_tail -- python3 $R/scratch/paths_loop.py 6 > /tmp/tail.log 2>&1
cd $R
python profiles/timeline_rocpd.py /tmp/prof_tail 14 0 | grep -E "copyBuffer|zeta_reduce|accumulate_kernel|^#" | tail -40 | cut -c1-120
ICICLE_SNARK_TRACE_HOST=1 python scratch/file_trace.py 2>&1 | grep -B16 "call 6" | grep "host\]" | tail -16
