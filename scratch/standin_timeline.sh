# one-prove timeline of a stand-in workload (default aadhaar_standin): rocprofv3 kernel trace of scratch/standin_loop.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof_standin -- python3 $R/scratch/standin_loop.py 8 > /tmp/standin.log 2>&1
cd $R
tail -6 /tmp/standin.log
python profiles/timeline_rocpd.py /tmp/prof_standin 5
