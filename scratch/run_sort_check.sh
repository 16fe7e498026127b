python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_sort -- python3 $GRAFT_REPO_ROOT/scratch/pmc_child.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $(ls /tmp/prof_sort/*/*_results.db | head -1) 2>&1 | grep -E "sort2|calls" | cut -c1-150
python scratch/paths_loop.py 30 2>/dev/null | tail -1
