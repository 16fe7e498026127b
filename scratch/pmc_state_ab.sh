#!/bin/bash
# does a rocprofv3 --pmc pass leave the box in a state in which later (unprofiled) runs see slower sorts / table builds?
run() { python bench.py --steps 10 --warmup 2 --no-pmc --no-dropin --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$1: file', round(d['ms_per_step'],2), 'tables alone', round(c['cold_path']['cold_tables_build_ms']), 'witness sort', round(r['scatter']['launch_ms'],2), 'acc', round(r['launch_ms'],2), 'scattered', c['box_access_gbps']['scattered_store_gbps'])"; }
run "before-1"
run "before-2"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_ab -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child > /dev/null 2>&1; rm -rf /tmp/pmc_ab )
echo "(one rocprofv3 --pmc FETCH_SIZE pass over bench.py --pmc-child done)"
run "after-1"
run "after-2"
sleep 20
run "after-3 (20 s later)"
