#!/bin/bash
# HIP API trace of a cold prove (first key of the process): which runtime call is long while a staging chunk's hipMemcpyAsync waits
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/hiptrace
rocprofv3 --hip-trace --output-format csv -d $R/gpurun_out/hiptrace -- python3 $R/scratch/cold_prove.py 1600000 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/hiptrace/**/*hip_api_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
print(len(rows), 'api calls;', rows[0].keys())
# long calls that are not waits
t0 = min(int(r['Start_Timestamp']) for r in rows)
waits = ('hipEventSynchronize', 'hipStreamSynchronize', 'hipDeviceSynchronize', 'hipStreamWaitEvent')
long = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 3_000_000 and r['Function'] not in waits]
long.sort(key=lambda r: int(r['Start_Timestamp']))
for r in long[-120:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e6:10.2f} ms  +{(e - s) / 1e6:7.2f} ms  tid {r['Thread_Id']}  {r['Function']}")
PY
rm -rf gpurun_out/hiptrace
