cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmcN -- python3 $R/scratch/ntt_only.py 21 2>&1 | grep "ntt logn"
cd $R
python3 - <<'PY'
import csv, glob, collections
import sys
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for fn in glob.glob('gpurun_out/pmcN/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:50] + '|lds' + str(r.get('LDS_Block_Size', '?'))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for fn in glob.glob('gpurun_out/pmcN/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r['Kernel_Name'][:50] + '|lds' + str(r.get('LDS_Block_Size', '?'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(list(csv.DictReader(open(glob.glob('gpurun_out/pmcN/**/*counter_collection.csv', recursive=True)[0])).fieldnames))
for k, d in agg.items():
    if 'ntt_pass' in k:
        print(k)
        for c, v in sorted(d.items()): print(f"    {c:24s} {sum(v)/len(v):.6g}  n={len(v)}")
for k, v in dur.items():
    if 'ntt_pass' in k: print(k, 'dur_us mean', sum(v)/len(v), 'min', min(v), 'n', len(v))
PY
rm -rf gpurun_out/pmcN
