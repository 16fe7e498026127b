# three separate PMC passes over scratch/msm_only.py g1 21 (bn254_msm, L = 2^21); summary → gpurun_out/pmc_msm_g1.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/scratch/msm_only.py g1 21 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/scratch/msm_only.py g1 21 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcC -- python3 $R/scratch/msm_only.py g1 21 > /dev/null 2>&1
cd $R
python3 - > gpurun_out/pmc_msm_g1.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("pmcA", "pmcB", "pmcC"):
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('isnark::', '').replace('void ', '')[:70]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for fn in glob.glob(f'gpurun_out/{d}/**/*kernel_trace.csv', recursive=True):
        if d != "pmcA": continue
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('isnark::', '').replace('void ', '')[:70]
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
    if k.startswith('msm_'):
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:24s} {sum(v)/len(v):.6g}")
        if dur[k]: print(f"    {'dur_us':24s} {sum(dur[k])/len(dur[k]):.6g}")
PY
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC
cat gpurun_out/pmc_msm_g1.txt | head -60
