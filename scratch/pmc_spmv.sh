cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcS -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcS2 -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcS3 -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmcS", "pmcS2", "pmcS3"):
  for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:60] + ' vgpr=' + r['VGPR_Count']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
    if 'qap_' in k or 'msm_hist' in k or 'ntt_pass' in k:
        print(k)
        for c, v in sorted(d.items()): print(f"    {c:24s} {sum(v)/len(v):.6g}  n={len(v)}")
PY
rm -rf gpurun_out/pmcS gpurun_out/pmcS2 gpurun_out/pmcS3
