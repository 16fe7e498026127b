# PMC passes over four proves of benchmark/1600k (scratch/spmv_only.py): bucket-accumulation launches in dispatch order
# (per prove: A, B1, C — 1.6 M scalars — then H — 2^21 scalars; the G2 one is B2); summary → gpurun_out/pmc_prove.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcC -- python3 $R/scratch/spmv_only.py > /dev/null 2>&1
cd $R
python3 - > gpurun_out/pmc_prove.txt <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for d in ("pmcA", "pmcB", "pmcC"):
    for fn in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if 'msm_accumulate_kernel' not in r['Kernel_Name']: continue
            key = 'G2' if 'Fq2Ops' in r['Kernel_Name'] else 'G1'
            per[(key, r['Counter_Name'])].append((int(r['Dispatch_Id']), float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
        for (key, cname), v in per.items():
            v.sort()
            # sum the per-XCD / per-SE rows of one dispatch
            agg = collections.OrderedDict()
            for did, val, dur in v:
                a = agg.setdefault(did, [0.0, dur]); a[0] += val
            for n, (did, (val, dur)) in enumerate(agg.items()):
                rows[(key, n)][cname] = val
                if d == "pmcA": rows[(key, n)]['dur_us'] = dur
print("# rocprofv3 --pmc ... -- python3 scratch/spmv_only.py (4 proves of benchmark/1600k, table mode c=20 W=13); bucket-accumulation launches in dispatch order")
for (key, n), d in sorted(rows.items()):
    which = ("A/B1/C (L=1600002)" if n % 4 != 3 else "H (L=2097152)") if key == 'G1' else "B2 (L=1600002)"
    print(f"{key} launch {n:2d} {which:20s} " + "  ".join(f"{c}={v:.6g}" for c, v in sorted(d.items())))
PY
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC
cat gpurun_out/pmc_prove.txt
