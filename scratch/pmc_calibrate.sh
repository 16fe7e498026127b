#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the calibration probes (known byte counts) → gpurun_out/r02_pmc_calibration.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/cal_$c -- python3 $R/scratch/pmc_probe.py > $R/gpurun_out/cal_$c.log 2>&1
done
cd $R
python3 - > gpurun_out/r02_pmc_calibration.txt <<'PY'
import csv, glob, collections
known = [int(x) for x in open('gpurun_out/cal_FETCH_SIZE.log').read().split('KNOWN')[1].split()[:5]]
names = ["64-byte random gathers (2 GiB table)", "128-byte random gathers", "coalesced 16 B/lane streaming read", "scattered 4-byte stores", "coalesced 4-byte stores"]
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = collections.OrderedDict()
    for fn in glob.glob(f'gpurun_out/cal_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'probe_' in r['Kernel_Name'] and r['Counter_Name'] == c:
                rows[int(r['Dispatch_Id'])] = rows.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
    val[c] = [v for _, v in sorted(rows.items())]
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over csrc/microbench.hip's probes (scratch/pmc_calibrate.sh), MI355X; counters in KB")
for i, n in enumerate(names):
    f, w = val["FETCH_SIZE"][i] * 1024, val["WRITE_SIZE"][i] * 1024
    print(f"{n:40s} known {known[i] / 1e6:9.1f} MB   FETCH_SIZE {f / 1e6:9.1f} MB (x{f / known[i]:.3f})   WRITE_SIZE {w / 1e6:9.1f} MB (x{w / known[i]:.3f})")
PY
rm -rf gpurun_out/cal_FETCH_SIZE gpurun_out/cal_WRITE_SIZE
cat gpurun_out/r02_pmc_calibration.txt
