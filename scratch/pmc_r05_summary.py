"""summary of the three PMC passes of scratch/pmc_r05.sh (gpurun_out/pmcA|B|C): per kernel family, the launches of the last prove"""
import collections, csv, glob, sys

FAMILIES = [("msm_accumulate_kernel", "Fq2Ops", "msm_accumulate<G2>"), ("msm_accumulate_kernel", "FqOps", "msm_accumulate<G1>"),
            ("msm_zeta_reduce_kernel", "Fq2Ops", "msm_zeta_reduce<G2>"), ("msm_zeta_reduce_kernel", "FqOps", "msm_zeta_reduce<G1>"),
            ("ntt_pass29_kernel", "", "ntt_pass29"), ("qap_spmv_kernel", "", "qap_spmv"),
            ("sort2_tile_hist", "", "sort2_tile_hist"), ("sort2_col_sum", "", "sort2_col_sum"), ("sort2_order_scan", "", "sort2_order_scan"), ("sort2_order_scatter", "", "sort2_order_scatter"),
            ("sort2_col_apply", "", "sort2_col_apply"), ("sort2_tile_partition", "", "sort2_tile_partition"), ("sort2_chunk_hist", "", "sort2_chunk_hist"),
            ("sort2_bucket_scan", "", "sort2_bucket_scan"), ("sort2_chunk_place", "", "sort2_chunk_place"),
            ("msm_order_", "", "msm_order_*"), ("msm_accumulate_large", "", "msm_accumulate_large"), ("msm_combine_large", "", "msm_combine_large")]


def family(name):
    for a, b, label in FAMILIES:
        if a in name and (not b or b in name):
            return label
    return None


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch id -> counter -> sum over XCD / SE rows
    meta, rows_per = {}, collections.Counter()
    for fn in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            fam = family(r["Kernel_Name"])
            if not fam:
                continue
            did = int(r["Dispatch_Id"])
            per[did][r["Counter_Name"]] += float(r["Counter_Value"])
            rows_per[(did, r["Counter_Name"])] += 1
            meta[did] = (fam, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0))
    return per, meta, rows_per


A, metaA, rowsA = load("pmcA")
B, metaB, _ = load("pmcB")
Cc, metaC, _ = load("pmcC")
if not metaA:
    sys.exit("no counter_collection.csv of pass A")
print("# rocprofv3 --kernel-trace --pmc <counters> -- python3 scratch/pmc_child.py   (three passes: SQ + GRBM | FETCH_SIZE | WRITE_SIZE)")
print("# benchmark/1600k, four proves from a host buffer; per family the launches of the LAST prove in dispatch order — its witness is split (HISTORY.md §4): sort2_* #0 = digit sort of the HEAD,")
print("# #1 = of the TAIL (bench.py's roofline.scatter.traffic = #0 + #1), #2 = of the H scalars; msm_accumulate<G1> #0-2 / <G2> #0 = the head's, the next four the tail's (into the same buckets), the last G1 one H's.")
print("# Under --pmc the kernels of a prove run serialised.")
print("# SQ_* are sums over all XCDs / shader engines (rows per dispatch and counter: %s); *_CYCLES of SQ count quad-cycles" % sorted(set(rowsA.values())))
print("# bytes: FETCH_SIZE, WRITE_SIZE in KB as reported; hbm_MB = (2 x FETCH + WRITE) / 1e3 (gfx950: FETCH_SIZE counts half of a coalesced stream), raw_MB = (FETCH + WRITE) / 1e3")


def last_prove(meta):
    # dispatches of the last prove: behind the last qap_spmv
    ids = sorted(meta)
    sp = [i for i in ids if meta[i][0] == "qap_spmv"]
    lo = sp[-1] if sp else ids[0]
    # the witness digit sort is enqueued before the spmv: take the sort kernels behind the previous prove's last reduction
    prev = [i for i in ids if i < lo and meta[i][0].startswith("msm_zeta_reduce")]
    start = (prev[-1] + 1) if prev else ids[0]
    return [i for i in ids if i >= start]


selA, selB, selC = last_prove(metaA), last_prove(metaB), last_prove(metaC)
byfamB, byfamC = collections.defaultdict(list), collections.defaultdict(list)
for i in selB:
    byfamB[metaB[i][0]].append(B[i].get("FETCH_SIZE", 0.0))
for i in selC:
    byfamC[metaC[i][0]].append(Cc[i].get("WRITE_SIZE", 0.0))
seen = collections.Counter()
hdr = f"{'kernel':24s} {'#':>2s} {'us':>9s} {'grid':>9s} {'waves':>9s} {'insts_valu':>12s} {'act_valu':>12s} {'act_any':>12s} {'wait_inst':>12s} {'busy_cyc':>12s} {'wave_cyc':>13s} {'gui_active':>12s} {'valu_busy':>9s} {'fetch_KB':>10s} {'write_KB':>10s} {'hbm_MB':>8s} {'raw_MB':>8s}"
print(hdr)
for i in selA:
    fam, us, grid, wg = metaA[i]
    k = seen[fam]
    seen[fam] += 1
    c = A[i]
    f = byfamB[fam][k] if k < len(byfamB[fam]) else float("nan")
    w = byfamC[fam][k] if k < len(byfamC[fam]) else float("nan")
    # VALU busy as rocprof's gfx94x formula: 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE, with GRBM_GUI_ACTIVE averaged over its rows (one per XCD)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / max(1, rowsA.get((i, "GRBM_GUI_ACTIVE"), 1))
    vb = 100.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024 / gui if gui else float("nan")
    print(f"{fam:24s} {k:2d} {us:9.1f} {grid:9d} {c.get('SQ_WAVES', 0):9.0f} {c.get('SQ_INSTS_VALU', 0):12.4g} {c.get('SQ_ACTIVE_INST_VALU', 0):12.4g} {c.get('SQ_ACTIVE_INST_ANY', 0):12.4g} "
          f"{c.get('SQ_WAIT_INST_ANY', 0):12.4g} {c.get('SQ_BUSY_CYCLES', 0):12.4g} {c.get('SQ_WAVE_CYCLES', 0):13.5g} {gui:12.4g} {vb:8.1f}% {f:10.0f} {w:10.0f} {(2 * f + w) / 1e3:8.1f} {(f + w) / 1e3:8.1f}")
