"""bench.py's PMC bookkeeping on canned counter_collection.csv rows (no GPU, no profiler): the digit sorts of a prove are found
by their launch pattern — two per prove (witness, H) or three when the witness is split into a head and a tail — and the
witness pass is reported as head + tail, H separately (round-4 verdict: with three sorts the round-4 parser dropped every one)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")

LDS_SORT = ["sort2_tile_hist_kernel", "sort2_col_sum_kernel", "sort2_col_apply_kernel", "sort2_tile_partition_kernel", "sort2_chunk_hist_kernel",
            "sort2_bucket_scan_kernel", "sort2_chunk_place_kernel", "sort2_order_scan_kernel", "sort2_order_scatter_kernel"]
LDS_SORT_R4 = ["msm_zero_kernel", "sort2_tile_hist_kernel", "sort2_col_sum_kernel", "sort2_col_base_kernel", "sort2_col_apply_kernel",
               "sort2_tile_partition_kernel", "sort2_chunk_hist_kernel", "sort2_bucket_scan_kernel", "sort2_chunk_place_kernel",
               "msm_order_hist_kernel", "msm_scan_sums_kernel", "msm_scan_top_kernel", "msm_scan_apply_kernel", "msm_order_scatter_kernel"]
CLASSIC_SORT = ["msm_zero_kernel", "msm_zero_kernel", "msm_coarse_hist_kernel", "msm_part_scan_kernel", "msm_partition_kernel", "msm_fine_count_kernel",
                "msm_fine_place_kernel", "msm_order_hist_kernel", "msm_scan_sums_kernel", "msm_scan_top_kernel", "msm_scan_apply_kernel", "msm_order_scatter_kernel"]
ACC_G1 = "void (anonymous namespace)::msm_accumulate_kernel<bn254::G1, false, false>(bn254::G1::A const*, unsigned int const*)"
ACC_G2 = "void (anonymous namespace)::msm_accumulate_kernel<bn254::G2, false, false>(bn254::G2::A const*, unsigned int const*)"


def canned(counter, sorts_per_prove, value_of, proves=3, xcds=8, sort_kernels=LDS_SORT):
    """rows of one PMC pass: per prove [head sort, 4 head accumulations,] witness sort, other kernels, 4 accumulations, H sort,
    H accumulation — every dispatch as `xcds` per-XCD rows whose values sum to value_of(tag, kernel, prove)"""
    rows, did = [], [0]

    def dispatch(name, tag, prove):
        did[0] += 1
        v = value_of(tag, name, prove)
        for x in range(xcds):
            rows.append({"Dispatch_Id": str(did[0]), "Kernel_Name": f"{name}(args…)" if "(" not in name else name, "Counter_Name": counter,
                         "Counter_Value": str(v / xcds)})
            rows.append({"Dispatch_Id": str(did[0]), "Kernel_Name": name, "Counter_Name": "SOME_OTHER_COUNTER", "Counter_Value": "5"})
    for p in range(proves):
        if sorts_per_prove == 3:
            for k in sort_kernels:
                dispatch(k, "head", p)
            for a in (ACC_G2, ACC_G1, ACC_G1, ACC_G1):
                dispatch(a, "acc_head", p)
        for k in sort_kernels:
            dispatch(k, "tail", p)
        dispatch("qap_spmv_kernel", "other", p)
        dispatch("ntt_pass29_kernel", "other", p)
        dispatch(ACC_G2, "acc", p)
        for k in sort_kernels:
            dispatch(k, "hsort", p)
        for a in (ACC_G1, ACC_G1, ACC_G1):
            dispatch(a, "acc", p)
        dispatch(ACC_G1, "acc_h", p)
        dispatch("msm_zeta_reduce_kernel", "other", p)
    return rows


def values(scale):
    base = {"head": 10.0, "tail": 100.0, "hsort": 1000.0, "acc_head": 50.0, "acc": 500.0, "acc_h": 2000.0, "other": 7.0}
    return lambda tag, name, prove: scale * base[tag] * (1 + prove) + (0.5 if "zero" in name else 0.0)


@pytest.mark.parametrize("sorts_per_prove", [2, 3])
@pytest.mark.parametrize("kernels", [LDS_SORT, CLASSIC_SORT, LDS_SORT_R4])
def test_sorts_of_a_prove_are_grouped_by_instance(sorts_per_prove, kernels):
    fetch = canned("FETCH_SIZE", sorts_per_prove, values(1.0), sort_kernels=kernels)
    write = canned("WRITE_SIZE", sorts_per_prove, values(0.25), sort_kernels=kernels)
    s = bench.pmc_summary(fetch, write)
    assert s["sort_instances_per_prove"] == sorts_per_prove
    nz = sum(1 for k in kernels if "zero" in k)
    first = kernels[0]
    last = 3                                           # values of the last prove are 3 × base
    per_sort = lambda base, scale: scale * base * last * len(kernels) + 0.5 * nz
    f_w = per_sort(100.0, 1.0) + (per_sort(10.0, 1.0) if sorts_per_prove == 3 else 0.0)
    w_w = per_sort(100.0, 0.25) + (per_sort(10.0, 0.25) if sorts_per_prove == 3 else 0.0)
    assert s["sort_w"] == pytest.approx(f_w * 1024 * 2 + w_w * 1024)              # witness pass = head + tail, FETCH ×2 (streams)
    assert s["sort_h"] == pytest.approx(per_sort(1000.0, 1.0) * 1024 * 2 + per_sort(1000.0, 0.25) * 1024)
    det = s["detail"]
    assert set(det) - {"acc_h"} == set(kernels)
    runs = sorts_per_prove - 1
    assert det[first]["launches"] == (nz if "zero" in first else 1) * runs and det[kernels[-1]]["launches"] == runs
    # the H accumulation: the G1 launch with the most bytes among the last four, FETCH_SIZE / 1.494 (64-byte gathers) + WRITE_SIZE
    assert det["acc_h"] == {"FETCH_SIZE_KB": 2000.0 * 3, "WRITE_SIZE_KB": 500.0 * 3}
    assert s["acc_h"] == pytest.approx(6000.0 * 1024 / 1.494 + 1500.0 * 1024)
    assert s["acc_h_x2"] == pytest.approx(6000.0 * 1024 * 2 + 1500.0 * 1024) and s["acc_h_raw"] == pytest.approx(7500.0 * 1024)


def test_a_stream_that_does_not_divide_gives_no_sort_traffic_but_keeps_the_rest():
    fetch = canned("FETCH_SIZE", 3, values(1.0))
    write = canned("WRITE_SIZE", 3, values(0.25))
    # drop one whole sort from the first prove: 8 sorts over 3 proves
    cut = lambda rows: [r for r in rows if not (int(r["Dispatch_Id"]) <= len(LDS_SORT))]
    s = bench.pmc_summary(cut(fetch), cut(write))
    assert "sort_w" not in s and "acc_h" in s
    # passes whose kernel sequences disagree (a pass that lost a dispatch) are not summed against each other
    last_sort = max(int(r["Dispatch_Id"]) for r in write if bench.sort_family(r["Kernel_Name"]))
    s2 = bench.pmc_summary(fetch, [r for r in write if int(r["Dispatch_Id"]) != last_sort])
    assert "sort_w" not in s2


def test_issue_counters_follow_the_same_launch():
    fetch, write = canned("FETCH_SIZE", 3, values(1.0)), canned("WRITE_SIZE", 3, values(0.25))
    sq = canned("SQ_INSTS_VALU", 3, values(1e6)) + canned("GRBM_GUI_ACTIVE", 3, values(1e3))
    s = bench.pmc_summary(fetch, write, sq)
    assert s["acc_h_valu"] == {"SQ_INSTS_VALU": pytest.approx(2000.0 * 3e6), "GRBM_GUI_ACTIVE": pytest.approx(2000.0 * 3e3)}
