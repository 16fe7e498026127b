#!/usr/bin/env python3
"""Timeline of ONE prove out of a rocprofv3 --kernel-trace [--memory-copy-trace] run (rocpd sqlite): start/end of every dispatch
of one prove relative to its first event, grouped by stream (queue), plus the busy-union of the GPU; with a memory-copy trace the
host→device copies of the witness upload are listed beside the kernels (merged into runs per engine queue).
usage: timeline_rocpd.py <dir with *_results.db> [prove index, default 4; negative = from the end] [min dispatch ns to list, default 30000]"""
import glob
import sqlite3
import sys


def main():
    db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in con.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = con.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    # a prove is anchored at its qap_spmv_kernel; its first kernel is the one behind the previous prove's last bucket reduction
    # (the digit sort of the witness — of the witness HEAD, milliseconds before the spmv, when the head / tail split is on)
    starts = [i for i, r in enumerate(rows) if "qap_spmv" in r[0]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    if k < 0:
        k = len(starts) + k   # counted from the end (bench.py proves dozens of times before its timed loop since round 5)
    k = max(0, min(k, len(starts) - 2))

    def first_of(i_spmv):
        lo = i_spmv
        while lo > 0 and "reduce" not in rows[lo - 1][0] and rows[i_spmv][1] - rows[lo - 1][1] < 8_000_000:
            lo -= 1
        return lo
    lo, hi = first_of(starts[k]), first_of(starts[k + 1])
    sel = rows[lo:hi]
    copies = []
    mc = [t for t in tables if t.startswith("rocpd_memory_copy")]
    if mc:
        ccols = [r[1] for r in con.execute(f"pragma table_info({mc[0]})")]
        if "start" in ccols and "end" in ccols and "size" in ccols:
            t_lo = min(r[1] for r in sel) - 4_000_000
            t_hi = max(r[2] for r in sel)
            copies = [c for c in con.execute(f"select start, end, size from {mc[0]} order by start").fetchall() if t_lo <= c[0] <= t_hi and c[2] >= (1 << 20)]
    t0 = min([r[1] for r in sel] + [c[0] for c in copies])
    print(f"# one prove: {len(sel)} dispatches, span {(max(r[2] for r in sel) - t0) / 1e6:.3f} ms (from the first copy / kernel to the last kernel)")
    if copies:
        tot = sum(c[2] for c in copies)
        print(f"# witness upload: {len(copies)} copies of >= 1 MiB, {tot / 1e6:.1f} MB, first starts {(copies[0][0] - t0) / 1e6:.3f} ms, last ends {(max(c[1] for c in copies) - t0) / 1e6:.3f} ms"
              f" ({tot / max(1, max(c[1] for c in copies) - copies[0][0]):.1f} GB/s)")
        # cumulative arrival: when 10 %, 20 %, … of the bytes had landed
        acc, marks = 0, []
        for c in sorted(copies, key=lambda c: c[1]):
            acc += c[2]
            while len(marks) < 10 and acc >= tot * (len(marks) + 1) / 10:
                marks.append((c[1] - t0) / 1e6)
        print("# bytes landed (10 % steps, ms): " + " ".join(f"{m:.2f}" for m in marks))
    print(f"{'queue':>6} {'start_ms':>9} {'end_ms':>9} {'dur_ms':>8}  kernel")
    for name, s, e, q in sel:
        short = name.split("(")[0].split("::")[-1][:60]
        tag = "G2" if "Fq2Ops" in name else ("G1" if "FqOps" in name else "")
        if (e - s) > (int(sys.argv[3]) if len(sys.argv) > 3 else 30_000):
            print(f"{q:>6} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {short} {tag}")
    ev = sorted([(s, 1) for _, s, e, _ in sel] + [(e, -1) for _, s, e, _ in sel])
    busy, depth, last = 0, 0, None
    hist = {}
    for t, d in ev:
        if depth > 0:
            busy += t - last
        if last is not None:
            hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    print(f"# GPU busy (≥1 kernel resident) {busy / 1e6:.3f} ms; time by number of concurrent kernels: " +
          ", ".join(f"{k}: {v / 1e6:.2f} ms" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
