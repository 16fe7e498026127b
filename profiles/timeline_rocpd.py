#!/usr/bin/env python3
"""Timeline of ONE prove out of a rocprofv3 --kernel-trace run (rocpd sqlite): start/end of every dispatch of
the last complete prove relative to its first kernel, grouped by stream (queue), plus the busy-union of the GPU.
usage: timeline_rocpd.py <dir with *_results.db> [prove index, default 4 = inside bench.py's timed loop] [min dispatch ns to list, default 30000]"""
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
    # a prove starts with qap_spmv_kernel; take the last complete one
    starts = [i for i, r in enumerate(rows) if "qap_spmv" in r[0]]
    # the witness sort (msm recode/hist) of the same prove is enqueued just before the spmv on another stream
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    k = min(k, len(starts) - 2)
    i0, i1 = starts[k], starts[k + 1]
    # include kernels launched shortly before the spmv (sort on stream g2)
    t_spmv = rows[i0][1]
    lo = i0
    while lo > 0 and t_spmv - rows[lo - 1][1] < 300_000 and "reduce" not in rows[lo - 1][0]:
        lo -= 1
    sel = rows[lo:i1]
    # drop kernels that belong to the next prove's early sort
    t_next = rows[i1][1]
    sel = [r for r in sel if r[1] < t_next - 300_000 or "msm_" not in r[0] or r[1] < t_next]
    t0 = min(r[1] for r in sel)
    print(f"# one prove: {len(sel)} dispatches, span {(max(r[2] for r in sel) - t0) / 1e6:.3f} ms (kernel activity only)")
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
