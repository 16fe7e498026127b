#!/usr/bin/env python3
"""Per-kernel summary (count, total, mean, min, max, share) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this image).
usage: summarize_rocpd.py results.db [skip_first_n_dispatches_per_kernel]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    rows = db.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
    stats = {}
    for name, st, en in rows:
        name = re.sub(r"\[clone.*", "", name).strip()
        stats.setdefault(name, []).append(en - st)
    total = sum(sum(v) for v in stats.values())
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'mean_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:90]:90s} {len(v):6d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.1f} {min(v) / 1e3:10.1f} {max(v) / 1e3:10.1f} {100 * sum(v) / total:6.2f}")
    print(f"total kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
