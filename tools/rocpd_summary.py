#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows:
    n = n[:70]
    print(f"{n:70s} {c:6d} {s/1e6:10.3f} {a/1e6:9.4f} {mn/1e6:9.4f} {mx/1e6:9.4f} {100*s/tot:6.2f}")
extra = [c for c in cols if c.lower() in ("vgpr_count", "arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size",
                                           "lds_block_size", "scratch_size", "workgroup_size", "grid_size")]
if extra:
    print()
    q = f"select {name_col}, " + ", ".join(f"max({c})" for c in extra) + f" from kernels group by {name_col}"
    print("kernel".ljust(50), *extra)
    for r in db.execute(q):
        print(r[0][:50].ljust(50), *r[1:])
