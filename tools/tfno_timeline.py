"""Per-kernel timeline of the last TFNO step in a rocprofv3 sqlite result (tools/gpu_*.sh): name, start offset, duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start"))
idx = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel")]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = rows[a][1]
agg = {}
for r in rows[a:b]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.2f}  {r[0][:60]:60s} wgs={r[3] // max(r[4], 1)} lds={r[5]}")
    k = r[0][:40]
    agg[k] = agg.get(k, 0.0) + (r[2] - r[1]) / 1e3
print("span us", (rows[b - 1][2] - t0) / 1e3, " kernel sum us", sum(agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {v:8.1f}  {k}")
