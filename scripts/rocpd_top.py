#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 rocpd database (used on the GPU box, where the raw db is too
large to copy back): name, calls, total ms, average us, percent."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
print("name,calls,total_ms,avg_us,percent")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f'"{r[0][:100]}",{r[1]},{r[2] / 1e3:.3f},{r[3]:.2f},{r[4]:.2f}')
print(f"# all kernels: {sum(r[2] for r in rows) / 1e3:.2f} ms in {sum(r[1] for r in rows)} launches")
