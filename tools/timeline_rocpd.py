#!/usr/bin/env python3
"""Kernel timeline of the last `n` dispatches of a rocprofv3 rocpd database: start offset, duration, gap to the previous
kernel's end, queue.  Usage: timeline_rocpd.py results.db [n] [out.md]"""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "d.queue_id" if "queue_id" in cols else "0"
rows = list(c.execute(f"""select d.start, d.end, s.kernel_name, {qcol} from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
rows = rows[-n:]
t0 = rows[0][0]
out = ["| start us | dur us | gap to previous end us | queue | kernel |", "|---|---|---|---|---|"]
prev_end = None
for st, en, name, q in rows:
    gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:.1f}"
    out.append(f"| {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {gap} | {q} | `{name[:60]}` |")
    prev_end = max(prev_end or en, en)
text = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text)
print(text)
