#!/usr/bin/env python3
"""HBM bytes per launch of the pass-B scan kernel from two rocprofv3 PMC summaries (FETCH_SIZE, WRITE_SIZE,
collected in separate passes) -> the JSON bench.py reads as roofline.traffic.
FETCH_SIZE is doubled: gfx950 reports half the bytes of a wide coalesced stream (MI355X_MICROARCH.md)."""
import json
import re
import sys
import time


KERNEL = r"_Z\d+k_scan\w*?ELi1EEv5ScanK\S*"  # k_scan / k_scan_wide, MODE 1 (pass B); --kernel REGEX names another (e.g. k_direct_topk)
if "--kernel" in sys.argv:
    i = sys.argv.index("--kernel")
    KERNEL = sys.argv[i + 1]
    del sys.argv[i:i + 2]


def per_launch(md, counter):
    """sum over the per-XCD/SE slices of one dispatch: avg per slice * slices / launches"""
    best = None
    for line in open(md):
        m = re.match(r"\| `(" + KERNEL + r") grid=(\d+)` \| (\w+) \| (\d+) \| ([\d.e+]+) \| ([\d.e+]+) \|", line)
        if m and m.group(3) == counter:
            total = float(m.group(6))
            if best is None or total > best[1]:
                best = (m.group(1), total, int(m.group(4)))
    return best


def launches(md, kernel):
    for line in open(md):
        if line.startswith("| `" + kernel + "`"):
            return int(line.split("|")[2])
    return None


fetch_md, write_md, bench_json = sys.argv[1:4]
f = per_launch(fetch_md, "FETCH_SIZE")
w = per_launch(write_md, "WRITE_SIZE")
nf, nw = launches(fetch_md, f[0]), launches(write_md, w[0])
j = json.load(open(bench_json))
cfg = j["config"]
fetch_kb = f[1] / nf
write_kb = w[1] / nw
out = {
    "rows": cfg["rows"], "dim": cfg["dim"], "dtype": j["dtype"], "batch": cfg["batch"], "kernel": f[0],
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb, "launches_averaged": [nf, nw],
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
              "(gfx950 reports half the bytes of a wide coalesced stream); WRITE_SIZE as reported",
    "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_launch"],
    "collected": time.strftime("%Y-%m-%d", time.gmtime()),
}
print(json.dumps(out, indent=1))
