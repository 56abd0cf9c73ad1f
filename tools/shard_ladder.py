#!/usr/bin/env python3
"""The shard ladder: the headline workload (BASELINE configs[2]: 10M x 768 int8, batches of 128, k = 100 — or --config 3) at the
shard size of N = 1, 2, 4, 8 GPUs, each as a ONE-GPU run of rows/N rows through `bench.py --force-comm` (a 1-rank RCCL
communicator in the path: local search -> ncclAllGather of the page record -> merge kernel, the exchange span measured by HIP
events).  What an N-GPU job costs per step is the slowest rank's step at that shard size plus an exchange that moves N records
instead of one (<= 0.31 MB per rank: latency-bound) — so rows/N-per-GPU step time is the PROJECTED step of the N-GPU job, and
batch / step its projected whole-job q/s.  Not a measurement of N GPUs: the driver's SCALE run is.

    python tools/shard_ladder.py [--config 2|3] [--out gpurun_out/ladder]   (GPU box)

Writes one bench line per rung (bench_r06_ladder_<cfg>_n<N>.json) and the summary bench.py replays into its default line
(projected_scaling_latest.json); copy both to profiles/."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {2: (10_000_000, 768, "i8", 128), 3: (100_000_000, 768, "i8", 256)}

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2, choices=[2, 3])
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ladder"))
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
rows, dim, dtype, batch = CFG[a.config]
os.makedirs(a.out, exist_ok=True)
by_n = {}
for n in (1, 2, 4, 8):
    shard = (rows + n - 1) // n
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(a.config), "--rows", str(shard), "--force-comm", "--steps", str(a.steps), "--warmup", "5",
           "--no-secondary", "--no-cpu-baseline", "--no-peaks", "--check-queries", "2"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        print(r.stderr[-2000:], file=sys.stderr)
        raise SystemExit(f"rung n={n} failed")
    line = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    open(os.path.join(a.out, f"bench_r06_ladder_cfg{a.config}_n{n}.json"), "w").write(line + "\n")
    j = json.loads(line)
    rl = j["roofline"]
    by_n[str(n)] = {"rows_per_gpu": shard, "ms_per_step": j["ms_per_step"], "projected_qps": round(batch / (j["ms_per_step"] * 1e-3), 1),
                    "scan_ms": rl["avg_launch_ms"], "scan_hbm_frac": rl["frac"], "exchange_ms": rl.get("exchange_avg_ms"),
                    "parity_ok": j.get("parity", {}).get("ids_and_distances_bit_exact"), "exchange": j["config"]["exchange"]}
    print(n, by_n[str(n)], flush=True)
base = by_n["1"]["projected_qps"]
for n in by_n:
    by_n[n]["speedup_vs_1"] = round(by_n[n]["projected_qps"] / base, 3)
    by_n[n]["efficiency"] = round(by_n[n]["projected_qps"] / base / int(n), 3)
out = {"rows": rows, "dim": dim, "dtype": dtype, "batch": batch, "collected": time.strftime("%Y-%m-%d", time.gmtime()),
       "how": "bench.py --force-comm at rows/N per GPU on ONE GPU (1-rank RCCL exchange in the step); projected, not measured on N GPUs",
       "by_n_gpus": by_n}
name = "projected_scaling_latest.json" if a.config == 2 else f"projected_scaling_cfg{a.config}.json"
json.dump(out, open(os.path.join(a.out, name), "w"), indent=1)
print(json.dumps(out))
