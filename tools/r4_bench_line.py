"""Prints the headline and secondary numbers of a bench.py JSON line.  Usage: python tools/r4_bench_line.py file.json"""
import json, sys
r = json.load(open(sys.argv[1]))
rf = r["roofline"]
print(f"headline {r['config']['workload']}: {r['value']} q/s, step {r['ms_per_step']} ms, scan {rf['avg_launch_ms']} ms = {rf['frac']} of HBM, mfma {rf['mfma']['frac']}, "
      f"sample {rf.get('sample_pass_avg_ms')} finalize {rf.get('finalize_avg_ms')}, parity {r.get('parity', {}).get('ids_and_distances_bit_exact')}, recall {r.get('recall_at_k')}")
for s in r.get("secondary", []):
    if "error" in s:
        print("secondary error:", s["error"])
        continue
    print(f"secondary {s['config']['workload']}: {s['value']} q/s, step {s['ms_per_step']} ms, scan {s['roofline']['avg_launch_ms']} ms = {s['roofline']['frac']} of HBM, "
          f"mfma {s['roofline']['mfma']['frac']}, {s.get('parity')}")
if "cpu_baseline" in r:
    print("cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["unit"], "cores", r["cpu_baseline"]["cores"])
