"""prints the key fields of bench.py's JSON line(s) read from stdin: tag, q/s, step ms, scan ms, pass A ms, pass C ms, parity"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(tag, d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("sample_pass_avg_ms"), r.get("finalize_avg_ms"), (d.get("parity") or {}).get("ids_and_distances_bit_exact"))
