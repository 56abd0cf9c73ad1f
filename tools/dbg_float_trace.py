"""One traced 32-query per-item AVG call over 4M x 768 f16 rows through the certified route (pvs_debug float_certify_trace)."""
import subprocess, sys
sys.path.insert(0, "/root/repo")
import panoptikon_amd as pvs
pvs.debug_set("float_certify_trace", 1)
sys.argv = ["one_avg_float.py", sys.argv[1] if len(sys.argv) > 1 else "f16"]
exec(open("/root/repo/tools/one_avg_float.py").read())
