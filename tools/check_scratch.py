"""Lists every k_scan / k_scan_wide instance whose code object uses scratch (.private_segment_fixed_size > 0) and FAILS on any
instance with more than 64 B of it that is not in the table below — whatever its VGPR count.

The scan kernels index their register-resident query fragments with compile-time loop indices; when LLVM declines an unroll the
array moves to scratch and the pass runs at half speed without any functional symptom (round 3: f32 768-d, <= 64 queries: 784 B per
lane).  Up to 64 B is a handful of spilled scalars outside the hot loop.  Beyond that an instance must be KNOWN: the table names the
instances that spill more at the 512-VGPR ceiling (the widest row pitches: 12 k-slabs of int8 / f16, 24 of f32, where the query
fragments alone take 384 VGPRs), the most they may spill, and what it costs — measured with tools/r4_spill_cost.sh against the next
narrower pitch, which does not spill (profiles/r04_spill_cost.txt).  A new spiller, or a known one that grew, is an error.
Usage: python tools/check_scratch.py   (compiles the scan translation units to assembly with the build's flags: a few minutes)"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptikon_amd import build as B  # noqa: E402

LIMIT = 64
# (dtype, k-slabs) of the instance families allowed to spill more than LIMIT bytes, the ceiling in bytes, and the measured cost.
# k_scan<DT, KSLABS, QG, METRIC, MODE>: DT 0 = f32, 1 = f16, 2 = int8.
KNOWN = {
    (2, 12): (800, "int8 3072-B rows: scan 0.778 / 0.779 / 0.640 of HBM at 1 / 32 / 128 queries against 0.784 / 0.784 / 0.635 for 2048-B rows (no spill): free"),
    (1, 12): (256, "f16 1536-d rows: 0.793 / 0.789 / 0.685 against 0.779 / 0.778 / 0.662 for 1024-d rows: free"),
    (0, 24): (256, "f32 1536-d rows: 0.740 / 0.739 / 0.609 against 0.762 / 0.759 / 0.715 for 1024-d rows: -3 % at 1-32 queries, -15 % at 128 queries"),
}

srcs = [s for s in B.SOURCES if s.startswith("pvs_scan_") and s.endswith(".hip")]
tmp = tempfile.mkdtemp()


def asm(src):
    out = os.path.join(tmp, src + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "panoptikon_amd", "csrc"),
           *B.HIPFLAGS, "--cuda-device-only", "-S", os.path.join(ROOT, "panoptikon_amd", "csrc", src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return src, open(out).read()


bad = 0
rows = []
with ThreadPoolExecutor(8) as ex:
    for src, text in ex.map(asm, srcs):
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
            name, priv, vg = m.group(1), int(m.group(2)), int(m.group(3))
            if not priv:
                continue
            t = re.match(r"_Z6k_scanILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EE", name)
            fam = (int(t.group(1)), int(t.group(2))) if t else None
            if priv <= LIMIT:
                kind = "a few spilled scalars"
            elif fam in KNOWN and priv <= KNOWN[fam][0]:
                kind = "known: " + KNOWN[fam][1]
            else:
                kind = "NOT IN THE TABLE: a register array in scratch, or a known spiller that grew"
                bad += 1
            rows.append((priv, f"{src}: {name}: {priv} B of scratch, {vg} VGPRs ({kind})"))
for _, line in sorted(rows, reverse=True):
    print(line)
print(f"instances with scratch: {len(rows)}; over {LIMIT} B and not in the table: {bad}")
sys.exit(1 if bad else 0)
