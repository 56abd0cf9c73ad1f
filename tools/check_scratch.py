"""Lists every k_scan / k_scan_wide instance whose code object uses scratch (.private_segment_fixed_size > 0).

The scan kernels index their register-resident query fragments with compile-time loop indices; when LLVM declines an unroll the
array moves to scratch and the pass runs at half speed without any functional symptom (round 3: f32 768-d, <= 64 queries).
Instances at the widest row pitches (512 VGPRs) spill a few dwords outside the hot loop: those are listed too, with their size.
Usage: python tools/check_scratch.py   (compiles the scan translation units to assembly with the build's flags: a few minutes)"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panoptikon_amd import build as B  # noqa: E402

srcs = [s for s in B.SOURCES if s.startswith("pvs_scan_") and s.endswith(".hip")]
tmp = tempfile.mkdtemp()


def asm(src):
    out = os.path.join(tmp, src + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "panoptikon_amd", "csrc"),
           *B.HIPFLAGS, "--cuda-device-only", "-S", os.path.join(ROOT, "panoptikon_amd", "csrc", src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return src, open(out).read()


bad = 0
with ThreadPoolExecutor(8) as ex:
    for src, text in ex.map(asm, srcs):
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
            name, priv, vg = m.group(1), int(m.group(2)), int(m.group(3))
            if priv:
                kind = "spills at 512 VGPRs" if vg >= 512 or priv < 64 else "REGISTER ARRAY IN SCRATCH"
                bad += kind.startswith("REG")
                print(f"{src}: {name}: {priv} B of scratch, {vg} VGPRs ({kind})")
print("instances with a register array in scratch:", bad)
sys.exit(1 if bad else 0)
