"""Builds libpvs.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m panoptikon_amd.build [--force] [-j N]

hipcc cross-compiles without a GPU.  Objects go to panoptikon_amd/csrc/build/,
the library to panoptikon_amd/libpvs.so (git-ignored; it travels to the GPU box).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libpvs.so")
LIB_SQLITE = os.path.join(HERE, "libpvs_sqlite.so")  # the SQLite loadable extension (pvs_sqlite.cpp), links libpvs.so
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = [
    "pvs_api.hip",
    "pvs_search.hip",
    "pvs_search_host.hip",
    "pvs_search_device.hip",
    "pvs_items.hip", "pvs_items_float.hip",
    "pvs_rrf_search.hip",
    "pvs_similar.hip",
    "pvs_kernels_util.hip",
    "pvs_kernels_scan.hip",
    "pvs_scan_i8.hip",
    "pvs_scan_i8_wide.hip",
    "pvs_scan_f16_small.hip",
    "pvs_scan_f16_large.hip",
    "pvs_scan_f16_xl.hip",
    "pvs_scan_i8_large.hip",
    "pvs_scan_f32_xl.hip",
    "pvs_scan_f32_small.hip",
    "pvs_scan_f32_mid.hip",
    "pvs_scan_f32_large.hip",
    "pvs_dense.hip",
    "pvs_dense_exact.hip", "pvs_dense_exact2.hip", "pvs_exact_wide.hip",
    "pvs_direct.hip",
    "pvs_direct_i8.hip",
    "pvs_direct_f16.hip",
    "pvs_direct_f32.hip",
    "pvs_select.hip",
    "pvs_groups.hip",
    "pvs_rrf.hip",
    "pvs_rrf_sharded.hip",
    "pvs_score_direct.hip",
    "pvs_sparse.hip", "pvs_rrf_device.hip",
    "pvs_comm.hip",
    "pvs_multi.hip", "pvs_multi_items.hip",
    "pvs_lifecycle.hip",
    "pvs_gate.hip",
    "pvs_microbench.hip",
    "pvs_host.cpp",
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# -pragma-unroll-threshold: the scan kernels keep a pass's query fragments in registers and index them with the (compile-time) chunk
# and step of fully unrolled loops.  LLVM declines a `#pragma unroll` whose unrolled body exceeds 16k instructions — pass B of the
# instances with 8+ k-slabs per row (f32 512-d and up, f16 1152-d and up) — and the fragment array then lives in scratch: the f32
# 768-d scan for <= 64 queries ran at 3.1 TB/s instead of 6.5 (found in round 3 through .private_segment_fixed_size of every instance)
HIPFLAGS = ["--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000"]
HOSTFLAGS = ["-ffp-contract=off"]


def _stale(src: str, obj: str) -> bool:
    """True when obj is older than the source, this script, or any header its depfile (-MD) names."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep) or os.path.getmtime(os.path.abspath(__file__)) > t:
        return True
    try:
        txt = open(dep).read().replace("\\\n", " ")
    except OSError:
        return True
    files = txt.split(":", 1)[1].split() if ":" in txt else []
    for f in [os.path.join(CSRC, src), *files]:
        if not f.startswith(("/opt/", "/usr/")) and (not os.path.exists(f) or os.path.getmtime(f) > t):
            return True
    return False


def _compile(src: str, force: bool = False) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and not _stale(src, obj):
        return obj
    dep = ["-MD", "-MF", obj[:-2] + ".d"]
    # tuning experiments: PVS_FLAGS_<stem>="-mllvm ..." adds flags to one translation unit
    extra = os.environ.get("PVS_FLAGS_" + os.path.splitext(src)[0], "").split()
    if src.endswith(".hip"):
        cmd = [HIPCC, *COMMON, *HIPFLAGS, *extra, *dep, "-c", path, "-o", obj]
    else:
        cmd = [HIPCC, *COMMON, *HOSTFLAGS, *dep, "-x", "c++", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"compile failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or min(len(SOURCES), os.cpu_count() or 4)
    before = {s: os.path.getmtime(os.path.join(OBJ, os.path.splitext(s)[0] + ".o")) if os.path.exists(os.path.join(OBJ, os.path.splitext(s)[0] + ".o")) else 0 for s in SOURCES}
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    changed = any(os.path.getmtime(o) != before[s] for s, o in zip(SOURCES, objs))
    if not changed and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(o) for o in objs):
        _build_sqlite_extension()
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    _build_sqlite_extension()
    return LIB


def _build_sqlite_extension() -> str:
    """libpvs_sqlite.so: plain C++ (no SQLite headers or library needed, see pvs_sqlite.cpp), next to libpvs.so."""
    src = os.path.join(CSRC, "pvs_sqlite.cpp")
    deps = [src, os.path.join(ROOT, "include", "pvs_sqlite.h"), os.path.join(ROOT, "include", "pvs.h"), LIB]
    if os.path.exists(LIB_SQLITE) and all(os.path.getmtime(LIB_SQLITE) >= os.path.getmtime(d) for d in deps):
        return LIB_SQLITE
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-I", os.path.join(ROOT, "include"),
           src, "-o", LIB_SQLITE, "-L", HERE, "-l:libpvs.so", "-Wl,-rpath,$ORIGIN", "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"sqlite extension build failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return LIB_SQLITE


if __name__ == "__main__":
    force = "--force" in sys.argv
    jobs = None
    if "-j" in sys.argv:
        jobs = int(sys.argv[sys.argv.index("-j") + 1])
    print(build(force=force, jobs=jobs))
