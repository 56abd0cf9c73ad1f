"""CPU tests: the C ABI loads and exports every declared symbol, and the host logic
(scale artifact, .npy ingestion, quant policy, aggregation, rank/RRF, page merge)
matches the oracle and the reference's own expectations.  No GPU compute here."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as orc
import panoptikon_amd as pvs
from panoptikon_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pvs.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pvs_[a-z0-9_]+)\s*\(", header))
    declared -= {"pvs_status"}
    assert len(declared) >= 35
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (pvs_[a-z0-9_]+)", out))
    missing = declared - exported
    assert not missing, f"declared in pvs.h but not exported: {sorted(missing)}"
    assert declared == set(L.SYMBOLS), f"binding drift: {sorted(declared ^ set(L.SYMBOLS))}"
    assert pvs.lib().pvs_abi_version() == 5
    # the oracle is test infrastructure: the product library must not reference it
    assert "orc_" not in out
    deps = subprocess.check_output(["ldd", L.LIB_PATH], text=True)
    assert "oracle" not in deps and "torch" not in deps


def test_abi_layout_manifest_matches_the_headers_the_ctypes_mirrors_and_the_rust_stubs():
    """include/abi_layout.txt (tools/abi_layout.py: sizeof / offsetof of every struct that crosses the C ABI, printed by a C program
    compiled against the headers) is current; the ctypes mirrors lay their fields out identically; the `#[repr(C)]` stubs of
    INTEGRATION.md — the binding a Rust host would add (the reference declares its native dependency's layouts by hand too,
    db/sql_functions.rs:83-128) — produce the same offsets under Rust's repr(C) rules."""
    import ctypes as C
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_layout

    fresh = abi_layout.generate()
    committed = open(abi_layout.MANIFEST).read()
    assert fresh == committed, "include/abi_layout.txt is stale: python tools/abi_layout.py --write"
    man = abi_layout.parse_manifest(committed)
    for name in ("pvs_index_desc", "pvs_stats", "pvs_profile", "pvs_similar_opts", "pvs_rrf_branch", "pvs_ready_pair", "pvs_quant_resolved",
                 "pvs_microbench_result", "pvs_sqlite_api", "pvs_sqlite_load_result", "pvs_sqlite_backfill_result"):
        assert name in man and man[name]["fields"], name
    # ctypes mirrors (panoptikon_amd/_lib.py)
    mirrors = {"pvs_index_desc": L.IndexDesc, "pvs_stats": L.Stats, "pvs_profile": L.Profile, "pvs_similar_opts": L.SimilarOpts,
               "pvs_rrf_branch": L.RrfBranch, "pvs_ready_pair": L.ReadyPair, "pvs_quant_resolved": L.QuantResolved,
               "pvs_microbench_result": L.MicrobenchResult}
    for name, cls in mirrors.items():
        assert C.sizeof(cls) == man[name]["size"], name
        got = [(f, getattr(cls, f).offset, getattr(cls, f).size) for f, _ in cls._fields_]
        assert got == man[name]["fields"], (name, got, man[name]["fields"])
    # the Rust stubs: repr(C) = C's rules (each field at the next multiple of its alignment, the struct padded to its largest)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    width = {"u8": 1, "i8": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8}
    rust_to_c = {"PvsIndexDesc": "pvs_index_desc", "PvsStats": "pvs_stats", "PvsProfile": "pvs_profile", "PvsSimilarOpts": "pvs_similar_opts",
                 "PvsRrfBranch": "pvs_rrf_branch", "PvsReadyPair": "pvs_ready_pair", "PvsQuantResolved": "pvs_quant_resolved"}
    seen = set()
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        rname, body = m.group(1), m.group(2)
        if rname not in rust_to_c:
            continue  # (opaque handles)
        seen.add(rname)
        off, align_max, fields = 0, 1, []
        for decl in [d for d in re.split(r",(?![^<(]*[>)])", body) if ":" in d]:
            fname, ftype = [x.strip() for x in decl.split(":", 1)]
            size = 8 if ftype.startswith(("*", "Option<")) else width[ftype]
            off = (off + size - 1) // size * size
            fields.append((fname, off, size))
            off += size
            align_max = max(align_max, size)
        total = (off + align_max - 1) // align_max * align_max
        c = man[rust_to_c[rname]]
        assert fields == c["fields"] and total == c["size"], (rname, fields, c)
    assert seen == set(rust_to_c), f"INTEGRATION.md lacks the stubs {sorted(set(rust_to_c) - seen)}"


def test_product_fails_loudly_without_a_gpu():
    if pvs.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pvs.PvsError) as e:
        pvs.VectorIndex(pvs.I8, 768)
    assert e.value.status == L.ERR_DEVICE and "no CPU path" in e.value.message
    with pytest.raises(pvs.PvsError):
        pvs.quantize_int8(np.ones(4, np.float32), 1.0)
    with pytest.raises(pvs.PvsError):
        pvs.absmax(np.ones(4, np.float32))


def test_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "panoptikon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                # comments may cite the oracle; code must never include, import, link or dlopen it
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", text), f
                assert not re.search(r"\borc_[a-z0-9_]+\s*\(", text), f
                assert "libpvs_oracle" not in text, f


# db/vector_quants.rs:3588-3626
def test_scale_artifact_kats():
    assert pvs.scale_from_absmax(0.0) == 1.0
    assert pvs.scale_from_absmax(float("nan")) == 1.0
    assert pvs.scale_from_absmax(float("inf")) == 1.0
    for a in (11.0, 3.5, 1e-30, 0.2345):
        assert pvs.scale_from_absmax(a) == orc.scale_from_absmax(a)
    s = pvs.scale_from_absmax(3.5)
    assert pvs.artifact_scale(pvs.scale_artifact(s)) == s
    assert pvs.scale_artifact(s) == orc.scale_artifact(s)
    for bad in (b"", bytes(5), np.float32(0).tobytes(), np.float32(-1).tobytes(), np.float32("nan").tobytes(),
                np.float32("inf").tobytes()):
        assert pvs.artifact_scale(bad) is None


# pql/embedding_utils.rs:377-436 (fixtures copied as data from panoptikon/tests/fixtures/npy)
NPY_EXPECT = {
    "f32_1d.npy": ([0.0, 1.5, -2.25, 3.0], 1e-6),
    "f16_2d_c.npy": ([1.0, 2.0, 3.0], 1e-3),
    "f16_2d_f.npy": ([1.0, 2.0, 3.0], 1e-3),
    "f64_1d.npy": ([1e-3, -1e3, 42.125], 1e-3),
    "i16_1d.npy": ([-2.0, 0.0, 2.0, 1234.0], 0.0),
    "u8_1d.npy": ([0.0, 200.0, 255.0], 0.0),
    "u16_1d.npy": ([0.0, 65535.0], 0.0),
    "bool_1d.npy": ([1.0, 0.0, 1.0], 0.0),
    "be_f32_1d.npy": ([1.0, -2.5, 100.25], 1e-6),
}


@pytest.mark.parametrize("name", sorted(NPY_EXPECT))
def test_npy_fixtures(golden_dir, name):
    raw = open(os.path.join(golden_dir, "npy", name), "rb").read()
    got = np.frombuffer(pvs.embedding_from_npy_bytes(raw), "<f4")
    exp, tol = NPY_EXPECT[name]
    assert got.shape == (len(exp),)
    assert np.all(np.abs(got - np.array(exp, np.float32)) <= tol)
    # independent check with numpy's own reader (first row of 2-D arrays)
    import io

    arr = np.load(io.BytesIO(raw))
    row = arr if arr.ndim == 1 else arr[0]
    assert np.array_equal(got, row.astype(np.float32))


def _npy(arr, version=(1, 0), fortran=False):
    import io

    import numpy.lib.format as fmt

    bio = io.BytesIO()
    if fortran:
        arr = np.asfortranarray(arr)
    fmt.write_array(bio, arr, version=version)
    return bio.getvalue()


def test_npy_dtypes_versions_and_errors():
    import base64

    base = np.array([[1.5, -2.0, 3.25, 100.0], [9, 9, 9, 9]])
    for dt in ("<f2", "<f4", "<f8", ">f4", ">f8", "<i1", "<i2", ">i4", "<i8", "<u1", ">u2", "<u4", "<u8", "?"):
        a = (base != 0) if dt == "?" else np.abs(base).astype(dt) if "u" in dt else base.astype(dt)
        for version in ((1, 0), (2, 0), (3, 0)):
            for fortran in (False, True):
                raw = _npy(a, version, fortran)
                got = np.frombuffer(pvs.embedding_from_npy_bytes(raw), "<f4")
                assert np.array_equal(got, a[0].astype(np.float32)), (dt, version, fortran)
    one_d = _npy(np.arange(5, dtype="<f4"))
    assert np.frombuffer(pvs.extract_embeddings(base64.b64encode(one_d).decode()), "<f4").tolist() == [0, 1, 2, 3, 4]
    # the reference's f16 widening halves subnormals (reference quirk Q1; oracle pins it)
    sub = np.array([1, 0x200, 0x3FF, 0x8001, 0x0400], np.uint16)
    got = np.frombuffer(pvs.embedding_from_npy_bytes(_npy(sub.view(np.float16))), "<f4")
    assert np.array_equal(got, orc.npy_f16_bits_to_f32(sub))
    assert got[0] == np.float32(2.0**-25) and got[4] == np.float32(2.0**-14)
    for bad, msg in [
        (b"short", "Numpy buffer too small"),
        (b"\x93NUMPX" + bytes(20), "Invalid numpy magic header"),
        (b"\x93NUMPY\x04\x00" + bytes(20), "Unsupported numpy version 4.0"),
        (one_d[:-3], "Numpy data truncated"),
        (_npy(np.zeros((2, 2, 2), "<f4")), "Only 1D or 2D embeddings are supported"),
        (_npy(np.zeros(3, "<c8")), "Unsupported numpy dtype: <c8"),
        (_npy(np.float32(1.0)), "Numpy array has empty shape"),
    ]:
        with pytest.raises(pvs.PvsError) as e:
            pvs.embedding_from_npy_bytes(bad)
        assert e.value.status == L.ERR_PARSE and msg in e.value.message
    with pytest.raises(pvs.PvsError) as e:
        pvs.extract_embeddings("@@not-base64@@")
    assert "Invalid base64 embeddings" in e.value.message


# pql/preprocess.rs:314-446
def test_resolve_vector_quant_policy():
    dim = 8
    emb = np.arange(dim, dtype="<f4") * 0.25 - 1.0
    ready = L.ReadyPair(1, 1, 1, 42, 0.0625, dim)
    # exact: never quant
    assert pvs.resolve_vector_quant(pvs.INDEX_EXACT, None, 10, ready, emb.tobytes()) is None
    # auto + ready: default profile, query quantized with the frozen scale
    pid, qq = pvs.resolve_vector_quant(pvs.INDEX_AUTO, None, 10, ready, emb.tobytes())
    assert pid == 42 and np.array_equal(qq, orc.quantize_int8(emb, 0.0625))
    # no embedding (similar_to): profile only
    assert pvs.resolve_vector_quant(pvs.INDEX_QUANT, None, 1, ready, None) == (42, None)
    # blank variant == unset
    assert pvs.resolve_vector_quant(pvs.INDEX_AUTO, "   ", 10, ready, emb.tobytes())[0] == 42
    # ann is reserved; k must be positive
    for args in ((pvs.INDEX_ANN, None, 10), (pvs.INDEX_AUTO, None, 0), (pvs.INDEX_QUANT, None, -3)):
        with pytest.raises(pvs.PvsError) as e:
            pvs.resolve_vector_quant(args[0], args[1], args[2], ready, emb.tobytes())
        assert e.value.status == L.ERR_INVALID_ARG
    assert "reserved" in e.value.message or "positive" in e.value.message
    # fallbacks under auto, errors under quant / named variant
    cases = [
        (L.ReadyPair(0, 1, 1, 42, 0.0625, dim), "unavailable in this context"),
        (L.ReadyPair(1, 0, 1, 42, 0.0625, dim), "no default vector quant profile"),
        (L.ReadyPair(1, 1, 0, 42, 0.0625, dim), "not ready"),
    ]
    for pair, msg in cases:
        assert pvs.resolve_vector_quant(pvs.INDEX_AUTO, None, 10, pair, emb.tobytes()) is None
        with pytest.raises(pvs.PvsError) as e:
            pvs.resolve_vector_quant(pvs.INDEX_QUANT, None, 10, pair, emb.tobytes())
        assert e.value.status == L.ERR_NOT_READY and msg in e.value.message
    # a named variant is strict even under auto, and does not need a default profile
    with pytest.raises(pvs.PvsError):
        pvs.resolve_vector_quant(pvs.INDEX_AUTO, "gsym", 10, L.ReadyPair(1, 0, 0, 0, 0, dim), emb.tobytes())
    assert pvs.resolve_vector_quant(pvs.INDEX_AUTO, "gsym", 10, L.ReadyPair(1, 0, 1, 7, 0.5, dim), emb.tobytes())[0] == 7
    # dimension mismatch: silent fallback under auto, error when strict
    short = emb[:4].tobytes()
    assert pvs.resolve_vector_quant(pvs.INDEX_AUTO, None, 10, ready, short) is None
    with pytest.raises(pvs.PvsError) as e:
        pvs.resolve_vector_quant(pvs.INDEX_QUANT, None, 10, ready, short)
    assert e.value.status == L.ERR_DIM_MISMATCH and "expected 8, got 4" in e.value.message


def test_host_query_quant_matches_codec_kats():
    ready = L.ReadyPair(1, 1, 1, 1, 1.0, 8)
    _, qq = pvs.resolve_vector_quant(pvs.INDEX_QUANT, None, 1, ready,
                                     np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 2.4999, -2.4999], "<f4").tobytes())
    assert qq.tolist() == [0, 2, 2, 0, -2, -2, 2, -2]
    s = pvs.scale_from_absmax(11.0)
    ready = L.ReadyPair(1, 1, 1, 1, s, 4)
    _, qq = pvs.resolve_vector_quant(pvs.INDEX_QUANT, None, 1, ready, np.array([11.0, -11.0, 1000.0, -1000.0], "<f4").tobytes())
    assert qq.tolist() == [127, -127, 127, -128]
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(768) * 0.05).astype("<f4")
    x[3] = np.nan
    ready = L.ReadyPair(1, 1, 1, 1, 0.0015, 768)
    _, qq = pvs.resolve_vector_quant(pvs.INDEX_AUTO, None, 5, ready, x.tobytes())
    assert np.array_equal(qq, orc.quantize_int8(x, 0.0015))


def test_aggregate_rank_rrf_match_oracle():
    rng = np.random.default_rng(2)
    n = 5000
    dist = rng.random(n).astype(np.float32)
    dist[rng.integers(0, n, 200)] = np.nan
    grp = np.sort(rng.integers(0, 900, n)).astype(np.int64)
    w = (rng.random(n) + 0.1).astype(np.float32)
    for agg, oagg in ((pvs.AGG_MIN, orc.AGG_MIN), (pvs.AGG_MAX, orc.AGG_MAX), (pvs.AGG_AVG, orc.AGG_AVG)):
        g, v = pvs.aggregate(dist, grp, agg)
        og, ov = orc.aggregate(dist, grp, oagg)
        assert np.array_equal(g, og) and np.array_equal(v.view(np.uint64), ov.view(np.uint64))
    g, v = pvs.aggregate(dist, grp, pvs.AGG_MIN, weights=w)
    og, ov = orc.aggregate(dist, grp, orc.AGG_MIN, w=w)
    assert np.array_equal(v.view(np.uint64), ov.view(np.uint64))
    # SQL semantics of SUM(d*w)/SUM(w) (exact.rs:67-80): a NULL distance drops out of SUM(d*w) only
    g, v = pvs.aggregate([np.nan, 1.0, 3.0, np.nan], [7, 7, 8, 9], pvs.AGG_AVG, weights=[2.0, 1.0, 4.0, 5.0])
    assert g.tolist() == [7, 8, 9] and v[0] == 1.0 / 3.0 and v[1] == 3.0 and np.isnan(v[2])
    og, ov = orc.aggregate([np.nan, 1.0, 3.0, np.nan], [7, 7, 8, 9], orc.AGG_AVG, w=[2.0, 1.0, 4.0, 5.0])
    assert np.array_equal(v.view(np.uint64), ov.view(np.uint64))
    with pytest.raises(pvs.PvsError):
        pvs.aggregate(dist[:3], [3, 2, 1], pvs.AGG_MIN)
    ids = rng.permutation(len(v)).astype(np.int64)
    assert np.array_equal(pvs.row_number(v, ids), orc.row_number(v, ids))
    # SQLite: NULL is the smallest value in a window ORDER BY — first ascending, last descending
    assert pvs.row_number([0.5, 0.1, np.nan, 0.1], ids=[4, 9, 1, 3]).tolist() == [4, 3, 1, 2]
    assert pvs.row_number([0.5, 0.1, np.nan, 0.1], ids=[4, 9, 1, 3], descending=True).tolist() == [1, 3, 4, 2]
    assert np.array_equal(pvs.row_number(v, ids, descending=True), orc.row_number(v, ids, descending=True))
    # RRF with the production weights (quant_ab.rs:233-246): 5/1.0, 5/1.0, 10/0.7
    ranks = rng.integers(-1, 2000, (3, 400)).astype(np.int64)
    ks, ws = [5, 5, 10], [1.0, 1.0, 0.7]
    fused = pvs.rrf_fuse(ranks, ks, ws)
    exp = np.array([orc.rrf_score(ranks[:, i], ks, ws) for i in range(ranks.shape[1])])
    assert np.array_equal(fused.view(np.uint64), exp.view(np.uint64))
    assert pvs.rrf_fuse([[1]], [1], [1.0])[0] == 0.5  # default Rrf{k=1, weight=1.0}
    assert pvs.rrf_fuse([[-1]], [1], [1.0])[0] == 1.0 / float(9223372036854775806)


def test_merge_topk_equals_global_sort():
    rng = np.random.default_rng(4)
    world, batch, k = 4, 6, 25
    ids = np.full((world, batch, k), -1, np.int64)
    dist = np.full((world, batch, k), np.nan, np.float32)
    cnt = np.zeros((world, batch), np.uint32)
    pool_i, pool_d = [[] for _ in range(batch)], [[] for _ in range(batch)]
    for w in range(world):
        for q in range(batch):
            c = int(rng.integers(0, k + 1))
            d = np.round(rng.random(c), 1).astype(np.float32)  # coarse -> many ties
            if c > 2:
                d[-1] = np.nan
            i = np.sort(rng.choice(10_000, c, replace=False)) + w * 10_000
            pi, pd = orc.topk(d, k, ids=i)
            ids[w, q, : len(pi)], dist[w, q, : len(pi)], cnt[w, q] = pi, pd, len(pi)
            pool_i[q] += i.tolist()
            pool_d[q] += d.tolist()
    mi, md, mc = pvs.merge_topk(ids, dist, cnt, k)
    for q in range(batch):
        ei, ed = orc.topk(np.array(pool_d[q], np.float32), k, ids=np.array(pool_i[q], np.int64))
        assert mc[q] == len(ei) and np.array_equal(mi[q, : len(ei)], ei)
        assert np.array_equal(np.isnan(md[q, : len(ei)]), np.isnan(ed))
        assert (mi[q, len(ei):] == -1).all()


def test_shard_range_partitions_rows():
    for n, world in ((10, 1), (10, 3), (100_000_000, 8), (7, 8), (0, 4)):
        spans = [pvs.shard_range(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(0 <= a <= b <= n for a, b in spans)


def test_keyed_merge_of_per_item_pages():
    """pvs_merge_group_pages_keyed: the shard merge of per-item pages under `ORDER BY value, last_modified DESC` (model.rs:547-553)
    — (value asc, NULL last, key DESC, group id asc); without keys it is pvs_merge_group_pages."""
    import panoptikon_amd as pvs

    rng = np.random.default_rng(4)
    world, batch, k = 3, 4, 12
    groups = np.full((world, batch, k), -1, np.int64)
    values = np.full((world, batch, k), np.nan)
    keys = np.zeros((world, batch, k), np.int64)
    counts = np.zeros((world, batch), np.uint32)
    exp = []
    for q in range(batch):
        ent = []
        for w in range(world):
            c = int(rng.integers(0, k + 1))
            g = rng.choice(np.arange(w, 300, world), c, replace=False)             # disjoint groups per shard
            v = rng.integers(0, 4, c).astype(np.float64)                            # heavy ties
            v[rng.random(c) < 0.15] = np.nan                                        # NULL aggregates
            kk = rng.integers(0, 3, c).astype(np.int64)
            order = np.lexsort((g, -kk, np.where(np.isnan(v), 0, v), np.isnan(v)))  # each shard's page in the keyed order
            groups[w, q, :c], values[w, q, :c], keys[w, q, :c], counts[w, q] = g[order], v[order], kk[order], c
            ent += list(zip(v[order], kk[order], g[order]))
        ent.sort(key=lambda e: (np.isnan(e[0]), 0.0 if np.isnan(e[0]) else e[0], -e[1], e[2]))
        exp.append(ent[:k])
    og, ov, oc = pvs.merge_group_pages(groups, values, counts, k, keys=keys)
    differs = 0
    pg, _, _ = pvs.merge_group_pages(groups, values, counts, k)
    for q in range(batch):
        assert oc[q] == len(exp[q]) and og[q, : oc[q]].tolist() == [e[2] for e in exp[q]], q
        a, e = ov[q, : oc[q]], np.array([e[0] for e in exp[q]])
        assert np.array_equal(np.isnan(a), np.isnan(e)) and np.array_equal(a[~np.isnan(a)], e[~np.isnan(e)])
        assert (og[q, oc[q]:] == -1).all()
        differs += int(not np.array_equal(pg[q], og[q]))
    assert differs, "the keys must matter in this test"



def test_exact_wide_instances_use_no_scratch(tmp_path):
    """k_exact_wide loads its rows by inline assembly into registers the compiler cannot know are still in flight: a spill of one of
    them would store and reload stale data.  Every instance must compile without scratch (the 16-query instances hold 3 rows per lane
    for this reason: 4 spilled 12-32 bytes)."""
    import re
    import subprocess

    from panoptikon_amd import build as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "wide.s"
    cmd = [B.HIPCC, "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "panoptikon_amd", "csrc"), *B.HIPFLAGS,
           "--cuda-device-only", "-S", os.path.join(root, "panoptikon_amd", "csrc", "pvs_exact_wide.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    found = re.findall(r"\.name:\s+(\S*k_exact_wide\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", out.read_text())
    assert len(found) == 8, found
    assert all(int(p) == 0 for _, p in found), found
