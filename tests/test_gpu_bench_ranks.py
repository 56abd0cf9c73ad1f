"""bench.py's N > 1 paths as the driver starts them — real rank processes, the one-process multi-device form — on the GPUs this box
has: with one device two ranks share it (RCCL refuses that: `--allow-host-gather` lets the control socket stand in for the
all-gather, everything else — row shards, the global scale, per-shard searches, the merge, the parity leg against the oracle over
the WHOLE corpus — is the code an 8-GPU node runs); with two or more devices the same commands go over RCCL.  Every run must
print exactly one JSON line for the GPU count asked for, bit-exact against the oracle.  The lines are kept under
gpurun_out/bench_ranks/ (copied to profiles/ by the builder).  SURVEY.md section 8(e); north_star: "per-shard top-k merged via a small RCCL
all-gather"."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = os.path.join(ROOT, "gpurun_out", "bench_ranks")


@pytest.fixture(scope="module")
def pvs():
    import panoptikon_amd as p

    if p.device_count() < 1:
        pytest.fail("no gfx950 device visible: the gpu tests need an MI355X")
    return p


def _bench(name, *argv, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PVS_CTL_SOCK"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"bench.py {' '.join(argv)} -> rc {r.returncode}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}:\n{r.stdout[-2000:]}"
    rec = json.loads(lines[0])
    os.makedirs(KEEP, exist_ok=True)
    with open(os.path.join(KEEP, name + ".json"), "w") as f:
        f.write(lines[0] + "\n")
    return rec, r.stderr


COMMON = ["--steps", "5", "--warmup", "2", "--no-peaks", "--no-secondary", "--cpu-sample-rows", "20000", "--cpu-sample-queries", "2"]


def test_two_rank_processes_row_sharded(pvs):
    rec, err = _bench("gpus2_ranks_2Mx768_i8_b128", "--gpus", "2", "--allow-host-gather", "--rows", "2000000", *COMMON)
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["value"] > 0
    assert rec["parity"]["ids_and_distances_bit_exact"] is True and rec["recall_at_k"] == 1.0
    want = "rccl-allgather" if pvs.device_count() >= 2 else "ctl-host-gather"
    assert rec["config"]["exchange"] == want, (rec["config"]["exchange"], err[-1500:])
    assert "2" in rec["config"]["parallelism"]


def test_two_rank_processes_config3_shape(pvs):
    """BASELINE configs[3] (100M x 768 int8, 256 queries per pass, row shards + one gather) at 1/25 of its rows."""
    rec, err = _bench("gpus2_ranks_cfg3_4Mx768_i8_b256", "--config", "3", "--gpus", "2", "--allow-host-gather", "--rows", "4000000", *COMMON)
    assert rec["n_gpus"] == 2 and rec["config"]["batch"] == 256
    assert rec["parity"]["ids_and_distances_bit_exact"] is True and rec["recall_at_k"] == 1.0


def test_two_rank_processes_config4_shape(pvs):
    """BASELINE configs[4] (two int8 indexes sharded BY FILE, the PQL or-composition fused by RRF across ranks) at 2 x 1M rows."""
    rec, err = _bench("gpus2_ranks_cfg4_2x1M_i8", "--config", "4", "--gpus", "2", "--allow-host-gather", "--rows", "1000000", "--steps", "5", "--warmup", "2")
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    par = rec.get("parity", {})
    assert par, rec.keys()
    assert all(v is True for k, v in par.items() if isinstance(v, bool)), par


def test_one_process_two_shards_on_one_device(pvs):
    """The one-process form (a multi-device index, peer-copy gather): device list [0, 0] — two real shards on the device this box has."""
    rec, err = _bench("single_process_devices_0_0_2Mx768_i8_b128", "--single-process", "--gpus", "2", "--devices", "0,0", "--rows", "2000000", *COMMON)
    assert rec["n_gpus"] == 2 and rec["parity"]["ids_and_distances_bit_exact"] is True and rec["recall_at_k"] == 1.0
    assert "multi-device" in rec["config"]["exchange"]


# ---- round 6: the shapes an 8-GPU node runs, at FULL size on the GPU this box has.  Eight rank processes
# share the device (12.5M-row shards of configs[3]: 76.8 GB of codes in total; file shards of configs[4]); RCCL refuses ranks
# that share a GPU, so the page exchange goes over the control socket (`--allow-host-gather`, the line says so) — rendezvous,
# shard ranges, the global int8 scale, 8 concurrent shard builds and searches, the 8-way merge and the oracle parity leg ON
# EVERY RANK are the code the 8-GPU job runs.
def _free_gb(pvs):
    import ctypes as C

    f, t = C.c_uint64(), C.c_uint64()
    pvs._lib.check(pvs.lib().pvs_device_mem_info(0, C.byref(f), C.byref(t)))
    return f.value / 2**30


@pytest.mark.parametrize("ranks", [8, 4])
def test_config3_full_size_as_rank_processes(pvs, ranks):
    """BASELINE configs[3]: 100M x 768 int8 row-sharded, 256 queries per pass, k = 100 — all 100M rows, `ranks` processes."""
    if _free_gb(pvs) < 150:
        pytest.skip("needs ~120 GB of free HBM (100M x 768 int8 in shards + staging)")
    rec, err = _bench(f"gpus{ranks}_ranks_cfg3_100Mx768_i8_b256", "--config", "3", "--gpus", str(ranks), "--allow-host-gather", "--steps", "5", "--warmup", "2",
                      "--no-peaks", "--no-secondary", "--no-cpu-baseline", "--check-queries", "2" if ranks == 8 else "1", timeout=1500)
    assert rec["n_gpus"] == ranks and rec["steps"] == 5 and rec["value"] > 0
    assert rec["config"]["rows"] == 100_000_000 and rec["config"]["batch"] == 256 and "configs[3]" in rec["config"]["workload"]
    par = rec["parity"]
    assert par["ids_and_distances_bit_exact"] is True and par["oracle_rows"] == 100_000_000 and rec["recall_at_k"] == 1.0
    assert par["checked_on"] == f"every one of the {ranks} ranks"
    want = "rccl-allgather" if pvs.device_count() >= ranks else "ctl-host-gather"
    assert rec["config"]["exchange"] == want, (rec["config"]["exchange"], err[-1500:])


@pytest.mark.parametrize("ranks", [8, 4])
def test_config4_full_size_as_rank_processes(pvs, ranks):
    """BASELINE configs[4]: 2 x 25M rows (512-d image + 1024-d text, int8) sharded BY FILE, the PQL or-composition fused by RRF across
    the ranks (pvs_rrf_search_sharded), checked on every rank by exact global window ranks from the oracle's distances."""
    if _free_gb(pvs) < 100:
        pytest.skip("needs ~60 GB of free HBM")
    rec, err = _bench(f"gpus{ranks}_ranks_cfg4_2x25M_i8", "--config", "4", "--gpus", str(ranks), "--allow-host-gather", "--steps", "5", "--warmup", "2", timeout=1500)
    assert rec["n_gpus"] == ranks and rec["value"] > 0 and rec["config"]["rows"] == 50_000_000
    par = rec.get("parity", {})
    assert par and all(v is True for v in par.values() if isinstance(v, bool)), par
