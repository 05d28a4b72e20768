"""The driver's contract for bench.py (one JSON line, fixed keys) -- checked on the GPU box with a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None, expect_rc=0):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=1500, cwd=ROOT, env={**os.environ, **(env or {})})
    if expect_rc != 0:
        assert out.returncode != 0, out.stdout[-2000:]
        return out
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    """The driver's own command form.  The default workload is the BASELINE headline: R-MAT scale 24, fp64, whole
    matrix on one GPU at N = 1, with the other GPU configs as a `configs` array (warm and cold figures)."""
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2")
    assert "R-MAT scale 24" in d["config"]["workload"] and d["config"]["nnz_per_gpu"] == 268435456
    assert d["scaling"] == "strong"
    subs = d["configs"]
    assert [s_["dtype"] for s_ in subs] == ["f64", "f64", "f32", "f64"] and all("error" not in s_ for s_ in subs), subs
    for s_ in subs:
        # working sets below the Infinity Cache: the COLD protocol's figure is the roofline figure, warm is a sub-key
        r_ = s_["roofline"]
        assert r_["protocol"].startswith("cold") and r_["copies"] >= 3 and s_["data"] == "synthetic stand-in"
        assert 0 < r_["frac"] <= r_["warm"]["frac"] <= 1.0, (s_["workload"], r_["frac"], r_["warm"]["frac"])
        assert abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-3
        assert s_["value"] <= s_["warm"]["value"] * 1.02 and "traffic_note" in r_
    assert abs(d["value_from_event_clock"] - d["value"]) < 0.05 * d["value"], "both clocks are printed and agree"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "GFLOPS" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r and "traffic_note" in r
    assert "warm" not in r, "R-MAT 24 is far beyond the Infinity Cache: back-to-back steps already stream from HBM"
    # flat scalar keys (what the driver's parsed record keeps): sub-config fractions, the snapshot protocol, the conversion
    for name in ("scircuit", "webbase", "nd24k", "nd24k_f64"):
        assert 0 < r[f"{name}_cold_frac"] <= r[f"{name}_warm_frac"] <= 1.0, name
    # the locality bracket of the power-law stand-ins: three scalars per point, every point at or above its harsh stand-in
    for tag, base in (("webbase_b06pl", "webbase"), ("webbase_b09pl", "webbase"), ("scircuit_b08", "scircuit"), ("scircuit_b095", "scircuit")):
        assert r[f"{base}_cold_frac"] * 0.98 <= r[f"{tag}_cold_frac"] <= r[f"{tag}_warm_frac"] <= 1.0, (tag, r[f"{tag}_cold_frac"])
        assert r[f"{tag}_path"] == "plain", (tag, r[f"{tag}_path"])
    assert 0 < r["x_snapshot_frac"] < 1 and r["conversion_ms"] > 0 and r["conversion_in_spmvs"] > 0 and 0 < r["conversion_frac"] < 1
    assert d["value_multi_gpu_protocol"] >= d["value"] * 0.98
    assert abs(d["value"] - 2.0 * d["config"]["nnz_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["max_rel_err_gpu_vs_cpu"] <= 1e-6


@pytest.mark.gpu
def test_bench_on_a_matrix_market_file():
    d = _run("--mtx", os.path.join(ROOT, "tests", "golden", "mtx", "symmetric_real.mtx"), "--steps", "20", "--warmup", "2",
             "--no-cpu-baseline")
    assert d["config"]["nnz_per_gpu"] == 717 and d["config"]["ingest_ms"] is not None


@pytest.mark.gpu
def test_bench_multi_gpu_form_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher must start N ranks itself and print n_gpus = N -- or refuse; it must
    never print an n_gpus = 1 line for N > 1.  On a 1-GPU box the ranks share the device (test hook, gloo)."""
    import torch
    if torch.cuda.device_count() < 2:
        _run("--gpus", "2", "--workload", "rmat16", "--steps", "5", "--warmup", "1", expect_rc=1)
    d = _run("--gpus", "2", "--workload", "rmat18", "--steps", "10", "--warmup", "2",
             env={"CSR5_BENCH_SHARE_GPU": "1"} if torch.cuda.device_count() < 2 else None)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["nnz_per_gpu"] < (1 << 18) * 16, "rank 0 holds one row block, not the whole matrix"
    # blocks are balanced by cost = nnz + 2 per row (sharding.ROW_WEIGHT): half of (16 + 2) * 2^18 each
    cost = d["config"]["nnz_per_gpu"] + 2 * d["config"]["m_per_gpu"]
    assert abs(cost - (1 << 18) * 9) < 0.05 * (1 << 18) * 9
    _check_multi_gpu_keys(d, 2, 1 << 18)


def _check_multi_gpu_keys(d, world, rows):
    """What lets a scaling record be verified rank by rank: the communicator size, the x-broadcast time, and every rank's
    rows / nnz / cost / step times / roofline fraction."""
    mg = d["multi_gpu"]
    assert mg["comm_world_size"] == world and mg["ranks_reporting"] == world and mg["collectives_per_step"] == 0
    assert mg["comm_backend"] in ("nccl", "gloo") and mg["x_broadcast_ms_max"] > 0 and mg["x_bytes"] == rows * 8
    ranks = mg["ranks"]
    assert [r["rank"] for r in ranks] == list(range(world))
    assert sum(r["rows"] for r in ranks) == rows and sum(r["nnz"] for r in ranks) == rows * 16
    costs = [r["cost_nnz_plus_2_rows"] for r in ranks]
    assert max(costs) - min(costs) < 0.1 * max(costs), costs
    for r in ranks:
        assert r["event_ms_per_step"] > 0 and r["wall_ms_per_step"] > 0 and 0 < r["roofline_frac"] <= 1.0
    # the job's step time is the slowest rank's
    assert d["event_ms_per_step"] >= max(r["event_ms_per_step"] for r in ranks) * 0.999
    # round 6: an N > 1 line carries correctness evidence and a CPU baseline like the N = 1 line -- EVERY rank's block checked
    # against CSR5_avx2 (exact on the integer data), rank 0's block and the whole matrix timed on the host cores, and the N = 1
    # step under the same x protocol (x captured once behind the broadcast) measured in the same run
    assert mg["max_rel_err_gpu_vs_cpu"] == 0.0 and mg["checker"] in ("reference", "port")
    assert len(mg["per_rank_max_rel_err"]) == world and all(e == 0.0 for e in mg["per_rank_max_rel_err"])
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "scope", "whole_matrix_value"):
        assert key in c, (key, c)
    assert "rank 0's row block" in c["sample"] and c["max_rel_err_gpu_vs_cpu"] == 0.0
    assert c["whole_matrix"]["max_rel_err_gpu_vs_cpu"] == 0.0 and "WHOLE matrix" in c["whole_matrix"]["sample"]
    n1 = mg["n1_same_protocol"]
    assert n1["ms_per_step"] > 0 and abs(mg["speedup_vs_n1_same_protocol"] - n1["ms_per_step"] / d["event_ms_per_step"]) < 0.01 * mg["speedup_vs_n1_same_protocol"] + 1e-3
    assert "broadcast" in d["config"]["x_protocol"] and "roofline" in d and "conversion_ms" in d["roofline"]


@pytest.mark.gpu
def test_bench_eight_ranks_report_every_rank():
    """N = 8 (the driver's largest scaling point) on R-MAT 20: on a box with fewer GPUs the ranks share the device (test
    hook, gloo) -- the control flow, the sharding and the per-rank report are the ones an 8-GPU node runs."""
    import torch
    shared = torch.cuda.device_count() < 8
    d = _run("--gpus", "8", "--workload", "rmat20", "--steps", "10", "--warmup", "2",
             env={"CSR5_BENCH_SHARE_GPU": "1"} if shared else None)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    _check_multi_gpu_keys(d, 8, 1 << 20)
    assert d["multi_gpu"]["shared_device_test_hook"] is shared
    assert abs(d["value"] - 2.0 * (16 << 20) / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"], "whole-job aggregate"


@pytest.mark.gpu
def test_bench_picks_up_real_suitesparse_files(tmp_path):
    """$CSR5_MTX_DIR/{scircuit,webbase-1M,nd24k}.mtx replace the synthetic stand-ins (no such file is obtainable offline:
    committed Matrix Market fixtures stand in for them here) and the entries say which data they ran on."""
    import shutil
    mtx = os.path.join(ROOT, "tests", "golden", "mtx")
    shutil.copy(os.path.join(mtx, "general_real.mtx"), tmp_path / "scircuit.mtx")
    shutil.copy(os.path.join(mtx, "symmetric_real.mtx"), tmp_path / "nd24k.mtx")
    d = _run("--workload", "scircuit", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", env={"CSR5_MTX_DIR": str(tmp_path)})
    assert "scircuit.mtx" in d["data"] and "suitesparse" in d["data"] and d["config"]["ingest_ms"] is not None
    assert "scircuit.mtx" in d["config"]["workload"]
    assert d["roofline"]["protocol"].startswith("cold") and "warm" in d["roofline"]
    d = _run("--workload", "nd24k", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", env={"CSR5_MTX_DIR": str(tmp_path)})
    assert "nd24k.mtx" in d["data"] and d["dtype"] == "f32" and d["config"]["nnz_per_gpu"] == 717
    # a workload whose file is absent keeps its stand-in
    d = _run("--workload", "webbase", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--scale", "0.05",
             env={"CSR5_MTX_DIR": str(tmp_path)})
    assert d["data"] == "synthetic"


def test_real_file_lookup(tmp_path, monkeypatch):
    """CPU: which file a workload maps to, and that nothing is picked up without $CSR5_MTX_DIR."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("CSR5_MTX_DIR", raising=False)
    assert bench.real_file_for("webbase") is None
    monkeypatch.setenv("CSR5_MTX_DIR", str(tmp_path))
    assert bench.real_file_for("webbase") is None
    (tmp_path / "webbase-1M.mtx").write_text("%%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 1.0\n")
    assert bench.real_file_for("webbase") == str(tmp_path / "webbase-1M.mtx")
    assert bench.real_file_for("rmat24") is None and bench.real_file_for("scircuit") is None
