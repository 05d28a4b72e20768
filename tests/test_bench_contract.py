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
    assert [s_["dtype"] for s_ in subs] == ["f64", "f64", "f32"] and all("error" not in s_ for s_ in subs), subs
    for s_ in subs:
        assert 0 < s_["roofline"]["cold"]["frac"] <= 1.0 and 0 < s_["roofline"]["frac"] <= 1.0
        assert s_["roofline"]["cold"]["copies"] >= 3
    assert abs(d["value_from_event_clock"] - d["value"]) < 0.05 * d["value"], "both clocks are printed and agree"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "GFLOPS" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
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
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "cpu_baseline" not in d
    assert d["config"]["nnz_per_gpu"] < (1 << 18) * 16, "rank 0 holds one row block, not the whole matrix"
    # blocks are balanced by cost = nnz + 2 per row (sharding.ROW_WEIGHT): half of (16 + 2) * 2^18 each
    cost = d["config"]["nnz_per_gpu"] + 2 * d["config"]["m_per_gpu"]
    assert abs(cost - (1 << 18) * 9) < 0.05 * (1 << 18) * 9
