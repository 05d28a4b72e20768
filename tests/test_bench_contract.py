"""The driver's contract for bench.py (one JSON line, fixed keys) -- checked on the GPU box with a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    d = _run("--gpus", "1", "--steps", "40", "--warmup", "5", "--cpu-seconds", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True
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
