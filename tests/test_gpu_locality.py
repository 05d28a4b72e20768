"""The auto rules that pick the SpMV path (plain tile kernel / column slabs / slabs + LDS hot table) along the LOCALITY axis of
the power-law BASELINE stand-ins (the full sweep with cold figures is scripts/experiments/round6/locality.py
-> profiles/r06_locality.md).  Decisions are asserted exactly; times with a margin a noisy box cannot break (cold protocol,
auto within 8 % of the best forced path; the sweep's own bar is 5 %)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from tests.test_gpu_parity import _run  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _locality():
    spec = importlib.util.spec_from_file_location("locality", os.path.join(ROOT, "scripts", "experiments", "round6", "locality.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# (stand-in, band, far columns) -> (column slabs, hot table) the auto rules must pick, and why
DECISIONS = [
    ("webbase", 0.3, "uniform", (4, 0)),    # x = 8 MB, scattered uniform columns: slabs, nothing worth a table slot
    ("webbase", 0.3, "powerlaw", (0, 0)),   # skewed columns on a 3 M-nnz matrix: too small for the table; L2 keeps the hubs -> plain
    ("webbase", 0.6, "uniform", (0, 0)),    # windows cover > 50 %: the x lines stay in L2 anyway
    ("webbase", 0.9, "powerlaw", (0, 0)),
    ("scircuit", 0.5, "uniform", (0, 0)),   # x = 1.4 MB < one XCD's L2
    ("scircuit", 0.95, "uniform", (0, 0)),
    ("rmat19", 0, "rmat", (8, 1)),          # 8.4 M nnz, 90 % covered: the table pays (13 % cold)
]


@pytest.mark.parametrize("kind,band,far,expect", DECISIONS)
def test_auto_rule_decisions_along_the_locality_axis(oracle, kind, band, far, expect):
    L = _locality()
    mat = L.make_matrix(kind, band, far, 1.0)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=21, mode="int")
    info = {}
    _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, y0=0.0, info_out=info)
    assert (info["column_slabs"], info["slab_hot"]) == expect, (kind, band, far, info)
    if (kind, far) == ("webbase", "powerlaw") and band == 0.3:
        assert info["slab_hot_cover_pct"] >= 25, "the estimate that explains the choice stays visible in csr5hip_info"
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    assert np.array_equal(ys[0], ref)


@pytest.mark.parametrize("kind,band,far", [("webbase", 0.3, "uniform"), ("webbase", 0.3, "powerlaw"), ("webbase", 0.9, "uniform")])
def test_auto_rule_is_not_a_mispick(kind, band, far):
    L = _locality()
    paths = [p for p in L.PATHS if p[0] in ("auto", "plain", "slabs", "slabs+table")]
    p = L.run_point(kind, band, far, 1.0, torch.device("cuda", 0), steps=200, paths=paths, cold=True)
    assert p["auto_over_best_cold"] <= 1.08, p
