"""Multi-GPU path behind the C ABI (csr5hip_multi_*): cost-balanced (nnz + 2 per row) row blocks, one handle per shard, x replicated
once, y sharded.  On a 1-GPU box every shard lives on device 0 (the device list repeats it), so the whole control
flow -- device-side row cuts, shard copies + rebase, per-shard conversion, set_x, per-stream SpMV, gather -- runs;
the RCCL broadcast between devices needs >= 2 distinct devices; its bindings and call sequence run here through a one-device
communicator (test_rccl_broadcast_runs_on_one_device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import _capi  # noqa: E402
from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from benchmark_spmv_using_csr5_amd import sharding as S  # noqa: E402
from tests import zoo  # noqa: E402

DEV = "cuda:0"


def _devices(G):
    n = torch.cuda.device_count()
    return [g % n for g in range(G)] if n >= 2 else [0] * G


def _run_multi(mat, val, x, G, dtype=np.float64, sigma=-1, slabs=None, row_weight=None, own_replicas=False, narrow=None):
    rp = torch.from_numpy(mat.row_ptr.astype(np.int32)).to(DEV)
    ci = torch.from_numpy(mat.col.astype(np.int32)).to(DEV)
    va = torch.from_numpy(val.astype(dtype)).to(DEV)
    xd = torch.from_numpy(x.astype(dtype)).to(DEV)
    A = H.MultiGpuHandle(_devices(G), mat.m, mat.n, dtype=np.dtype(dtype).name)
    if row_weight is not None:
        assert A.setOption(100, row_weight) == 0  # CSR5HIP_MULTI_OPT_ROW_WEIGHT, before inputCSR
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0
    assert A.setSigma(sigma) == 0
    if slabs is not None:
        assert A.setOption(6, slabs) == 0
    if narrow is not None:
        assert A.setOption(9, 2) == 0  # LDS hot table forced (a 2 M-nnz shard is below the auto rule's size threshold)
    if narrow == "before":
        assert A.setOption(_capi.OPT_NARROW_VALUES, 1) == 0
    assert A.asCSR5() == 0
    if narrow == "after":  # every shard rebuilds its slab structure
        assert A.setOption(_capi.OPT_NARROW_VALUES, 1) == 0
    if narrow is not None:
        narrowed = [A.shard_info(g).slab_values_narrowed for g in range(G)]
        hot = [A.shard_info(g).slab_hot for g in range(G)]
        assert narrowed == hot and any(hot), "integer data: every shard with a hot table streams fp32 values"
    if own_replicas:
        assert A.setOption(_capi.MULTI_OPT_OWN_REPLICAS, 1) == 0
    assert A.setX(xd) == 0
    if own_replicas:
        torch.cuda.synchronize()
        xd.fill_(float("nan"))  # the shards must read the replica the broadcast filled, not the caller's vector
        torch.cuda.synchronize()
    assert A.fill_y(0x7F) == 0  # a recognisable pattern in the rows SpMV must leave untouched
    assert A.spmv(1.0) == 0 and A.synchronize() == 0
    y = A.gather_y()
    cuts = [A.shard(g).row_lo for g in range(G)] + [A.shard(G - 1).row_hi]
    nnzs = [A.shard(g).nnz for g in range(G)]
    bkind = A.shard(0).x_broadcast
    tails = [A.shard(g).row_lo + A.shard_info(g).tail_partition_start for g in range(G)]
    # the caller's arrays are copies-from only: still plain CSR
    assert np.array_equal(ci.cpu().numpy(), mat.col)
    assert A.destroy() == 0
    A.close()
    return y, cuts, nnzs, bkind, tails


def test_rccl_broadcast_runs_on_one_device(oracle):
    """The RCCL branch of csr5hip_multi_set_x (dlopen of librccl, the hand-declared prototypes of ncclCommInitAll /
    ncclGroupStart / ncclBroadcast / ncclGroupEnd, the ncclUint8 constant) executed on the 1-GPU box: a one-device
    communicator and a grouped broadcast from devices[0] into the shard's own replica of x.  What an 8-GPU node runs per
    device, proven against the installed librccl before the first multi-GPU lease (nothing to match in the reference:
    CSR5_cuda/main.cu:25-26 is cudaSetDevice(0))."""
    for mat in zoo.small_zoo()[:6]:
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=9, mode="int")
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        y, cuts, nnzs, bkind, tails = _run_multi(mat, val, x, 1, own_replicas=True)
        assert bkind == 1, f"x was not replicated by the RCCL broadcast (kind {bkind}): {_capi.last_error()}"
        has = np.diff(mat.row_ptr) > 0
        assert np.array_equal(y[has], ref[has]), mat.name
    # fp32, a matrix large enough for slabs: the replica feeds the slab child's permuted copy
    mat = M.rmat(16, 8, seed=3)
    val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=4, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
    y, cuts, nnzs, bkind, tails = _run_multi(mat, val, x, 1, dtype=np.float32, own_replicas=True)
    assert bkind == 1
    has = np.diff(mat.row_ptr) > 0
    assert np.array_equal(y[has].astype(np.float64), ref[has])


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_multi_zoo_exact(oracle, G):
    poison = np.frombuffer(bytes([0x7F] * 8), dtype=np.float64)[0]
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        weight = (None, 0, 5)[(G + mat.m) % 3]
        y, cuts, nnzs, bkind, tails = _run_multi(mat, val, x, G, row_weight=weight)
        want = S.partition_rows_by_cost(mat.row_ptr, G, S.ROW_WEIGHT if weight is None else weight)
        assert cuts == list(want), "device-side cuts == sharding.partition_rows_by_cost"
        assert sum(nnzs) == mat.nnz
        lens = np.diff(mat.row_ptr)
        nonempty = lens > 0
        assert np.array_equal(y[nonempty], ref[nonempty]), (mat.name, G)
        # empty rows before a shard's tail keep the caller's bytes; rows from the tail start on are written (0)
        rows = np.arange(mat.m)
        for g in range(G):
            blk = (rows >= cuts[g]) & (rows < cuts[g + 1]) & ~nonempty
            before = blk & (rows < tails[g])
            assert np.all(y[before] == poison), (mat.name, G, g)
            assert np.all(y[blk & (rows >= tails[g])] == 0), (mat.name, G, g)
        if G > 1 and torch.cuda.device_count() >= G:
            assert bkind == 1, "distinct devices: x must travel by the RCCL broadcast"


def test_multi_rmat20_eight_row_blocks_on_one_device(oracle):
    """The 8 cost-balanced row blocks of one R-MAT 20 (what 8 GPUs would hold), each through its own handle; the
    concatenated y must equal the full-matrix product exactly, with and without column slabs in the shards."""
    mat = M.rmat(20, 16, seed=4)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=8, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    nonempty = np.diff(mat.row_ptr) > 0
    for slabs, narrow in ((0, None), (1, None), (1, "before"), (8, "after")):
        y, cuts, nnzs, _, _ = _run_multi(mat, val, x, 8, slabs=slabs, narrow=narrow)
        assert np.array_equal(y[nonempty], ref[nonempty]), (slabs, narrow)
        cost = np.asarray(nnzs) + S.ROW_WEIGHT * np.diff(cuts)
        assert cost.max() <= 1.1 * cost.sum() / 8, "row blocks are balanced by non-zeros + ROW_WEIGHT * rows (power-law rows)"
        assert max(nnzs) <= 1.25 * mat.nnz / 8


def test_multi_set_x_captures_x_once(oracle):
    """The multi handle's x contract (csr5hip.h, round 6): set_x CAPTURES x -- every shard's permuted copy of x (hot-table path)
    is taken once behind the broadcast, not by every spmv().  Shards default to CSR5HIP_OPT_X_SNAPSHOT = 1; writing to x without
    set_x leaves y as it was, set_x with the same pointer picks the new contents up; X_SNAPSHOT = 0 through the multi handle
    gives the shards that BORROW the caller's vector their live reads back (a library-owned replica stays captured)."""
    mat = M.rmat(16, 16, seed=6)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=2, mode="int")
    x2 = (x + 1.0) % 7
    has = np.diff(mat.row_ptr) > 0
    ref1 = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    ref2 = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x2)
    rp = torch.from_numpy(mat.row_ptr.astype(np.int32)).to(DEV)
    ci = torch.from_numpy(mat.col.astype(np.int32)).to(DEV)
    va = torch.from_numpy(val).to(DEV)
    for own in (False, True):
        xd = torch.from_numpy(x).to(DEV)
        G = 2
        A = H.MultiGpuHandle(_devices(G), mat.m, mat.n)
        assert A.inputCSR(mat.nnz, rp, ci, va) == 0
        assert A.setSigma(-1) == 0
        assert A.setOption(6, 8) == 0 and A.setOption(9, 2) == 0  # 8 column slabs + LDS hot table in every shard
        if own:
            assert A.setOption(_capi.MULTI_OPT_OWN_REPLICAS, 1) == 0
        assert A.setX(xd) == 0  # before the conversion: as_csr5 takes the copy
        assert A.asCSR5() == 0
        for g in range(G):
            i = A.shard_info(g)
            assert i.slab_hot == 1 and i.slab_x_permuted == 1 and i.x_snapshot == 1, (g, i.slab_hot, i.x_snapshot)
        assert A.spmv(1.0) == 0 and A.synchronize() == 0
        assert np.array_equal(A.gather_y()[has], ref1[has])
        xd.copy_(torch.from_numpy(x2).to(DEV))  # the caller writes to x and does NOT tell the handle
        torch.cuda.synchronize()
        assert A.spmv(1.0) == 0 and A.synchronize() == 0
        assert np.array_equal(A.gather_y()[has], ref1[has]), "x was captured at set_x"
        assert A.setX(xd) == 0  # same pointer, new contents
        assert A.spmv_repeat(1.0, 3) == 0 and A.synchronize() == 0
        assert np.array_equal(A.gather_y()[has], ref2[has])
        # live reads for borrowing shards only
        assert A.setOption(_capi.OPT_X_SNAPSHOT, 0) == 0
        borrowing = [not own and A.shard(g).device == A.shard(0).device for g in range(G)]
        assert [A.shard_info(g).x_snapshot for g in range(G)] == [0 if b else 1 for b in borrowing]
        if all(borrowing):
            xd.copy_(torch.from_numpy(x).to(DEV))
            torch.cuda.synchronize()
            assert A.spmv(1.0) == 0 and A.synchronize() == 0
            assert np.array_equal(A.gather_y()[has], ref1[has]), "live x"
        assert A.destroy() == 0
        A.close()


def test_multi_real_data_fp32_and_graph_replay(oracle):
    mat = zoo.small_zoo()[14]  # scircuit-like, small
    val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=3, mode="real")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
    scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64), np.abs(x).astype(np.float64))
    rp = torch.from_numpy(mat.row_ptr).to(DEV)
    ci = torch.from_numpy(mat.col).to(DEV)
    va = torch.from_numpy(val).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    A = H.MultiGpuHandle(_devices(4), mat.m, mat.n, dtype="float32")
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setSigma(-1) == 0 and A.asCSR5() == 0 and A.setX(xd) == 0
    assert A.timer_start() == 0
    assert A.spmv_repeat(1.0, 20) == 0
    ms = A.timer_stop()
    assert ms > 0
    y = A.gather_y().astype(np.float64)
    assert np.all(np.abs(y - ref) <= 1e-5 * np.maximum(scale, 1.0))  # fp32 bar of the single-GPU tests
    A.destroy()
    A.close()


def test_sharding_hip_local_spmv_single_process(oracle):
    """`sharding.ShardedSpmv.run` with the GPU kernel factory (the per-rank piece of the one-process-per-GPU form):
    every cost-balanced row block of one matrix through `hip_local_spmv`, concatenated == the full product."""
    mat = M.scircuit_like(scale=0.2)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=2, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    xd = torch.from_numpy(x).to(DEV)
    world = 4
    parts = []
    for rank in range(world):
        sh = S.ShardedSpmv(mat.row_ptr, mat.col, val, mat.n, rank, world)
        run = S.hip_local_spmv(torch.device(DEV))
        y = sh.run(xd, run)  # no process group: broadcast_x is a no-op, as on rank 0 of a 1-rank job
        torch.cuda.synchronize()
        parts.append(y.cpu().numpy())
        run.state["A"].destroy()
        run.state["A"].close()
    assert np.array_equal(np.concatenate(parts), ref)


def test_cli_on_several_gpus(tmp_path):
    """`CSR5_GPU_LIST=0,0,0 ./spmv file.mtx` (or CSR5_GPUS=G with G real devices): the reference protocol through
    anonymouslibMultiHandle -- the reference's lines, one line per shard, the x replication line, and its self-check."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "benchmark_spmv_using_csr5_amd", "csrc", "spmv")
    mat = M.scircuit_like(scale=0.05)
    mat.val[:] = 1.0
    path = tmp_path / "m.mtx"
    M.write_mtx(str(path), mat)
    devs = _devices(3)
    env = dict(os.environ, CSR5_SEED="7", CSR5_GPU_LIST=",".join(str(d) for d in devs))
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr
    text = out.stdout
    pos = -1
    for token in ["PRECISION = 64-bit Double Precision", f" ) nnz = {mat.nnz}", "cpu sequential time = ", "Device [",
                  "GPU shard 0 on device", "GPU shard 2 on device", "CSR->CSR5 time = ", "x replicated on 3 GPUs",
                  "CSR5-based SpMV time = ", "max over 3 GPUs", "Check... PASS!"]:
        nxt = text.find(token, pos + 1)
        assert nxt > pos, (token, text)
        pos = nxt
