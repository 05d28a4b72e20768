"""1-D row-block sharding of a CSR matrix over the GPUs of one node.

The reference is single-device (CSR5_cuda/main.cu:25-26 `cudaSetDevice(0)`); this is the MI355X-native
addition named by BASELINE.json: rows are independent, so the matrix is cut into contiguous row blocks
balanced by COST = non-zeros + ROW_WEIGHT per row (split points = upper_bound of the cost prefix at g*total/G,
the same primitive the reference uses for tile_ptr, utils_cuda.h:25-53; ROW_WEIGHT = 0 is the plain nnz
balance), every rank converts and multiplies its own block with its own
handle, x is replicated once by an RCCL broadcast over xGMI, y stays sharded.  There is no per-iteration
collective.  One process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in the
CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


# What a row costs in units of one non-zero when row blocks are balanced.  Measured on the eight row blocks of R-MAT
# scale 24 (scripts/experiments/shard_alone.py): blocks of equal nnz take 200 us where rows hold 477 non-zeros and 285 us
# where they hold 4.6 -- the y element, the row pointer and the row's share of the slab combine are paid per ROW:
# (285 - 200) us / 7.2 M rows = 11.7 ps per row against 6.0 ps per non-zero.
ROW_WEIGHT = 2


def partition_rows_by_cost(row_ptr: np.ndarray, parts: int, row_weight: int = ROW_WEIGHT) -> np.ndarray:
    """Row split points r_0=0 <= r_1 <= ... <= r_parts=m with ~1/parts of the cost per block, where the cost of rows
    [0, r) is row_ptr[r] + row_weight * r (non-zeros plus row_weight per row; row_weight = 0 is SURVEY 8(e)'s plain
    nnz balance).

    r_g = (number of r whose cost prefix is <= g*total/parts) - 1 clipped to [r_{g-1}, m]: the reference's own
    `tile_ptr` primitive (upper bound on a sorted array, utils_cuda.h:25-53) on the cost prefix."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    m = row_ptr.size - 1
    key = row_ptr + int(row_weight) * np.arange(m + 1, dtype=np.int64)
    total = int(key[m])
    cuts = np.zeros(parts + 1, dtype=np.int64)
    cuts[parts] = m
    for g in range(1, parts):
        target = (g * total) // parts
        r = int(np.searchsorted(key, target, side="right")) - 1
        cuts[g] = min(max(r, cuts[g - 1]), m)
    return cuts


def partition_rows_by_nnz(row_ptr: np.ndarray, parts: int) -> np.ndarray:
    """~nnz/parts non-zeros per block: the row that contains non-zero number g*nnz/parts starts block g."""
    return partition_rows_by_cost(row_ptr, parts, 0)


@dataclass
class RowBlock:
    """The shard one rank owns: rows [row_lo, row_hi) with a rebased row_ptr and GLOBAL columns."""
    rank: int
    row_lo: int
    row_hi: int
    n: int
    row_ptr: np.ndarray
    col: np.ndarray
    val: np.ndarray

    @property
    def m(self) -> int:
        return self.row_hi - self.row_lo

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])


def extract_row_block(row_ptr, col, val, n: int, cuts: np.ndarray, rank: int) -> RowBlock:
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    a, b = int(row_ptr[lo]), int(row_ptr[hi])
    rp = (np.asarray(row_ptr[lo: hi + 1], dtype=np.int64) - a).astype(np.int32)
    return RowBlock(rank, lo, hi, n, rp, np.ascontiguousarray(col[a:b]), np.ascontiguousarray(val[a:b]))


def broadcast_x(x, src: int = 0):
    """The ONE collective of the sharded SpMV: replicate x on every rank (n*sizeof(vT) bytes)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(x, src=src)
    return x


def gather_y(y_local, cuts: np.ndarray):
    """Collect the y shards on every rank (correctness checks only -- not part of the timed path)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    sizes = [int(cuts[r + 1] - cuts[r]) for r in range(world)]
    width = max(sizes) if sizes else 0
    pad = torch.zeros(width, dtype=y_local.dtype, device=y_local.device)
    pad[: y_local.numel()] = y_local
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


class ShardedSpmv:
    """y_shard = A[row block of this rank] * x, with x broadcast once.

    `local_spmv(block, x) -> y_block` is the per-rank kernel: on the GPU box it wraps an
    `anonymouslibHandle` (see `hip_local_spmv`); the CPU tests pass a host CSR loop."""

    def __init__(self, row_ptr, col, val, n: int, rank: int, world: int):
        self.cuts = partition_rows_by_cost(row_ptr, world)
        self.block = extract_row_block(row_ptr, col, val, n, self.cuts, rank)
        self.rank, self.world = rank, world

    def run(self, x, local_spmv):
        x = broadcast_x(x, src=0)
        return local_spmv(self.block, x)


def _check(rc: int, what: str) -> None:
    """State-changing handle calls must run (and fail loudly) under ``python -O`` too: never inside an assert."""
    if rc != 0:
        from . import _capi
        raise RuntimeError(f"{what} -> {rc}: {_capi.last_error()}")


def hip_local_spmv(device, sigma: int = -1, mode: int = 1):
    """Factory for ShardedSpmv.run on a GPU: converts the block to CSR5 once, returns y (device).
    (One process per GPU over torch.distributed; the single-process form of the same sharding lives behind the C ABI,
    ``csr5hip_multi_*`` / ``handle.MultiGpuHandle``.)"""
    import torch

    from . import handle as H

    state = {}

    def run(block: RowBlock, x):
        if "A" not in state:
            tdt = torch.float64 if block.val.dtype == np.float64 else torch.float32
            rp = torch.from_numpy(block.row_ptr).to(device)
            ci = torch.from_numpy(block.col.astype(np.int32)).to(device)
            va = torch.from_numpy(block.val).to(device)
            A = H.anonymouslibHandle(block.m, block.n, dtype=str(block.val.dtype))
            _check(A.inputCSR(block.nnz, rp, ci, va), "inputCSR")
            _check(A.setSigma(sigma), "setSigma")
            _check(A.setSpmvMode(mode), "setSpmvMode")
            _check(A.setStream(torch.cuda.current_stream(device).cuda_stream), "setStream")
            _check(A.asCSR5(), "asCSR5")
            state.update(A=A, keep=(rp, ci, va), y=torch.zeros(block.m, dtype=tdt, device=device))
        A = state["A"]
        _check(A.setX(x), "setX")
        _check(A.spmv(1.0, state["y"]), "spmv")
        return state["y"]

    run.state = state
    return run


# ---------------------------------------------------------------------------------------------------
# Iterative-solver coupling: y feeds the next x, so every iteration needs the
# y shards on every rank -- the first place a per-iteration collective appears.
# ---------------------------------------------------------------------------------------------------
class PaddedLayout:
    """Vector layout for coupled iterations: `world` slots of `width` elements; slot g holds the entries
    cuts[g] .. cuts[g+1]-1 of the global vector and zero padding behind them.

    With x kept in this layout a rank's SpMV writes its y shard STRAIGHT into its slot of the next x and
    one in-place all-gather (equal counts, no staging copy, no unpadding) completes the vector on every
    rank.  Column indices of the row blocks are remapped once (`remap_columns`) so that they address the
    padded vector; nothing else about the CSR changes."""

    ALIGN = 64  # elements; keeps every slot 256-/512-byte aligned

    def __init__(self, cuts):
        self.cuts = np.asarray(cuts, dtype=np.int64)
        self.world = self.cuts.size - 1
        widest = int(np.max(np.diff(self.cuts))) if self.world else 0
        self.width = max(self.ALIGN, (widest + self.ALIGN - 1) // self.ALIGN * self.ALIGN)
        self.shift = np.arange(self.world, dtype=np.int64) * self.width - self.cuts[:-1]

    @property
    def padded_len(self) -> int:
        return self.world * self.width

    def owner(self, idx):
        return np.clip(np.searchsorted(self.cuts, idx, side="right") - 1, 0, self.world - 1)

    def remap_columns(self, col):
        col = np.asarray(col, dtype=np.int64)
        return (col + self.shift[self.owner(col)]).astype(np.int32)

    def to_padded(self, vec):
        """global vector (numpy) -> padded copy"""
        out = np.zeros(self.padded_len, dtype=np.asarray(vec).dtype)
        for g in range(self.world):
            lo, hi = int(self.cuts[g]), int(self.cuts[g + 1])
            out[g * self.width: g * self.width + hi - lo] = vec[lo:hi]
        return out

    def from_padded(self, padded):
        return np.concatenate([np.asarray(padded[g * self.width: g * self.width + int(self.cuts[g + 1] - self.cuts[g])])
                               for g in range(self.world)])

    def slot(self, t, rank: int):
        """view of slot `rank` of a padded torch tensor"""
        return t[rank * self.width: (rank + 1) * self.width]


def allgather_slots(x_padded, layout: PaddedLayout, rank: int):
    """The per-iteration collective: in-place all-gather of the `world` slots (RCCL over xGMI on the GPUs)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_gather_into_tensor(x_padded, layout.slot(x_padded, rank))
    return x_padded


class CoupledSpmv:
    """x_{k+1} = A x_k (optionally normalised) on a square matrix sharded by row blocks.

    `local_spmv(block, x_padded, y_slot)` multiplies this rank's block (columns already remapped to the
    padded layout, block.n == layout.padded_len) by the padded x and writes block.m results into y_slot."""

    def __init__(self, row_ptr, col, val, n: int, rank: int, world: int):
        m = int(np.asarray(row_ptr).size - 1)
        if m != n:
            raise ValueError("coupled iterations need a square matrix (y becomes the next x)")
        self.cuts = partition_rows_by_cost(row_ptr, world)
        self.layout = PaddedLayout(self.cuts)
        blk = extract_row_block(row_ptr, col, val, n, self.cuts, rank)
        self.block = RowBlock(rank, blk.row_lo, blk.row_hi, self.layout.padded_len, blk.row_ptr,
                              self.layout.remap_columns(blk.col), blk.val)
        self.rank, self.world = rank, world

    def step(self, local_spmv, x_in, x_out):
        """one iteration: x_out <- A x_in, complete on every rank when the call returns (stream-ordered)"""
        y_slot = self.layout.slot(x_out, self.rank)[: self.block.m]
        local_spmv(self.block, x_in, y_slot)
        return allgather_slots(x_out, self.layout, self.rank)

    def power_iteration(self, local_spmv, x0_padded, scratch, iters: int, normalise: bool = True):
        """`iters` coupled steps ping-ponging between the two padded buffers; returns (x, rayleigh) with
        rayleigh = <x_k, A x_k> / <x_k, x_k> of the last step (every rank holds the same numbers)."""
        import torch

        a, b = x0_padded, scratch
        rayleigh = None
        for _ in range(iters):
            self.step(local_spmv, a, b)
            rayleigh = torch.dot(a, b) / torch.dot(a, a)
            if normalise:
                b.mul_(1.0 / torch.linalg.vector_norm(b))
            a, b = b, a
        return a, rayleigh


    def power_iteration_graph(self, local_spmv, x0_padded, scratch, iters: int, normalise: bool = True):
        """The same iteration replayed from ONE hipGraph (torch.cuda.CUDAGraph is plumbing): two coupled steps
        (a -> b -> a: SpMV kernels, the in-place all-gather, dot, norm, scale) are captured once on a side stream and
        replayed iters/2 times -- no per-kernel host launch cost, no host synchronisation inside the loop.  `iters`
        must be even.  Returns (x, rayleigh) like power_iteration.  `local_spmv` must not have been used on another
        stream before (the handle is bound to the stream of its first call)."""
        import torch

        if iters % 2:
            raise ValueError("power_iteration_graph replays pairs of steps: iters must be even")
        a, b = x0_padded, scratch
        side = torch.cuda.Stream(device=a.device)
        side.wait_stream(torch.cuda.current_stream(a.device))
        ray = torch.zeros((), dtype=a.dtype, device=a.device)

        def pair():
            for src, dst in ((a, b), (b, a)):
                self.step(local_spmv, src, dst)
                ray.copy_(torch.dot(src, dst) / torch.dot(src, src))
                if normalise:
                    dst.mul_(1.0 / torch.linalg.vector_norm(dst))

        start = a.clone()
        with torch.cuda.stream(side):
            pair()  # warm-up outside the capture: binds the handle to `side`, creates its CSR5 form
            a.copy_(start)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pair()
        a.copy_(start)
        torch.cuda.synchronize(a.device)
        for _ in range(iters // 2):
            graph.replay()
        return a, ray


def hip_coupled_spmv(device, sigma: int = -1, mode: int = 1):
    """Factory for CoupledSpmv.step on a GPU: one CSR5 handle for the block, y written in place."""
    import torch

    from . import handle as H

    state = {}

    def run(block: RowBlock, x_padded, y_slot):
        if "A" not in state:
            rp = torch.from_numpy(block.row_ptr).to(device)
            ci = torch.from_numpy(block.col.astype(np.int32)).to(device)
            va = torch.from_numpy(block.val).to(device)
            A = H.anonymouslibHandle(block.m, block.n, dtype=str(block.val.dtype))
            _check(A.inputCSR(block.nnz, rp, ci, va), "inputCSR")
            _check(A.setSigma(sigma), "setSigma")
            _check(A.setSpmvMode(mode), "setSpmvMode")
            _check(A.setStream(torch.cuda.current_stream(device).cuda_stream), "setStream")
            # y becomes the next x: EVERY entry of the slot must be defined by the SpMV.  The reference semantics leave
            # rows without non-zeros untouched (csr5hip.h), which would keep values from two iterations ago in the
            # ping-pong buffer (R-MAT: half the rows), so the library is asked to store 0 there.
            _check(A.setZeroEmptyRows(1), "setZeroEmptyRows")
            _check(A.asCSR5(), "asCSR5")
            state.update(A=A, keep=(rp, ci, va))
        A = state["A"]
        _check(A.setX(x_padded), "setX")
        _check(A.spmv(1.0, y_slot), "spmv")
        return y_slot

    run.state = state
    return run
