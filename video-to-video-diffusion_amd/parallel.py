"""Multi-GPU execution of the sampling path (one process per GPU, torch.distributed over RCCL/xGMI).

Two ways the path shards:

* data parallel — volumes (or sliding-window patches) are independent: every rank runs the
  single-GPU engine on its own units, nothing is exchanged while computing, results are gathered
  at the end (`shard_units`, `gather_units`).  This is what `bench.py --gpus N` measures.

* depth sharding — one volume's depth D is cut into `world` equal slabs.  D is never resampled inside
  the U-Net / VAE and every non-pointwise conv has k_D = 3, stride 1, pad 1, so a slab plus one halo
  slice per side is exact.  Exchanges per network evaluation:
      - before each depth-3 conv: neighbour send/recv of the two boundary slices (zeros at the ends),
      - after each GroupNorm statistics pass: all-reduce of (sum, sumsq) per (sample, group), fp64,
      - TemporalAttention: all-reduce of the depth-sum S = sum_d x  (n, h, w, c) fp32
        (the reference's einsum makes attention a depth *sum*, so no all-to-all is needed).
  The engine emits these as `comm` ops inside the program (engine.Program.halo_exchange/all_reduce).

`DistComm` binds them to torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" for the CPU tests);
`LocalComm` runs `world` virtual ranks inside one process in lock-step so the sharded arithmetic can be
checked against the unsharded engine on a single GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from .lib import CtsiError


# ---- data parallel helpers --------------------------------------------------------------------------------
def shard_units(num_units: int, rank: int, world: int) -> range:
    """Contiguous, balanced partition of independent units (volumes, patches): rank r owns
    [r*q + min(r, rem), ...) — the first `rem` ranks take one extra unit."""
    q, rem = divmod(num_units, world)
    start = rank * q + min(rank, rem)
    return range(start, start + q + (1 if rank < rem else 0))


def gather_units(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """All-gather ragged batches along dim 0 (each rank contributes counts[rank] items)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)


def depth_slab(depth: int, rank: int, world: int) -> range:
    """Contiguous depth slab of rank r: balanced, the first depth % world ranks own one slice more (300 thin slices over 8
    GPUs = 38,38,38,38,37,37,37,37).  Every rank needs at least one slice."""
    if depth < world:
        raise CtsiError(f"depth sharding needs at least one slice per rank (depth {depth}, {world} ranks)")
    return shard_units(depth, rank, world)


# ---- communicators ---------------------------------------------------------------------------------------------
# One interface, three transports.  A call is ONE sync point of the sharded program:
#     exchange(rank, lo_own, hi_own, lo_halo, hi_halo, sums=None, f32=None, sptr=None)
#         lo_halo <- rank-1's hi_own, hi_halo <- rank+1's lo_own (zeros at the volume's two ends), and in the same sync
#         point `sums` (fp64 GroupNorm statistics) / `f32` (TemporalAttention depth sum) are summed over all ranks in
#         place.  Every part is optional (slices None = a pure all-reduce).
#     gather_depth(rank, slab)  -> the full tensor, slabs concatenated along depth (dim 2)
# `sptr` is the engine's HIP stream (only the RCCL transport needs it).
class RcclComm:
    """The C-ABI transport (csrc/comm.hip): RCCL calls issued by libctsi on the engine stream, one ncclGroup per sync
    point, stream-capture safe.  The 128-byte RCCL unique id is created on rank 0 and broadcast through the
    torch.distributed group the caller already has (any backend), so this works under torchrun as launched by
    bench.py: `RcclComm.from_process_group()`."""
    capturable = True

    def __init__(self, handle, rank: int, world: int):
        from .lib import get_lib
        self.lib = get_lib()
        self.handle = handle
        self.rank, self.world = rank, world

    @classmethod
    def from_process_group(cls, group=None, device=None):
        import ctypes as C
        import torch.distributed as dist
        from .lib import get_lib
        if not dist.is_initialized():
            raise CtsiError("RcclComm.from_process_group needs an initialised torch.distributed process group")
        lib = get_lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            lib.comm_unique_id(C.cast(buf, C.c_void_p))
        on_gpu = dist.get_backend(group) == "nccl"
        dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev if on_gpu else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        idb = (C.c_ubyte * 128)(*t.cpu().tolist())
        handle = C.c_void_p()
        lib.comm_init(C.byref(handle), C.cast(idb, C.c_void_p), rank, world)
        return cls(handle, rank, world)

    @classmethod
    def single(cls, with_rccl: bool = False):
        """World of one.  `with_rccl` builds a real one-rank RCCL communicator (exercises the binding on one GPU)."""
        import ctypes as C
        from .lib import get_lib
        lib = get_lib()
        handle = C.c_void_p()
        if with_rccl:
            buf = (C.c_ubyte * 128)()
            lib.comm_unique_id(C.cast(buf, C.c_void_p))
            lib.comm_init(C.byref(handle), C.cast(buf, C.c_void_p), 0, 1)
        else:
            lib.comm_init(C.byref(handle), None, 0, 1)
        return cls(handle, 0, 1)

    def exchange(self, rank, lo_own, hi_own, lo_halo, hi_halo, sums=None, f32=None, sptr=None):
        import ctypes as C
        p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        nbytes = 0 if lo_own is None else lo_own.numel() * lo_own.element_size()
        self.lib.halo_exchange_reduce(self.handle, p(lo_own), p(hi_own), p(lo_halo), p(hi_halo), nbytes, p(sums),
                                      0 if sums is None else sums.numel(), p(f32), 0 if f32 is None else f32.numel(),
                                      sptr)

    def gather_depth(self, rank, slab, sptr=None, counts=None):
        """`counts`: slices per rank when the slabs are ragged (every rank then sends max(counts) slices, zero padded)."""
        import ctypes as C
        dmax = slab.shape[2] if counts is None else max(counts)
        if slab.shape[2] != dmax:
            slab = torch.cat([slab, slab.new_zeros(slab.shape[:2] + (dmax - slab.shape[2],) + slab.shape[3:])], dim=2)
        slab = slab.contiguous()
        out = torch.empty((self.world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
        if sptr is None:
            sptr = C.c_void_p(torch.cuda.current_stream(slab.device).cuda_stream)
        self.lib.comm_allgather(self.handle, C.c_void_p(slab.data_ptr()), C.c_void_p(out.data_ptr()),
                                slab.numel() * slab.element_size(), sptr)
        return torch.cat([out[r][:, :, :(dmax if counts is None else counts[r])] for r in range(self.world)], dim=2)

    def __del__(self):
        try:
            if self.handle:
                self.lib.comm_destroy(self.handle)
        except Exception:
            pass


class DistComm:
    """The same sync points over torch.distributed (gloo for the CPU tests of the host logic; also usable with the
    nccl backend, uncaptured)."""
    capturable = False

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise CtsiError("DistComm needs an initialised torch.distributed process group")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    @staticmethod
    def _raw(t: torch.Tensor) -> torch.Tensor:
        return t.view(torch.int16) if t.dtype == torch.bfloat16 else t

    def exchange(self, rank, lo_own, hi_own, lo_halo, hi_halo, sums=None, f32=None, sptr=None):
        dist, ops = self.dist, []
        if lo_own is not None:
            if rank > 0:
                ops.append(dist.P2POp(dist.isend, self._raw(lo_own), rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, self._raw(lo_halo), rank - 1, self.group))
            else:
                lo_halo.zero_()
            if rank < self.world - 1:
                ops.append(dist.P2POp(dist.isend, self._raw(hi_own), rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, self._raw(hi_halo), rank + 1, self.group))
            else:
                hi_halo.zero_()
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for t in (sums, f32):
            if t is not None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        for req in reqs:
            req.wait()

    def all_reduce(self, rank: int, t: torch.Tensor):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def gather_depth(self, rank: int, slab: torch.Tensor, sptr=None, counts=None) -> torch.Tensor:
        dmax = slab.shape[2] if counts is None else max(counts)
        if slab.shape[2] != dmax:
            slab = torch.cat([slab, slab.new_zeros(slab.shape[:2] + (dmax - slab.shape[2],) + slab.shape[3:])], dim=2)
        outs = [torch.empty_like(slab) for _ in range(self.world)]
        self.dist.all_gather(outs, slab.contiguous(), group=self.group)
        return torch.cat([o[:, :, :(dmax if counts is None else counts[r])] for r, o in enumerate(outs)], dim=2)


class LocalComm:
    """`world` virtual ranks in one process, driven in lock-step (op i of rank 0, 1, ... then op i+1):
    each collective call parks its tensors; the last rank to arrive performs the exchange for all."""
    capturable = False

    def __init__(self, world: int):
        self.world = world
        self._pending: List[tuple] = []

    def _arrive(self, item) -> Optional[List[tuple]]:
        self._pending.append(item)
        if len(self._pending) < self.world:
            return None
        items, self._pending = sorted(self._pending, key=lambda it: it[0]), []
        return items

    def exchange(self, rank, lo_own, hi_own, lo_halo, hi_halo, sums=None, f32=None, sptr=None):
        items = self._arrive((rank, lo_own, hi_own, lo_halo, hi_halo, sums, f32))
        if items is None:
            return
        if items[0][1] is not None:
            for r, it in enumerate(items):      # (own and halo slices never alias: plain copies)
                if r > 0:
                    it[3].copy_(items[r - 1][2])
                else:
                    it[3].zero_()
                if r < self.world - 1:
                    it[4].copy_(items[r + 1][1])
                else:
                    it[4].zero_()
        for k in (5, 6):
            if items[0][k] is not None:
                total = items[0][k].clone()
                for it in items[1:]:
                    total += it[k]
                for it in items:
                    it[k].copy_(total)

    def all_reduce(self, rank: int, t: torch.Tensor):
        self.exchange(rank, None, None, None, None, sums=t)

    def gather_depth(self, rank: int, slab: torch.Tensor, sptr=None, counts=None):
        items = self._arrive((rank, slab))
        if items is None:
            return None
        return torch.cat([s for _, s in items], dim=2)


@dataclass
class ShardSpec:
    rank: int
    world: int
    comm: object          # RcclComm | DistComm | LocalComm
    depth_total: int
    overlap: bool = True  # plain halo exchanges run on a second stream under the interior slices of the consuming conv

    @property
    def depth_local(self) -> int:
        return len(depth_slab(self.depth_total, self.rank, self.world))

    @property
    def depth_start(self) -> int:
        return depth_slab(self.depth_total, self.rank, self.world).start

    @property
    def depth_counts(self) -> List[int]:
        return [len(depth_slab(self.depth_total, r, self.world)) for r in range(self.world)]


def run_lockstep(programs: Sequence, launches: int = 1):
    """Drive the programs of all virtual ranks op by op (LocalComm)."""
    nops = len(programs[0].ops)
    if any(len(p.ops) != nops for p in programs):
        raise CtsiError("virtual ranks built different programs")
    for _ in range(launches):
        for i in range(nops):
            for p in programs:
                p.ops[i]()


def allreduce_gradients(params, group=None, bucket_bytes: int = 256 << 20, average: bool = True):
    """Data-parallel gradient exchange for the training path: the .grad tensors of `params` are summed over
    the ranks in a few large flat buckets (xGMI rings are per-link bound: few big all-reduces beat many small
    ones; 265 M fp32 gradients = 1.06 GB = 4-5 buckets of 256 MB) and divided by the world size.  Works on
    any torch.distributed backend (nccl = RCCL on ROCm; gloo in the CPU tests).  Returns the bucket count."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and (size + nb > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nb
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        dist.all_reduce(flat, group=group)
        if average:
            flat /= world
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return len(buckets)
