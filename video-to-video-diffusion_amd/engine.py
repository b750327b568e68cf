"""Host-side execution engine: turns the reference's module trees (UNet3D, the VAE encoder and
decoder) into flat programs of libctsi launches on one HIP stream, optionally captured into a
hipGraph (one graph per denoising step).

PyTorch is used here for device memory, the stream object and parameter storage only; every
arithmetic op of the hot path is a libctsi kernel.  There is no CPU path: constructing a context
without a HIP device (or without libctsi.so) raises ``CtsiError``.

Data layout in HBM
  activations   bf16, channels-last (n, d, h, w, c), c contiguous; 16-byte (8-channel) accesses
  weights       bf16, re-laid out once per layer into the gather-GEMM image [class][cout_pad][K]
  GN statistics fp32 per-tile column sums -> fp64 (sum, sumsq) per (sample, group)
  sampler state fp32 NDHWC (z) + its bf16 copy inside the U-Net input tensor [z | cond]
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .lib import ConvDesc, ConvOut, CtsiError, get_lib

_CTX: Dict[int, "Ctx"] = {}

# Packed-weight cache ("weight pre-packing cache keyed by checkpoint hash").  The bf16 kernel-layout
# image of a layer's weights is content-addressed: key = (layout signature of the plan, 2 x 64-bit checksum of the fp32
# weight tensor the layer's weight_fn returns).  Every program built from the same weights -- the sampler program, the
# plain forward, other latent shapes, window batches, the depth-sharded variants: 265 M parameters = 0.53 GB of packed
# bf16 per program otherwise -- shares ONE image and packs it once.  Entries are immutable and die with their last user
# (weak values); a program whose weights changed looks up / packs new images and drops its captured graph.
_PACKED: Dict[int, "weakref.WeakValueDictionary"] = {}
# every live program (weak): check_device_errors re-zeroes their split-K hand-off workspaces after a reported device error
_LIVE_PROGRAMS: "weakref.WeakSet" = weakref.WeakSet()


_KEY_CHUNK = 1 << 22


def _content_key(wt: torch.Tensor) -> Tuple[int, int, int]:
    """Two independent 63-bit position-dependent checksums of the fp32 bit patterns + the element count.  Each element is
    mixed with its index through an odd-constant multiply and an xor-shift (wrapping int64 arithmetic), so neither sum is
    linear in the data nor periodic in the position; computed in 4 M-element chunks (16 MB of temporaries, not 3 x numel
    int64) and read back with ONE host sync per tensor."""
    xi = wt.reshape(-1).view(torch.int32)
    acc = torch.zeros(2, dtype=torch.int64, device=wt.device)
    for off in range(0, xi.numel(), _KEY_CHUNK):
        v = xi[off:off + _KEY_CHUNK].to(torch.int64) & 0xFFFFFFFF
        idx = torch.arange(off, off + v.numel(), device=wt.device, dtype=torch.int64)
        h = v + 0x632BE59BD9B4E019 + idx * -0x61C8864680B583EB      # odd (golden-ratio) constant; int64 wraps
        h = h ^ (h >> 29)
        h = h * -0x4B47D0C5B1A3F1B5
        h = h ^ (h >> 32)
        g = (v ^ (idx * 0x2545F4914F6CDD1D)) * -0x395B586CA42E166B
        g = g ^ (g >> 31)
        acc[0] += h.sum()
        acc[1] += g.sum()
    a = acc.tolist()
    return int(a[0]), int(a[1]), int(xi.numel())


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class Ctx:
    """Per-device engine context: the library handle and the engine's own HIP stream."""

    def __init__(self, device: torch.device):
        self.lib = get_lib()
        if not torch.cuda.is_available() or not self.lib.device_available():
            raise CtsiError("the HIP engine needs a ROCm device; none is visible to this process")
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.sptr = C.c_void_p(self.stream.cuda_stream)

    @staticmethod
    def get(device) -> "Ctx":
        device = torch.device(device)
        if device.type != "cuda":
            raise CtsiError(
                f"the HIP engine runs on ROCm devices only (got device '{device}'); there is no CPU path")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in _CTX:
            _CTX[idx] = Ctx(torch.device("cuda", idx))
        return _CTX[idx]

    def comm_stream_ptr(self) -> C.c_void_p:
        """The context's second HIP stream (created on first use): halo transfers that overlap interior compute."""
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(device=self.device)
        return C.c_void_p(self._comm_stream.cuda_stream)

    # make the engine stream wait for work queued on torch's current stream, and vice versa
    def enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def leave(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    @contextlib.contextmanager
    def scope(self):
        """Run a block on the engine stream, ordered after/before the caller's current stream."""
        self.enter()
        try:
            with torch.cuda.stream(self.stream):
                yield self
        finally:   # also on an exception: the caller's stream must still wait for what was queued
            self.leave()


class _Pool:
    """Tiny size-keyed buffer pool used while a program is being laid out (ops run in stream
    order, so a buffer may be handed out again as soon as its last reader has been emitted)."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[Tuple[int, torch.dtype], List[torch.Tensor]] = {}
        self.all: List[torch.Tensor] = []   # ops hold raw pointers: every buffer lives as long as the program
        self.total_bytes = 0

    def get(self, numel: int, dtype: torch.dtype) -> torch.Tensor:
        lst = self.free.get((numel, dtype))
        if lst:
            return lst.pop()
        self.total_bytes += numel * torch.empty(0, dtype=dtype).element_size()
        t = torch.empty(numel, dtype=dtype, device=self.device)
        self.all.append(t)
        return t

    def put(self, t: torch.Tensor):
        self.free.setdefault((t.numel(), t.dtype), []).append(t)


class Act:
    """A bf16 NDHWC activation buffer with its logical shape.  In depth-sharded programs (`halo` = 1)
    the buffer holds d + 2 slices: one halo slice below and above the rank's own d slices."""
    __slots__ = ("t", "n", "c", "d", "h", "w", "halo", "dirty", "grad")

    def __init__(self, t, n, c, d, h, w, halo=0):
        self.t, self.n, self.c, self.d, self.h, self.w = t, n, c, d, h, w
        self.halo = halo
        self.dirty = True  # halo slices stale (they are refreshed lazily, right before a depth-3 conv)
        self.grad = None   # gradient Act (training programs only)

    @property
    def vox(self):
        return self.d * self.h * self.w

    @property
    def slice_elems(self):
        return self.h * self.w * self.c

    @property
    def ip(self) -> C.c_void_p:
        """pointer to the first own (interior) slice"""
        return C.c_void_p(self.t.data_ptr() + self.halo * self.slice_elems * 2)

    @property
    def fp(self) -> C.c_void_p:
        """pointer to the whole buffer (lower halo slice first)"""
        return C.c_void_p(self.t.data_ptr())


class Program:
    """A flat list of libctsi launches plus the buffers they touch."""

    def __init__(self, ctx: Ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        self.ops: List[Callable[[], None]] = []
        self.op_meta: List[Tuple[str, float, str]] = []
        self.op_bytes: List[float] = []
        self.op_alg_bytes: List[float] = []   # convs: algorithmic HBM bytes (inputs + outputs + weights, each once)
        self.pool = _Pool(ctx.device)
        self.keep: List[object] = []       # tensors / ctypes structs that must outlive the ops
        self.plans: List[C.c_void_p] = []  # conv plan handles (destroyed with the program)
        self.pack_fns: List[Callable[[], None]] = []
        self.flops = 0.0
        self.conv_flops: List[Tuple[str, float]] = []
        self.gn_slots = 0
        self._gn_sums: Optional[torch.Tensor] = None
        self._colsum: Optional[torch.Tensor] = None
        self._colsum_need = 0
        self._gn_need: List[Tuple[int, int]] = []
        self.graph = None
        self._params: List[torch.Tensor] = []
        self._versions: Tuple[int, ...] = ()
        self.shard = None  # parallel.ShardSpec for depth-sharded programs
        self.weight_cache = True    # share packed weight images between programs (training programs repack in place)
        self._weights_moved = False
        self.pack_stats = dict(packed=0, shared=0)
        # fast re-pack (see fast_repack): what every pack function reads and writes, so that the launches can be replayed from
        # prebuilt pointer tables without touching torch
        self._f32_meta: List[dict] = []     # dev_f32 buffers: dict(buf, make, parts, scale, fn)
        self._pack_meta: List[dict] = []    # conv weight images: dict(plan, weight_fn, holder, fn)
        self._fast = None
        self._sk_workspaces: List[torch.Tensor] = []   # split-K hand-off workspaces (tickets / flags + parked partial sums)
        _LIVE_PROGRAMS.add(self)

    def _emit(self, fn: Callable[[], None], name: str = "op", flops: float = 0.0, kernel: str = "", nbytes: float = 0.0,
              alg_bytes: float = 0.0):
        """`nbytes`: algorithmic HBM bytes of an HBM-bound op (what bench.py divides by the launch time for GB/s)."""
        self.ops.append(fn)
        self.op_meta.append((name, flops, kernel))
        self.op_bytes.append(float(nbytes))
        self.op_alg_bytes.append(float(alg_bytes))

    # ---- buffers -------------------------------------------------------------------------------------
    def act(self, n, c, d, h, w, halo: Optional[int] = None) -> Act:
        if halo is None:
            halo = 1 if self.shard is not None else 0
        return Act(self.pool.get(n * c * (d + 2 * halo) * h * w, torch.bfloat16), n, c, d, h, w, halo)

    # ---- depth-sharding collectives (no-ops in single-GPU programs) -----------------------------------
    # A sharded program is ~64 sync points per U-Net evaluation (was 119): the GroupNorm statistics of a tensor travel in
    # the SAME sync point as the raw boundary slices of that tensor (`sync_stats_and_halos`); the normalisation is then
    # applied to the own slices AND the received halo slices (`ext` ranges below), so the tensor a depth-3 conv consumes
    # never needs a second exchange.  Only tensors no GroupNorm follows (conv_in / Downsample / Upsample outputs, the
    # sampler's updated input) use the plain `halo_exchange`.
    def _slices(self, a: Act):
        se, d = a.slice_elems, a.d
        buf = a.t
        return buf[se:2 * se], buf[d * se:(d + 1) * se], buf[0:se], buf[(d + 1) * se:(d + 2) * se]

    def halo_exchange(self, a: Act):
        """Refresh a's two halo slices from the depth neighbours (zeros at the volume's ends)."""
        if self.shard is None or not a.halo or not a.dirty:
            return
        a.dirty = False
        lo_own, hi_own, lo_halo, hi_halo = self._slices(a)
        spec, sptr = self.shard, self.ctx.sptr

        def run():
            spec.comm.exchange(spec.rank, lo_own, hi_own, lo_halo, hi_halo, sptr=sptr)

        self._emit(run, "halo.exchange", 0.0, "comm")

    def sync_stats_and_halos(self, a: Optional[Act], slot: Optional[int], nvals: int = 0,
                             f32: Optional[torch.Tensor] = None, name: str = "gn.sync"):
        """ONE sync point: all-reduce of the fp64 statistics slot (and of `f32`, the attention depth sum) together with
        the neighbour exchange of a's raw boundary slices (a = None: statistics only)."""
        if self.shard is None:
            return
        spec, sptr, prog = self.shard, self.ctx.sptr, self
        sl = self._slices(a) if (a is not None and a.halo) else (None, None, None, None)
        if a is not None and a.halo:
            a.dirty = False
        holder = {}

        def run():
            sums = None
            if slot is not None:
                if "v" not in holder:
                    holder["v"] = prog._gn_sums[slot:slot + nvals]
                sums = holder["v"]
            spec.comm.exchange(spec.rank, sl[0], sl[1], sl[2], sl[3], sums=sums, f32=f32, sptr=sptr)

        self._emit(run, name, 0.0, "comm")

    def ext(self, a: Act) -> Tuple[int, int]:
        """(slices below the own ones, total depth) of the range an elementwise op covers on `a`: the own slices plus
        the halo slices that exist -- none on a single GPU, none towards a volume end (those stay zero: the conv's
        zero padding)."""
        if self.shard is None or not a.halo:
            return 0, a.d
        lo = 1 if self.shard.rank > 0 else 0
        hi = 1 if self.shard.rank < self.shard.world - 1 else 0
        return lo, a.d + lo + hi

    def ext_ptr(self, a: Act, lo: int) -> C.c_void_p:
        return C.c_void_p(a.ip.value - lo * a.slice_elems * 2)

    def zero_end_halos(self, a: Act):
        """A NEW buffer written over its `ext` range: the halo slice towards a volume end is not covered and must read
        as zeros (emitted on every rank -- the lock-step test driver needs equal op lists; a no-op on inner ranks)."""
        if self.shard is None or not a.halo:
            return
        spec, lib, sptr = self.shard, self.lib, self.ctx.sptr
        _, _, lo_halo, hi_halo = self._slices(a)
        nbytes = a.slice_elems * 2

        def run():
            if spec.rank == 0:
                lib.memset_async(_ptr(lo_halo), 0, nbytes, sptr)
            if spec.rank == spec.world - 1:
                lib.memset_async(_ptr(hi_halo), 0, nbytes, sptr)

        self._emit(run, "halo.zero_ends")

    def release(self, a: Act):
        self.pool.put(a.t)

    def _conv_workspace(self, plan) -> int:
        """Device scratch of a split-K conv plan (ctsi_conv_plan_workspace_bytes): one zero-initialised buffer per layer, kept
        for the program's lifetime (the kernel leaves its hand-off tickets reset after every launch)."""
        nbytes = self.lib.conv_plan_workspace_bytes(plan)
        if not nbytes:
            return 0
        ws = self.persistent((nbytes,), torch.uint8, zero=True)
        self._sk_workspaces.append(ws)
        return ws.data_ptr()

    def check_errors(self):
        """Programs with split-K layers: raise if one of their hand-offs failed (see check_device_errors).  The samplers
        check once per sample() whatever the program holds; the VAE legs and the training step call this, which costs a
        stream synchronisation only when the program has such a layer at all."""
        if self._sk_workspaces:
            check_device_errors(self.ctx)

    def persistent(self, shape, dtype, zero=False) -> torch.Tensor:
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.ctx.device)
        self.keep.append(t)
        return t

    # ---- parameters ------------------------------------------------------------------------------------
    def track(self, *params: torch.Tensor):
        self._params.extend(params)

    def dev_f32(self, make: Callable[[], torch.Tensor], parts: Optional[Sequence[Callable[[], torch.Tensor]]] = None,
                scale: float = 1.0) -> torch.Tensor:
        """A persistent fp32 device copy of a (derived) parameter, refreshed by repack().  `parts` (optional) says how the
        value is made of parameter storage -- buf = scale * concat(part() for part in parts), every part a contiguous view
        of a parameter -- so that fast_repack() can refresh it from a pointer table; a `make()` that itself returns such a
        view needs no `parts`."""
        src = make().detach()
        buf = torch.empty(src.shape, dtype=torch.float32, device=self.ctx.device)
        self.keep.append(buf)

        def refresh():
            buf.copy_(make().detach().to(device=self.ctx.device, dtype=torch.float32))

        self.pack_fns.append(refresh)
        self._f32_meta.append(dict(buf=buf, make=make, parts=parts, scale=float(scale), fn=refresh))
        return buf

    # ---- fast re-pack after an optimizer step (optim.FusedAdamW) -------------------------------------------------------
    # The generic repack() goes through torch for every operand (600 small copies and pack launches for the training
    # program: ~6 ms, all of it host time).  After an optimizer step nothing but the VALUES changed, so the same device
    # work can be issued from tables built once: ONE ctsi_copy_scale_multi launch for all small fp32 operands and one
    # ctypes call per conv image (+ its dgrad re-layout), with every pointer precomputed.  Only programs that own their
    # weight images (weight_cache False: the training program) take this path; operands it cannot express (values that are
    # not contiguous views of parameter storage) keep their generic refresh function.
    def _param_view(self, t: torch.Tensor) -> bool:
        """A contiguous fp32 view INTO THE STORAGE OF A TRACKED PARAMETER: only such an address is still the value's home
        after the next optimizer step (a temporary -- cat, scaled copy, matmul result -- passes every other test, and its
        freed address would be baked into the pointer table and the recorded graph)."""
        if not (torch.is_tensor(t) and t.is_cuda and t.device == self.ctx.device and t.dtype == torch.float32
                and t.is_contiguous() and t.numel() > 0):
            return False
        homes = getattr(self, "_param_storages", None)
        if homes is None or self._param_storages_n != len(self._params):
            homes = self._param_storages = {p.untyped_storage().data_ptr() for p in self._params if torch.is_tensor(p)}
            self._param_storages_n = len(self._params)
        return t.untyped_storage().data_ptr() in homes

    def _build_fast(self):
        import struct
        if self._fast and self._fast.get("graph") is not None:      # (a parameter moved: the recorded pointers are stale)
            self.lib.graph_destroy(self._fast["graph"])
            self._fast["graph"] = None
        segs, slow, packs = [], [], []
        for ent in self._f32_meta:
            parts = ent["parts"] if ent["parts"] is not None else [ent["make"]]
            vals = [pt() for pt in parts]
            if all(self._param_view(v.detach()) for v in vals) and sum(v.numel() for v in vals) == ent["buf"].numel():
                off = 0
                for v in vals:
                    segs.append((v.data_ptr(), ent["buf"].data_ptr() + 4 * off, v.numel(), ent["scale"]))
                    off += v.numel()
            else:
                slow.append(ent["fn"])
        lib, sptr, dev = self.lib, self.ctx.sptr, self.ctx.device
        for ent in self._pack_meta:
            holder, plan, wfn = ent["holder"], ent["plan"], ent["weight_fn"]
            fast = getattr(wfn, "fast_layout", None)
            if holder[0] is None:
                slow.append(ent["fn"])
                continue
            dst = C.c_void_p(holder[0].data_ptr())
            if fast is not None:      # data-gradient image: flipped / transposed re-layout of a parameter view, then the pack
                src_fn, co_w, ci_w, T, coff, cnt = fast
                src = src_fn().detach()
                if not self._param_view(src):
                    slow.append(ent["fn"])
                    continue
                tmp = torch.empty((cnt * co_w * T,), dtype=torch.float32, device=dev)
                self.keep.append(tmp)
                sp, tp = C.c_void_p(src.data_ptr()), C.c_void_p(tmp.data_ptr())
                packs.append((lib.weight_dgrad_layout, (sp, tp, co_w, ci_w, T, coff, cnt, sptr)))
                packs.append((lib.conv_plan_pack_weights, (plan, tp, dst, sptr)))
                continue
            wt = wfn().detach()
            if self._param_view(wt):
                packs.append((lib.conv_plan_pack_weights, (plan, C.c_void_p(wt.data_ptr()), dst, sptr)))
            else:
                slow.append(ent["fn"])
        # refresh functions registered outside the two tables (an attention block's folded matrix, the packs of an overlapped
        # depth-sharded conv) have no fast form: they run as they are, and keep the step out of the recorded graph
        known = {id(ent["fn"]) for ent in self._f32_meta} | {id(ent["fn"]) for ent in self._pack_meta}
        slow.extend(fn for fn in self.pack_fns if id(fn) not in known and fn not in slow)
        seg_bytes = b"".join(struct.pack("<QQqfi", s_, d_, n_, sc, 0) for (s_, d_, n_, sc) in segs)
        pieces = [(i, q) for i, (_, _, n_, _) in enumerate(segs) for q in range((n_ + 4095) // 4096)]
        pc_bytes = b"".join(struct.pack("<ii", i, q) for (i, q) in pieces)
        seg_t = torch.frombuffer(bytearray(seg_bytes or b"\0" * 32), dtype=torch.uint8).to(dev)
        pc_t = torch.frombuffer(bytearray(pc_bytes or b"\0" * 8), dtype=torch.uint8).to(dev)
        self._fast = dict(segs=seg_t, pieces=pc_t, npieces=len(pieces), packs=packs, slow=slow,
                          ptrs=tuple(p.data_ptr() for p in self._params), nseg=len(segs))

    def fast_repack(self):
        """Re-pack after the parameter VALUES changed in place (same storage): see the comment above.  Falls back to the
        generic repack() for programs that share weight images, and rebuilds its tables when a parameter moved."""
        if self.weight_cache:
            return self.repack()
        if self._fast is None or self._fast["ptrs"] != tuple(p.data_ptr() for p in self._params):
            with torch.cuda.stream(self.ctx.stream):
                self._build_fast()
        f = self._fast
        with torch.cuda.stream(self.ctx.stream):
            if f.get("graph") is None and not f["slow"] and not os.environ.get("CTSI_NO_REPACK_GRAPH"):
                # every launch of the list has constant arguments: record it once, replay it as ONE hipGraph afterwards
                # (240 launches for the training program: ~1 ms of ctypes calls otherwise)
                self.lib.graph_begin_capture(self.ctx.sptr)
                ok = False
                try:
                    self.lib.copy_scale_multi(_ptr(f["segs"]), _ptr(f["pieces"]), f["npieces"], self.ctx.sptr)
                    for fn, args in f["packs"]:
                        fn(*args)
                    ok = True
                finally:
                    g = C.c_void_p()
                    self.lib.graph_end_capture(self.ctx.sptr, C.byref(g))
                    if not ok and g:          # a launch failed under capture: the partial graph must not be replayed
                        self.lib.graph_destroy(g)
                f["graph"] = g
            if f.get("graph") is not None:
                self.lib.graph_launch(f["graph"], self.ctx.sptr)
            else:
                self.lib.copy_scale_multi(_ptr(f["segs"]), _ptr(f["pieces"]), f["npieces"], self.ctx.sptr)
                for fn, args in f["packs"]:
                    fn(*args)
                for fn in f["slow"]:
                    fn()
        self._versions = self._fingerprint()

    def _fingerprint(self):
        """What the packed weights were made from: the tracked Parameter objects' in-place version counters and
        storage addresses, plus (when the program knows its module) the objects and addresses the module's
        parameter names resolve to NOW -- so optimizer steps, `load_state_dict` (in place or assign=True) and a
        replaced `module.weight` are all seen.  Writes through `p.data` / raw pointers bump nothing torch can
        observe: call `model.invalidate_engine_cache()` after those."""
        fp = [(p._version, p.data_ptr()) for p in self._params]
        mod = getattr(self, "_track_module", None)
        if mod is not None:
            fp.extend((id(p), p._version, p.data_ptr()) for p in mod.parameters())
        return tuple(fp)

    def track_module(self, module: nn.Module):
        self._track_module = module
        self.track(*list(module.parameters()))

    def repack(self):
        with torch.cuda.stream(self.ctx.stream):
            for fn in self.pack_fns:
                fn()
        self._versions = self._fingerprint()
        if self._weights_moved:
            # content-addressed weight images: changed weights live in NEW buffers, so a captured graph (which holds the
            # old pointers) is dropped; the samplers re-capture on their next call, launch() runs eagerly until then
            self._weights_moved = False
            if self.graph is not None:
                self.lib.graph_destroy(self.graph)
                self.graph = None

    def ensure_fresh(self):
        if self._fingerprint() != self._versions:
            self.repack()

    def needs_rebuild(self) -> bool:
        return False

    # ---- conv ----------------------------------------------------------------------------------------------
    def conv(self, name: str, weight_fn, bias_fn, x1: Act, x2: Optional[Act], *, transposed=False,
             k=(3, 3, 3), s=(1, 1), p=(1, 1, 1), cout: int, cin_w: Optional[int] = None,
             out: Optional[Act] = None, want_stats=False, f32_out: Optional[torch.Tensor] = None,
             f32_strides: Optional[Sequence[int]] = None, act: int = 0, fuse_gn=None, ext_out: bool = False,
             norm_in=None):
        """Emit one convolution.  weight_fn/bias_fn return the *current* fp32 parameter tensors
        (possibly derived, e.g. scaled or pre-multiplied).  Returns (out_act, stats_handle).
        `fuse_gn` = (h: Act, slot, gn: nn.GroupNorm, silu: bool): the epilogue stores silu?(gn(h) + conv result)
        instead of the conv result (ctsi_conv_out.gn_x; h may be the output buffer itself).
        `norm_in` = (slot, gn: nn.GroupNorm, silu: bool, tb) with tb = None or (tbias, tbias_off, tbias_stride, step_ptr):
        x1 is the RAW output of the previous conv and this conv's input is silu?(gn(x1)) + tbias: the normalisation is
        emitted as its own in-place pass first (depth-sharded: its statistics travel with the raw boundary slices in one
        sync point), so x1 must not be used raw afterwards.  (A normalise-on-load variant of the 512-voxel kernel that
        applied it to the staged tile in LDS was bit-identical but slower: csrc/experiments/conv3_halo_m512.hip.)"""
        lib = self.lib
        nin_synced = False
        if norm_in is not None and self.shard is not None and x1.halo:
            # depth-sharded: statistics + RAW boundary slices in one sync point, whichever way the normalisation happens
            self.sync_stats_and_halos(x1, norm_in[0], x1.n * norm_in[1].num_groups * 2)
            nin_synced = True
        deep = k[0] > 1 and x1.halo == 1   # depth taps read the halo slices: "valid" conv along depth
        if x1.halo and x1.n != 1:
            raise CtsiError("a depth-sharded program holds one volume (batches run volume by volume, see "
                            "sampler.run_sampler_sharded)")
        if (deep and self.shard is not None and getattr(self.shard, "overlap", True)
                and (self.shard.world > 1 or self.shard.overlap == "force")
                and (x1.dirty or (x2 is not None and x2.dirty)) and min(self.shard.depth_counts) >= 3 and not transposed and tuple(s) == (1, 1)
                and f32_out is None and fuse_gn is None and norm_in is None and not ext_out and out is None and act == 0):
            return self._conv_overlapped(name, weight_fn, bias_fn, x1, x2, k, p, cout, cin_w, want_stats)
        if deep:
            self.halo_exchange(x1)
            if x2 is not None:
                self.halo_exchange(x2)
        # a pointwise conv whose inputs carry valid halo slices may cover them too (`ext_out`): its output then has valid
        # halos without an exchange of its own (the fused residual tail of a sharded ResBlock)
        ext_lo, ext_d = (0, x1.d)
        if ext_out:
            if deep or x1.dirty or (x2 is not None and x2.dirty):
                raise CtsiError("internal: ext_out needs a pointwise conv over inputs with valid halo slices")
            ext_lo, ext_d = self.ext(x1)
        di = x1.d + 2 if deep else ext_d
        desc = ConvDesc(int(transposed), k[0], k[1], k[2], s[0], s[1], p[0], p[1], p[2], x1.n, x1.c,
                        0 if x2 is None else x2.c, cout, di, x1.h, x1.w, 1 if deep else 0)
        plan = C.c_void_p()
        lib.conv_plan_create(C.byref(plan), C.byref(desc))
        self.plans.append(plan)
        if cin_w is not None:
            lib.conv_plan_set_weight_cin(plan, cin_w)
        if fuse_gn is not None and tuple(k) == (1, 1, 1) and f32_out is None and not want_stats and act == 0:
            # the ResBlock tails are HBM-bound passes: the streaming kernel where the layer qualifies (csrc/conv1_stream.hip)
            lib.conv_plan_set_stream_tail(plan, 1)
        if norm_in is not None:
            nslot, ngn, nsilu, ntb = norm_in      # normalise in place, then a plain conv
            kw_tb = {} if ntb is None else dict(tbias=ntb[0], tbias_off=ntb[1], tbias_stride=ntb[2], step_ptr=ntb[3])
            self.gn_apply(x1, nslot, ngn, silu_pre=nsilu, out=x1, synced=nin_synced, **kw_tb)
            norm_in = None
        do, ho, wo = C.c_int(), C.c_int(), C.c_int()
        lib.conv_plan_out_dims(plan, C.byref(do), C.byref(ho), C.byref(wo))
        do, ho, wo = do.value, ho.value, wo.value
        wbytes = lib.conv_plan_weight_bytes(plan)
        bias = (self.dev_f32(bias_fn, parts=getattr(bias_fn, "parts", None), scale=getattr(bias_fn, "scale", 1.0))
                if bias_fn is not None else None)
        sptr = self.ctx.sptr
        bm, bn, mode = C.c_int(), C.c_int(), C.c_int()
        lib.conv_plan_config(plan, C.byref(bm), C.byref(bn), C.byref(mode))
        layout = "gather" if mode.value in (0, 2) else mode.value      # the packed image depends on the kernel family,
        sig = (layout, int(transposed), tuple(k), tuple(s), x1.c, 0 if x2 is None else x2.c, cout, cin_w,   # not the shape
               lib.conv_plan_cout_pad(plan), wbytes, bn.value, bm.value, bool(lib.conv_plan_workspace_bytes(plan)))   # (k32: the
                                                                                   # epilogue form, hence the cout order, goes by tile / split-K)
        holder: List[Optional[torch.Tensor]] = [None]
        prog = self

        def pack():
            wt = weight_fn().detach().to(device=prog.ctx.device, dtype=torch.float32).contiguous()
            if prog.weight_cache:
                cache = _PACKED.setdefault(prog.ctx.device.index, weakref.WeakValueDictionary())
                key = (sig, _content_key(wt))
                t = cache.get(key)
                if t is None:
                    t = torch.empty(wbytes, dtype=torch.uint8, device=prog.ctx.device)
                    lib.conv_plan_pack_weights(plan, _ptr(wt), _ptr(t), sptr)
                    cache[key] = t
                    prog.pack_stats["packed"] += 1
                else:
                    prog.pack_stats["shared"] += 1
            else:
                t = holder[0] if holder[0] is not None else torch.empty(wbytes, dtype=torch.uint8, device=prog.ctx.device)
                lib.conv_plan_pack_weights(plan, _ptr(wt), _ptr(t), sptr)
            wt.record_stream(prog.ctx.stream)
            if holder[0] is not t:
                prog._weights_moved = prog._weights_moved or holder[0] is not None
                holder[0] = t

        self.pack_fns.append(pack)
        self._pack_meta.append(dict(plan=plan, weight_fn=weight_fn, holder=holder, fn=pack))
        fl = lib.conv_plan_flops(plan)
        self.flops += fl
        self.conv_flops.append((name, fl))

        stats = None
        colsum_ptr = 0
        if want_stats:
            tiles = lib.conv_plan_tiles(plan)
            cpad = lib.conv_plan_cout_pad(plan)
            self._colsum_need = max(self._colsum_need, 2 * tiles * cpad)
            stats = dict(tps=lib.conv_plan_tiles_per_sample(plan), cpad=cpad,
                         nclass=4 if transposed else 1)
        co = ConvOut()
        co.workspace = self._conv_workspace(plan)
        if f32_out is not None:
            co.y = f32_out.data_ptr()
            co.mode = 1
            co.sn, co.sc, co.sd, co.sh, co.sw = [int(v) for v in f32_strides]
            out_act = None
        else:
            if out is None:
                out = self.act(x1.n, cout, do - (ext_d - x1.d), ho, wo, halo=x1.halo)
            out.dirty = not ext_out
            co.y = self.ext_ptr(out, ext_lo).value
            co.mode = 0
            co.cout_stride = out.c
            co.c_off = 0
            out_act = out
        co.act = act
        if fuse_gn is not None:
            gh, gslot, gmod, gsilu = fuse_gn
            if f32_out is not None or want_stats or (gh.n, gh.c, gh.d, gh.h, gh.w) != (x1.n, cout, out.d, ho, wo):
                raise CtsiError("internal: fused GroupNorm tail needs a bf16 output of the normalised tensor's shape")
            ggamma = self.dev_f32(lambda: gmod.weight)
            gbeta = self.dev_f32(lambda: gmod.bias)
            d_stat = self.shard.depth_total if (self.shard is not None and gh.halo) else gh.d
            co.gn_x = self.ext_ptr(gh, ext_lo).value
            co.gn_gamma, co.gn_beta = ggamma.data_ptr(), gbeta.data_ptr()
            co.gn_groups, co.gn_eps = gmod.num_groups, float(gmod.eps)
            co.gn_count = (cout // gmod.num_groups) * d_stat * gh.h * gh.w
            co.gn_silu = int(gsilu)
        self.keep.append(co)
        x1p = x1.fp if deep else self.ext_ptr(x1, ext_lo)
        x2p = C.c_void_p(0) if x2 is None else (x2.fp if deep else self.ext_ptr(x2, ext_lo))
        bp = _ptr(bias)

        gn_slot = fuse_gn[1] if fuse_gn is not None else None

        def run():
            co.colsum = prog._colsum.data_ptr() if want_stats else 0
            if gn_slot is not None:
                co.gn_sums = prog._gn_sums.data_ptr() + gn_slot * 8
            lib.conv_fwd(plan, x1p, x2p, _ptr(holder[0]), bp, C.byref(co), sptr)

        form = "t" if transposed else ("d" if tuple(s) == (2, 2) else "")        # t: ConvTranspose form, d: Downsample form,
        kernel = "conv_mfma_%dx%d_m%d%s%s" % (bm.value, bn.value, mode.value, form if mode.value == 9 else "",
                                              "s" if co.workspace else "")      # s: split-K form
        cin_all = x1.c + (0 if x2 is None else x2.c)
        alg = (2.0 * x1.n * di * x1.h * x1.w * cin_all + float(wbytes)
               + (4.0 if f32_out is not None else 2.0) * x1.n * do * ho * wo * cout
               + (2.0 * x1.n * do * ho * wo * cout if fuse_gn is not None else 0.0))
        self._emit(run, name, fl, kernel, alg_bytes=alg)
        return out_act, stats

    def _conv_overlapped(self, name, weight_fn, bias_fn, x1: Act, x2: Optional[Act], k, p, cout, cin_w, want_stats):
        """A depth-3 stride-1 conv whose input still needs its plain halo exchange (tensors no GroupNorm follows: conv_in /
        Downsample / Upsample outputs, the sampler's updated input): the transfer runs on the context's second stream while
        the INTERIOR output slices -- which read own slices only -- are computed; the two BOUNDARY slices follow the join
        (SURVEY section 7-10).  Three launches of the same layer on three views of the same buffers:
            interior  input = the d own slices taken as a halo'd tensor   -> output slices 1 .. d-2
            lower     input = [lo halo, own 0, own 1]                     -> output slice 0
            upper     input = [own d-2, own d-1, hi halo]                 -> output slice d-1"""
        lib, sptr, prog, ctx = self.lib, self.ctx.sptr, self, self.ctx
        spec = self.shard
        cs = ctx.comm_stream_ptr()
        ev_fork, ev_join = C.c_void_p(), C.c_void_p()
        lib.event_create(C.byref(ev_fork))
        lib.event_create(C.byref(ev_join))
        self._events = getattr(self, "_events", []) + [ev_fork, ev_join]     # destroyed with the program
        todo = [a for a in (x1, x2) if a is not None and a.halo and a.dirty]
        sl = [self._slices(a) for a in todo]
        for a in todo:
            a.dirty = False

        def run_fork():
            lib.event_record(ev_fork, sptr)                 # the producer of the boundary slices is done ...
            lib.stream_wait_event(cs, ev_fork)              # ... before the transfer starts, on the second stream
            for (lo_own, hi_own, lo_halo, hi_halo) in sl:
                spec.comm.exchange(spec.rank, lo_own, hi_own, lo_halo, hi_halo, sptr=cs)
            lib.event_record(ev_join, cs)

        self._emit(run_fork, "halo.exchange.async", 0.0, "comm")
        d, se_in1 = x1.d, x1.slice_elems * 2
        se_in2 = 0 if x2 is None else x2.slice_elems * 2
        out = self.act(x1.n, cout, d, x1.h, x1.w, halo=x1.halo)
        out.dirty = True
        se_out = out.slice_elems * 2
        bias = (self.dev_f32(bias_fn, parts=getattr(bias_fn, "parts", None), scale=getattr(bias_fn, "scale", 1.0))
                if bias_fn is not None else None)
        parts, col_off = [], 0
        views = [("interior", d, x1.ip.value, 0 if x2 is None else x2.ip.value, out.ip.value + se_out),
                 ("lower", 3, x1.fp.value, 0 if x2 is None else x2.fp.value, out.ip.value),
                 ("upper", 3, x1.ip.value + (d - 2) * se_in1, 0 if x2 is None else x2.ip.value + (d - 2) * se_in2,
                  out.ip.value + (d - 1) * se_out)]
        for vi, (tag, di, p1, p2, py) in enumerate(views):
            if vi == 1:
                def run_join():
                    lib.stream_wait_event(sptr, ev_join)    # halos have landed: the boundary slices may be computed
                self._emit(run_join, "halo.join", 0.0, "comm.join")
            desc = ConvDesc(0, k[0], k[1], k[2], 1, 1, p[0], p[1], p[2], x1.n, x1.c, 0 if x2 is None else x2.c, cout, di,
                            x1.h, x1.w, 1)
            plan = C.c_void_p()
            lib.conv_plan_create(C.byref(plan), C.byref(desc))
            self.plans.append(plan)
            if cin_w is not None:
                lib.conv_plan_set_weight_cin(plan, cin_w)
            wbytes = lib.conv_plan_weight_bytes(plan)
            bm, bn, mode = C.c_int(), C.c_int(), C.c_int()
            lib.conv_plan_config(plan, C.byref(bm), C.byref(bn), C.byref(mode))
            layout = "gather" if mode.value in (0, 2) else mode.value
            sig = (layout, 0, tuple(k), (1, 1), x1.c, 0 if x2 is None else x2.c, cout, cin_w, lib.conv_plan_cout_pad(plan),
                   wbytes, bn.value, bm.value, bool(lib.conv_plan_workspace_bytes(plan)))
            holder: List[Optional[torch.Tensor]] = [None]

            def pack(plan=plan, sig=sig, wbytes=wbytes, holder=holder):
                wt = weight_fn().detach().to(device=prog.ctx.device, dtype=torch.float32).contiguous()
                cache = _PACKED.setdefault(prog.ctx.device.index, weakref.WeakValueDictionary())
                key = (sig, _content_key(wt))
                t = cache.get(key)
                if t is None:
                    t = torch.empty(wbytes, dtype=torch.uint8, device=prog.ctx.device)
                    lib.conv_plan_pack_weights(plan, _ptr(wt), _ptr(t), sptr)
                    cache[key] = t
                    prog.pack_stats["packed"] += 1
                else:
                    prog.pack_stats["shared"] += 1
                wt.record_stream(prog.ctx.stream)
                if holder[0] is not t:
                    prog._weights_moved = prog._weights_moved or holder[0] is not None
                    holder[0] = t

            self.pack_fns.append(pack)
            fl = lib.conv_plan_flops(plan)
            self.flops += fl
            self.conv_flops.append((name + "." + tag, fl))
            co = ConvOut()
            co.workspace = self._conv_workspace(plan)
            co.y, co.mode, co.cout_stride, co.c_off = py, 0, cout, 0
            self.keep.append(co)
            off = col_off
            if want_stats:
                tiles, cpad = lib.conv_plan_tiles(plan), lib.conv_plan_cout_pad(plan)
                parts.append(dict(tps=lib.conv_plan_tiles_per_sample(plan), cpad=cpad, nclass=1, off=off))
                col_off += 2 * tiles * cpad
            bp = _ptr(bias)

            def run(plan=plan, co=co, p1=p1, p2=p2, holder=holder, off=off):
                co.colsum = (prog._colsum.data_ptr() + off * 4) if want_stats else 0
                lib.conv_fwd(plan, C.c_void_p(p1), C.c_void_p(p2), _ptr(holder[0]), bp, C.byref(co), sptr)

            self._emit(run, name + "." + tag, fl, "conv_mfma_%dx%d_m%d" % (bm.value, bn.value, mode.value))
        if want_stats:
            self._colsum_need = max(self._colsum_need, col_off)
        return out, (dict(parts=parts) if want_stats else None)

    # ---- GroupNorm ----------------------------------------------------------------------------------------
    def _gn_slot(self, n, groups) -> int:
        off = self.gn_slots
        self.gn_slots += n * groups * 2
        return off

    def gn_finalize(self, x: Act, groups: int, stats: dict) -> int:
        """colsum slab(s) (already written by the producer) -> fp64 sums slot; returns the slot offset.  A tensor produced
        by several launches (`stats["parts"]`: the interior / boundary split of a depth-sharded conv) is finalised slab by
        slab, the later ones accumulating (ordered on the stream: still deterministic)."""
        slot = self._gn_slot(x.n, groups)
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        n, c = x.n, x.c
        parts = stats.get("parts") or [dict(tps=stats["tps"], cpad=stats["cpad"], nclass=stats["nclass"], off=0)]

        def run():
            for i, pt in enumerate(parts):
                lib.gn_finalize(C.c_void_p(prog._colsum.data_ptr() + pt["off"] * 4),
                                C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), n, c, pt["cpad"], groups, pt["tps"],
                                pt["nclass"], int(i > 0), sptr)

        self._emit(run, "gn.finalize")
        # depth-sharded programs: the slot holds this slab's sums until `sync_stats_and_halos` (emitted by the consumer:
        # gn_apply / attention / the fused residual tail) all-reduces it together with the tensor's boundary slices
        return slot

    def gn_colsum(self, x: Act) -> dict:
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        tps = lib.gn_colsum_tiles(x.d, x.h, x.w)
        self._colsum_need = max(self._colsum_need, 2 * x.n * tps * x.c)
        xp = x.ip
        n, c, d, h, w = x.n, x.c, x.d, x.h, x.w

        def run():
            lib.gn_colsum(xp, _ptr(prog._colsum), n, c, d, h, w, None, sptr)

        self._emit(run, "gn.colsum", nbytes=2.0 * n * c * d * h * w)
        return dict(tps=tps, cpad=x.c, nclass=1)

    def gn_apply(self, x: Act, slot: int, gn: nn.GroupNorm, *, silu_pre: bool, tbias=None,
                 tbias_off: int = 0, tbias_stride: int = 0, step_ptr: Optional[torch.Tensor] = None,
                 residual: Optional[Act] = None, silu_post: bool = False, out: Optional[Act] = None,
                 synced: bool = False) -> Act:
        """y = [silu](gn(x)) [+ tbias] [+ residual] [silu].  Depth-sharded: first the sync point that all-reduces the
        statistics and exchanges x's RAW boundary slices (unless `synced`: the caller already did), then the
        normalisation over own + received halo slices -- y's halos are valid without an exchange of its own."""
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        if self.shard is not None and not synced:
            self.sync_stats_and_halos(x if x.halo else None, slot, x.n * gn.num_groups * 2)
        if residual is not None and self.shard is not None and residual.halo:
            self.halo_exchange(residual)     # (already valid in the networks of this repo: a no-op)
        lo, d_ext = self.ext(x)
        gamma = self.dev_f32(lambda: gn.weight)
        beta = self.dev_f32(lambda: gn.bias)
        self.track(gn.weight, gn.bias)
        fresh = out is None
        if out is None:
            out = self.act(x.n, x.c, x.d, x.h, x.w, halo=x.halo)
        out.dirty = bool(self.shard is not None and x.halo and x.dirty)
        xp, yp, gp, bp = self.ext_ptr(x, lo), self.ext_ptr(out, lo), _ptr(gamma), _ptr(beta)
        tbp = C.c_void_p(0 if tbias is None else tbias.data_ptr() + tbias_off * 4)
        stp = _ptr(step_ptr)
        rp = C.c_void_p(0) if residual is None else self.ext_ptr(residual, lo)
        n, c, h, w, groups, eps = x.n, x.c, x.h, x.w, gn.num_groups, float(gn.eps)
        d_stat = self.shard.depth_total if (self.shard is not None and x.halo) else x.d   # statistics span all ranks
        d = d_ext

        def run():
            lib.gn_apply(xp, yp, C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), gp, bp, n, c, d, h, w, d_stat,
                         groups, eps, int(silu_pre), tbp, tbias_stride, stp, rp, int(silu_post), sptr)

        self._emit(run, "gn.apply", nbytes=(2 + (residual is not None)) * 2.0 * n * c * d * h * w)
        if fresh:
            self.zero_end_halos(out)
        return out

    # ---- U-Net blocks ------------------------------------------------------------------------------------
    def unet_resblock(self, m, x: Act, skip: Optional[Act], tbias: torch.Tensor, tbias_off: int,
                      tbias_stride: int, step_ptr: Optional[torch.Tensor]) -> Act:
        """ResBlock3D of the U-Net (models/unet3d.py:116-133); `skip` is the second half of a
        channel concatenation feeding the block (never materialised)."""
        cout = m.conv1.conv.out_channels
        has_res_conv = not isinstance(m.residual_conv, nn.Identity)
        if not has_res_conv and skip is not None:
            raise CtsiError("identity residual with a concatenated input")
        c1, st = self.conv("rb.conv1", lambda: m.conv1.conv.weight, lambda: m.conv1.conv.bias, x, skip,
                           cout=cout, want_stats=True)
        slot = self.gn_finalize(c1, m.conv1.norm.num_groups, st)
        # conv2 consumes silu(gn(c1)) + time bias: normalised on load where the plan supports it (c1 is never rewritten)
        c2, st = self.conv("rb.conv2", lambda: m.conv2[0].weight, lambda: m.conv2[0].bias, c1, None, cout=cout,
                           want_stats=True,
                           norm_in=(slot, m.conv1.norm, True, (tbias, tbias_off, tbias_stride, step_ptr)))
        self.release(c1)
        slot = self.gn_finalize(c2, m.conv2[1].num_groups, st)
        if not has_res_conv:
            return self.gn_apply(c2, slot, m.conv2[1], silu_pre=False, residual=x, silu_post=True, out=c2)
        if os.environ.get("CTSI_NO_FUSE_RES"):   # tuning / test aid: residual conv and GroupNorm tail as two launches
            if self.shard is not None:
                for a in (x, skip):
                    if a is not None:
                        self.halo_exchange(a)
            r, _ = self.conv("res1x1", lambda: m.residual_conv.weight, lambda: m.residual_conv.bias, x, skip,
                             k=(1, 1, 1), p=(0, 0, 0), cout=cout, ext_out=self.shard is not None)
            out = self.gn_apply(c2, slot, m.conv2[1], silu_pre=False, residual=r, silu_post=True, out=c2)
            self.release(r)
            return out
        # block tail with a 1x1x1 residual conv: out = silu(gn(c2) + W_r [x | skip] + b_r) in ONE launch -- the conv's
        # epilogue applies the GroupNorm of c2 and the activation, so the residual tensor never goes to HBM
        if self.shard is not None:
            self.sync_stats_and_halos(c2, slot, c2.n * m.conv2[1].num_groups * 2)
            for a in (x, skip):
                if a is not None:
                    self.halo_exchange(a)    # valid already (a no-op) for every block of the U-Net
        out, _ = self.conv("res1x1+gn", lambda: m.residual_conv.weight, lambda: m.residual_conv.bias, x, skip,
                           k=(1, 1, 1), p=(0, 0, 0), cout=cout, out=c2, fuse_gn=(c2, slot, m.conv2[1], True),
                           ext_out=self.shard is not None)
        return out

    def attention(self, m, x: Act, mode: str = "fast") -> Act:
        """TemporalAttention (models/unet3d.py:163-194), see csrc/attention.hip."""
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        n, c, d, h, w = x.n, x.c, x.d, x.h, x.w
        dev = self.ctx.device
        heads = m.num_heads
        tps = lib.attn_depthsum_tiles(c, h, w)
        self._colsum_need = max(self._colsum_need, 2 * n * tps * c)
        depthsum = self.pool.get(n * h * w * c, torch.float32)
        xp, dsp = x.ip, _ptr(depthsum)
        world = self.shard.world if (self.shard is not None and x.halo) else 1
        d_all = self.shard.depth_total if (self.shard is not None and x.halo) else d

        def run_ds():
            lib.attn_depthsum(xp, dsp, _ptr(prog._colsum), n, c, d, h, w, sptr)

        self._emit(run_ds, "attn.depthsum", nbytes=2.0 * n * c * d * h * w)
        slot = self.gn_finalize(x, m.norm.num_groups, dict(tps=tps, cpad=c, nclass=1))
        if self.shard is not None and x.halo:
            # one sync point: GroupNorm statistics (fp64) + the depth sum (fp32) over all ranks
            self.sync_stats_and_halos(None, slot, n * m.norm.num_groups * 2, f32=depthsum, name="attn.sync")
            self.halo_exchange(x)        # the block's input (valid already in this repo's networks: a no-op)
        gamma = self.dev_f32(lambda: m.norm.weight)
        beta = self.dev_f32(lambda: m.norm.bias)
        groups, eps = m.norm.num_groups, float(m.norm.eps)
        gp, bp = _ptr(gamma), _ptr(beta)

        # fold proj_out . V-projection:  P = (Wp Wv) xs + (D Wp bv + bp)
        def wv():
            return m.qkv.weight[2 * c:3 * c, :, 0, 0, 0].double()

        def wpv():
            wp = m.proj_out.weight[:, :, 0, 0, 0].double()
            return (wp @ wv()).float().reshape(c, c, 1, 1, 1)

        def bpv():
            wp = m.proj_out.weight[:, :, 0, 0, 0].double()
            return (float(d_all) * (wp @ m.qkv.bias[2 * c:3 * c].double()) + m.proj_out.bias.double()).float()

        if lib.attn_pv_supported(c, groups) and not os.environ.get("CTSI_NO_ATTN_PV"):
            # one launch: normalise the depth sum in registers and multiply by the folded matrix (csrc/attention.hip)
            wbuf = self.persistent((c, c), torch.bfloat16)
            self.pack_fns.append(lambda: wbuf.copy_(wpv().reshape(c, c).to(device=dev, dtype=torch.bfloat16)))
            bias_pv = self.dev_f32(bpv)
            pterm = self.act(n, c, 1, h, w, halo=0)
            wp_, bpp, ptp = _ptr(wbuf), _ptr(bias_pv), _ptr(pterm.t)

            def run_pv():
                lib.attn_pv(dsp, C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), gp, bp, wp_, bpp, ptp, n, c, d_all, h, w,
                            groups, eps, sptr)

            fl_pv = 2.0 * n * h * w * c * c
            self.flops += fl_pv
            self._emit(run_pv, "attn.pv", fl_pv, "attn_pv_mfma")
            self.pool.put(depthsum)
        else:
            xs = self.act(n, c, 1, h, w, halo=0)
            xsp = _ptr(xs.t)

            def run_ns():
                lib.attn_normsum(dsp, C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), gp, bp, xsp, n, c, d_all, h, w,
                                 groups, eps, sptr)

            self._emit(run_ns, "attn.normsum")
            pterm, _ = self.conv("attn.pv", wpv, bpv, xs, None, k=(1, 1, 1), p=(0, 0, 0), cout=c)
            self.pool.put(depthsum)
            self.release(xs)
        rowsum = None
        if mode == "exact":
            # evaluate the softmax row sums the reference multiplies in (they equal 1 up to rounding)
            xn = self.gn_apply(x, slot, m.norm, silu_pre=False)
            qk, _ = self.conv("attn.qk", lambda: m.qkv.weight[:2 * c], lambda: m.qkv.bias[:2 * c], xn, None,
                              k=(1, 1, 1), p=(0, 0, 0), cout=2 * c)
            self.release(xn)
            if world > 1:
                raise CtsiError("exact-mode attention needs every key on one rank; use the fast mode when sharding")
            rowsum = self.pool.get(n * d * h * w * heads, torch.float32)
            qkp, rsp = qk.ip, _ptr(rowsum)

            def run_rs():
                lib.attn_softmax_rowsum(qkp, rsp, n, c, d, h, w, heads, sptr)

            self._emit(run_rs, "attn.softmax_rowsum")
            self.release(qk)
        out = self.act(n, c, d, h, w, halo=x.halo)
        lo, d_ext = self.ext(x)          # x + P over own AND halo slices: P does not depend on depth
        out.dirty = False
        pp, op_, rsp2 = pterm.ip, self.ext_ptr(out, lo), _ptr(rowsum)
        xep = self.ext_ptr(x, lo)

        def run_ba():
            lib.attn_broadcast_add(xep, pp, rsp2, heads, op_, n, c, d_ext, h, w, sptr)

        self._emit(run_ba, "attn.broadcast_add", nbytes=4.0 * n * c * d_ext * h * w)
        self.zero_end_halos(out)
        self.release(pterm)
        if rowsum is not None:
            self.pool.put(rowsum)
        return out

    # ---- finish layout / execution ----------------------------------------------------------------------------
    def finalize_layout(self):
        dev = self.ctx.device
        self._colsum = torch.empty(max(self._colsum_need, 1), dtype=torch.float32, device=dev)
        self._gn_sums = torch.zeros(max(self.gn_slots, 1), dtype=torch.float64, device=dev)
        self.repack()

    def zero_gn_op(self):
        """Kept for the program builders' call sites: ctsi_gn_finalize WRITES its (sum, sumsq) slots (one block per
        (sample, group), fixed-order reduce), so the statistics buffer needs no per-evaluation zeroing any more."""

    def run(self):
        for op in self.ops:
            op()

    def profile_ops(self, repeats: int = 1):
        """Eager run with a HIP event pair around every op (on the engine stream).  Returns a list of
        (name, kernel, flops, milliseconds) — used by bench.py for the live roofline figure."""
        lib, sptr = self.lib, self.ctx.sptr
        nops = len(self.ops)
        evs = []
        for _ in range(nops + 1):
            e = C.c_void_p()
            lib.event_create(C.byref(e))
            evs.append(e)
        acc = [0.0] * nops
        for _ in range(repeats):
            lib.event_record(evs[0], sptr)
            for i, op in enumerate(self.ops):
                op()
                lib.event_record(evs[i + 1], sptr)
            for i in range(nops):
                ms = C.c_float()
                lib.event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms))
                acc[i] += ms.value
        for e in evs:
            lib.event_destroy(e)
        return [(m[0], m[2], m[1], acc[i] / repeats) for i, m in enumerate(self.op_meta)]

    def capture(self):
        lib = self.lib
        lib.graph_begin_capture(self.ctx.sptr)
        try:
            self.run()
        finally:
            g = C.c_void_p()
            lib.graph_end_capture(self.ctx.sptr, C.byref(g))
        self.graph = g

    def launch(self):
        if self.graph is None:
            self.run()
        else:
            self.lib.graph_launch(self.graph, self.ctx.sptr)

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.graph_destroy(self.graph)
            for p in self.plans:
                self.lib.conv_plan_destroy(p)
            for ev in getattr(self, "_events", ()):
                self.lib.event_destroy(ev)
            if self._fast and self._fast.get("graph") is not None:
                self.lib.graph_destroy(self._fast["graph"])
        except Exception:
            pass


# ==========================================================================================================
# U-Net
# ==========================================================================================================
class UNetProgram(Program):
    """One U-Net evaluation (models/unet3d.py:357-413) at a fixed latent shape, followed optionally by
    a sampler update; `step_ptr` selects the timestep row, so one captured graph serves all steps."""

    def __init__(self, ctx: Ctx, unet, n: int, d: int, h: int, w: int, max_rows: int, attention_mode="fast",
                 shard=None):
        """`d` is the depth this program owns: the whole volume on one GPU, or the rank's slab of
        `shard.depth_total // shard.world` slices when `shard` (parallel.ShardSpec) is given."""
        super().__init__(ctx)
        self.shard = shard
        if shard is not None and n != 1:
            raise CtsiError("depth-sharded programs support one volume per rank")
        self.unet = unet
        self.n, self.d, self.h, self.w = n, d, h, w
        L = unet.latent_dim
        self.L = L
        dev = ctx.device
        self.attention_mode = attention_mode
        self.max_rows = max_rows
        halo = 0 if shard is None else 1
        self.xin = Act(self.persistent((n * (d + 2 * halo) * h * w * 2 * L,), torch.bfloat16, zero=True), n, 2 * L, d,
                       h, w, halo)
        self.eps = self.persistent((n, d, h, w, L), torch.float32)
        self.z = self.persistent((n, d, h, w, L), torch.float32, zero=True)
        self.step_ptr = self.persistent((1,), torch.int32, zero=True)
        self.t_rows = self.persistent((max_rows,), torch.int32, zero=True)
        self.coef = self.persistent((max_rows, 8), torch.float32, zero=True)
        self.noise = None  # fp32 NCDHW, allocated on demand
        self.track_module(unet)

        # ---- time embedding: stack every ResBlock's Linear(time_dim -> cout) -------------------------
        te = unet.time_embed.time_mlp
        self.dim = unet.model_channels
        self.time_dim = te[1].out_features
        blocks = [m for m in unet.modules() if type(m).__name__ == "ResBlock3D"]
        self.tb_off = {}
        off = 0
        for m in blocks:
            self.tb_off[id(m)] = off
            off += m.time_mlp[1].out_features
        self.total_out = off
        self.w1 = self.dev_f32(lambda: te[1].weight)
        self.b1 = self.dev_f32(lambda: te[1].bias)
        self.w2 = self.dev_f32(lambda: te[3].weight)
        self.b2 = self.dev_f32(lambda: te[3].bias)
        self.w_all = self.dev_f32(lambda: torch.cat([m.time_mlp[1].weight for m in blocks], 0))
        self.b_all = self.dev_f32(lambda: torch.cat([m.time_mlp[1].bias for m in blocks], 0))
        self.tbias = self.persistent((max_rows, self.total_out), torch.float32, zero=True)
        self.te_scratch = self.persistent((max_rows * (self.dim + 2 * self.time_dim),), torch.float32)

        # ---- network ------------------------------------------------------------------------------------
        self.zero_gn_op()
        x, _ = self.conv("conv_in", lambda: unet.conv_in.weight, lambda: unet.conv_in.bias, self.xin, None,
                         cout=unet.conv_in.out_channels)
        skips: List[Act] = []

        def step(layer, xcur: Act, skip: Optional[Act] = None) -> Act:
            ynew = self._layer(layer, xcur, skip)
            if not any(xcur is sk for sk in skips):
                self.release(xcur)
            if skip is not None:
                self.release(skip)
            return ynew

        for level_blocks, down in zip(unet.down_blocks, unet.down_samples):
            for block_list in level_blocks:
                for layer in block_list:
                    x = step(layer, x)
            skips.append(x)  # stays allocated until the decoder has consumed it
            if not isinstance(down, nn.Identity):
                x, _ = self.conv("down", (lambda m=down: m.conv.weight), (lambda m=down: m.conv.bias), x, None,
                                 k=(3, 4, 4), s=(2, 2), p=(1, 1, 1), cout=down.conv.out_channels)
        x = step(unet.mid_block1, x)
        x = step(unet.mid_attn, x)
        x = step(unet.mid_block2, x)
        for level_blocks, up in zip(unet.up_blocks, unet.up_samples):
            for j, block_list in enumerate(level_blocks):
                skip = skips.pop() if j == 0 else None
                for layer in block_list:
                    x = step(layer, x, skip)
                    skip = None
            if not isinstance(up, nn.Identity):
                xo, _ = self.conv("up", (lambda m=up: m.conv.weight), (lambda m=up: m.conv.bias), x, None,
                                  transposed=True, k=(3, 4, 4), s=(2, 2), p=(1, 1, 1),
                                  cout=up.conv.out_channels)
                self.release(x)
                x = xo
        gn, conv = unet.conv_out[0], unet.conv_out[2]
        st = self.gn_colsum(x)
        slot = self.gn_finalize(x, gn.num_groups, st)
        y = self.gn_apply(x, slot, gn, silu_pre=True)
        self.release(x)
        vox = d * h * w
        self.conv("conv_out", lambda: conv.weight, lambda: conv.bias, y, None, cout=L, f32_out=self.eps,
                  f32_strides=(vox * L, 1, h * w * L, w * L, L))
        self.release(y)
        self.unet_op_count = len(self.ops)
        self.finalize_layout()

    def _layer(self, layer, x: Act, skip: Optional[Act]) -> Act:
        kind = type(layer).__name__
        if kind == "ResBlock3D":
            return self.unet_resblock(layer, x, skip, self.tbias, self.tb_off[id(layer)], self.total_out,
                                      self.step_ptr)
        if kind == "TemporalAttention":
            return self.attention(layer, x, self.attention_mode)
        raise CtsiError(f"unsupported U-Net layer {kind}")

    # -- inputs / schedule --
    def load_latents(self, z_ncdhw: Optional[torch.Tensor], cond_ncdhw: Optional[torch.Tensor]):
        """Take (this rank's depth slab of) the fp32 NCDHW latent / conditioning into the engine layout."""
        lib, sptr = self.lib, self.ctx.sptr
        n, L, d, h, w = self.n, self.L, self.d, self.h, self.w
        lo = 0 if self.shard is None else self.shard.depth_start
        if z_ncdhw is not None:
            z = z_ncdhw.detach()[:, :, lo:lo + d].to(device=self.ctx.device, dtype=torch.float32).contiguous()
            lib.ncdhw_f32_to_ndhwc_f32(_ptr(z), _ptr(self.z), n, L, d, h, w, sptr)
            lib.ncdhw_f32_to_ndhwc_bf16(_ptr(z), self.xin.ip, n, L, d, h, w, 2 * L, 0, sptr)
            z.record_stream(self.ctx.stream)
            self.xin.dirty = True
        if cond_ncdhw is not None:
            cnd = cond_ncdhw.detach()[:, :, lo:lo + d].to(device=self.ctx.device, dtype=torch.float32).contiguous()
            lib.ncdhw_f32_to_ndhwc_bf16(_ptr(cnd), self.xin.ip, n, L, d, h, w, 2 * L, L, sptr)
            cnd.record_stream(self.ctx.stream)

    def set_schedule(self, t_rows: Sequence[int], coef_rows: Optional[torch.Tensor] = None):
        """Upload the timestep of every (step, sample) row, embed all of them in one go and rewind the
        device-side step counter."""
        rows = len(t_rows)
        if rows > self.max_rows:
            raise CtsiError(f"schedule needs {rows} rows but the program was built for {self.max_rows}")
        lib, sptr = self.lib, self.ctx.sptr
        self.ensure_fresh()
        tt = torch.tensor(list(t_rows), dtype=torch.int32)
        self.t_rows[:rows].copy_(tt, non_blocking=False)
        if coef_rows is not None:
            self.coef[:coef_rows.shape[0]].copy_(coef_rows.to(torch.float32))
        self.step_ptr.zero_()
        lib.time_embed_fwd(_ptr(self.t_rows), rows, self.dim, self.time_dim, _ptr(self.w1), _ptr(self.b1),
                           _ptr(self.w2), _ptr(self.b2), _ptr(self.w_all), _ptr(self.b_all), self.total_out,
                           _ptr(self.te_scratch), _ptr(self.tbias), sptr)

    def eps_ncdhw(self) -> torch.Tensor:
        out = torch.empty((self.n, self.L, self.d, self.h, self.w), dtype=torch.float32, device=self.ctx.device)
        self.lib.ndhwc_f32_to_ncdhw_f32(_ptr(self.eps), _ptr(out), self.n, self.L, self.d, self.h, self.w,
                                        self.ctx.sptr)
        return out

    def z_ncdhw(self) -> torch.Tensor:
        """The sampler state as fp32 NCDHW (the rank's slab only when depth-sharded; gather with
        shard.comm.gather_depth)."""
        out = torch.empty((self.n, self.L, self.d, self.h, self.w), dtype=torch.float32, device=self.ctx.device)
        self.lib.ndhwc_f32_to_ncdhw_f32(_ptr(self.z), _ptr(out), self.n, self.L, self.d, self.h, self.w,
                                        self.ctx.sptr)
        return out

    def add_sampler_step(self, kind: str, with_noise: bool):
        """Append the DDIM/DDPM update and the step-counter increment (done once, before capture)."""
        lib, sptr = self.lib, self.ctx.sptr
        n, L, d, h, w = self.n, self.L, self.d, self.h, self.w
        if with_noise and self.noise is None:
            self.noise = self.persistent((n, L, d, h, w), torch.float32, zero=True)
        zp, ep, xp, cp, sp = _ptr(self.z), _ptr(self.eps), self.xin.ip, _ptr(self.coef), _ptr(self.step_ptr)
        npz = _ptr(self.noise if with_noise else None)
        # what the reference's NaN/Inf checkpoints would report: rows 0..max_rows-1 = per step {noise_pred, z_0_pred, z}
        # x {NaN, Inf}; the two extra rows = initial noise and conditioning (sampler.py:268-275).  Read once per sample().
        self.nonfinite = self.persistent((self.max_rows + 2, 6), torch.int32, zero=True)
        nfp = _ptr(self.nonfinite)

        xin = self.xin

        def run_step():
            if kind == "ddim":
                lib.ddim_step(zp, ep, npz, xp, 2 * L, 0, cp, sp, n, L, d, h, w, nfp, sptr)
            else:
                lib.ddpm_step(zp, ep, npz, xp, 2 * L, 0, cp, sp, n, L, d, h, w, sptr)
            xin.dirty = True

        def run_adv():
            lib.step_advance(sp, sptr)

        self._emit(run_step, "sampler.step", nbytes=(4 + 4 + 4 + 2 + (4 if with_noise else 0)) * float(n * L * d * h * w))
        self._emit(run_adv, "sampler.advance")
        self.sampler_kind = (kind, with_noise)


# ==========================================================================================================
# VAE
# ==========================================================================================================
def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


class VAEEncodeProgram(Program):
    """VideoEncoder.forward + scaling (models/vae.py:139-147, 235-247) at a fixed input shape."""

    def __init__(self, ctx: Ctx, vae, n, d, h, w):
        super().__init__(ctx)
        enc = vae.encoder
        self.n, self.d, self.h, self.w = n, d, h, w
        cin = vae.in_channels
        self.cin, self.cin_pad = cin, _pad8(cin)
        self.track_module(enc)
        self.xin = Act(self.persistent((n * d * h * w * self.cin_pad,), torch.bfloat16, zero=True), n,
                       self.cin_pad, d, h, w)
        self.zero_gn_op()
        x = self._conv_gn_act(enc.conv_in, self.xin, cin_w=cin, free=False)
        for stage in (enc.down1, enc.down2):
            for m in stage:
                x = self._block(m, x)
        for m in enc.mid:
            x = self._block(m, x)
        y, _ = self.conv("enc.conv_out", lambda: enc.conv_out.weight, lambda: enc.conv_out.bias, x, None,
                         cout=enc.conv_out.out_channels)
        self.release(x)
        L = vae.latent_dim
        self.L = L
        hl, wl = y.h, y.w
        self.hl, self.wl = hl, wl
        self.out = self.persistent((n, L, d, hl, wl), torch.float32)
        vox = d * hl * wl
        sf = lambda: float(vae.scaling_factor)
        self.conv("enc.quant", lambda: enc.quant_conv.weight * sf(), lambda: enc.quant_conv.bias * sf(), y, None,
                  k=(1, 1, 1), p=(0, 0, 0), cout=L, f32_out=self.out,
                  f32_strides=(L * vox, vox, hl * wl, wl, 1))
        self.release(y)
        self.finalize_layout()

    def _conv_gn_act(self, m, x: Act, *, cin_w=None, free=True, k=(3, 3, 3), s=(1, 1), transposed=False) -> Act:
        c, st = self.conv("conv+gn", lambda: m.conv.weight, lambda: m.conv.bias, x, None, k=k, s=s,
                          transposed=transposed, cout=m.conv.out_channels, cin_w=cin_w, want_stats=True)
        if free:
            self.release(x)
        slot = self.gn_finalize(c, m.norm.num_groups, st)
        return self.gn_apply(c, slot, m.norm, silu_pre=True, out=c)

    def _resblock(self, m, x: Act) -> Act:
        c1, st = self.conv("rb.conv1", lambda: m.conv1.conv.weight, lambda: m.conv1.conv.bias, x, None,
                           cout=x.c, want_stats=True)
        slot = self.gn_finalize(c1, m.conv1.norm.num_groups, st)
        c2, st = self.conv("rb.conv2", lambda: m.conv2[0].weight, lambda: m.conv2[0].bias, c1, None, cout=x.c,
                           want_stats=True, norm_in=(slot, m.conv1.norm, True, None))
        self.release(c1)
        slot = self.gn_finalize(c2, m.conv2[1].num_groups, st)
        out = self.gn_apply(c2, slot, m.conv2[1], silu_pre=False, residual=x, silu_post=True, out=c2)
        self.release(x)
        return out

    def _block(self, m, x: Act) -> Act:
        kind = type(m).__name__
        if kind == "ResBlock3D":
            return self._resblock(m, x)
        if kind == "DownsampleBlock":
            return self._conv_gn_act(m, x, k=(3, 4, 4), s=(2, 2))
        if kind == "UpsampleBlock":
            return self._conv_gn_act(m, x, k=(3, 4, 4), s=(2, 2), transposed=True)
        raise CtsiError(f"unsupported VAE block {kind}")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        lib, sptr = self.lib, self.ctx.sptr
        self.ensure_fresh()
        xx = x.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        lib.ncdhw_f32_to_ndhwc_bf16(_ptr(xx), _ptr(self.xin.t), self.n, self.cin, self.d, self.h, self.w,
                                    self.cin_pad, 0, sptr)
        xx.record_stream(self.ctx.stream)
        self.launch()
        out = self.out.clone()
        self.check_errors()
        return out


class VAEDecodeProgram(VAEEncodeProgram):
    """VideoDecoder.forward + unscaling (models/vae.py:190-204, 249-260) at a fixed latent shape.
    With `shard` the program decodes this rank's depth slab (d = local depth); halos are exchanged before
    every depth-3 conv and the GroupNorm statistics are all-reduced, as in the U-Net."""

    def __init__(self, ctx: Ctx, vae, n, d, h, w, shard=None):
        Program.__init__(self, ctx)
        self.shard = shard
        dec = vae.decoder
        self.n, self.d, self.h, self.w = n, d, h, w
        L = vae.latent_dim
        self.L, self.L_pad = L, _pad8(L)
        self.track_module(dec)
        halo = 0 if shard is None else 1
        self.zin = Act(self.persistent((n * (d + 2 * halo) * h * w * self.L_pad,), torch.bfloat16, zero=True), n,
                       self.L_pad, d, h, w, halo)
        self.zero_gn_op()
        inv = lambda: 1.0 / float(vae.scaling_factor)
        x, _ = self.conv("dec.post_quant", lambda: dec.post_quant_conv.weight * inv(),
                         lambda: dec.post_quant_conv.bias, self.zin, None, k=(1, 1, 1), p=(0, 0, 0),
                         cout=dec.post_quant_conv.out_channels, cin_w=L)
        x = self._conv_gn_act(dec.conv_in, x)
        for m in dec.mid:
            x = self._block(m, x)
        x = self._block(dec.up2_upsample, x)
        for m in dec.up2_res:
            x = self._block(m, x)
        x = self._block(dec.up3_upsample, x)
        for m in dec.up3_res:
            x = self._block(m, x)
        co = dec.conv_out.out_channels
        self.co = co
        self.ho, self.wo = x.h, x.w
        self.out = self.persistent((n, co, d, x.h, x.w), torch.float32)
        vox = d * x.h * x.w
        self.conv("dec.conv_out", lambda: dec.conv_out.weight, lambda: dec.conv_out.bias, x, None, cout=co,
                  f32_out=self.out, f32_strides=(co * vox, vox, x.h * x.w, x.w, 1), act=1)
        self.release(x)
        self.finalize_layout()

    def load(self, z: torch.Tensor):
        lib, sptr = self.lib, self.ctx.sptr
        self.ensure_fresh()
        lo = 0 if self.shard is None else self.shard.depth_start
        zz = z.detach()[:, :, lo:lo + self.d].to(device=self.ctx.device, dtype=torch.float32).contiguous()
        lib.ncdhw_f32_to_ndhwc_bf16(_ptr(zz), self.zin.ip, self.n, self.L, self.d, self.h, self.w, self.L_pad, 0, sptr)
        zz.record_stream(self.ctx.stream)

    def __call__(self, z: torch.Tensor) -> torch.Tensor:
        self.load(z)
        if self.shard is None:
            self.launch()
            out = self.out.clone()
            self.check_errors()
            return out
        self.run()
        out = self.shard.comm.gather_depth(self.shard.rank, self.out, counts=self.shard.depth_counts)
        self.check_errors()
        return out


# ==========================================================================================================
# program caches hung off the modules
# ==========================================================================================================
PROGRAM_CACHE_SIZE = int(os.environ.get("CTSI_PROGRAM_CACHE", "4"))   # programs kept per module (they own their
                                                                        # activation buffers: ~5 GB for the 512^2 U-Net)


def invalidate_engine_cache(module: nn.Module):
    """Drop every cached program (packed weights, captured graphs, activation buffers) of `module` and its
    sub-modules; the next call rebuilds from the current parameters."""
    for m in module.modules():
        m.__dict__.pop("_ctsi_programs", None)


def cached_program(module: nn.Module, key, build: Callable[[], Program]) -> Program:
    cache = module.__dict__.setdefault("_ctsi_programs", {})
    prog = cache.get(key)
    if prog is not None and prog.needs_rebuild():
        cache.pop(key)
        prog = None
    if prog is not None:
        cache[key] = cache.pop(key)   # LRU: most recently used last
    if prog is None:
        if len(cache) >= max(1, PROGRAM_CACHE_SIZE):
            cache.pop(next(iter(cache)))
        prog = build()
        cache[key] = prog
    return prog


def check_device_errors(ctx: Ctx):
    """Raise CtsiError if a kernel recorded a device-side error since the last check (ctsi_device_error_status: a split-K
    conv block whose wait for its partner expired raises a sticky count instead of passing a wrong result on
    silently).  One 8-byte synchronous read: called where the host reads results anyway (once per sample())."""
    cnt, det = C.c_uint(0), C.c_uint(0)
    ctx.stream.synchronize()       # the read below is a blocking copy on the null stream, which does not order with ours
    ctx.lib.device_error_status(C.byref(cnt), C.byref(det), 1)
    if cnt.value:
        # the failed hand-off left its ticket / flag words set (on purpose: a late partner must not meet reset words): start
        # every split-K layer of every live program on this device from a clean workspace again
        with torch.cuda.stream(ctx.stream):
            for prog in list(_LIVE_PROGRAMS):
                if prog.ctx is ctx:
                    for ws in prog._sk_workspaces:
                        ws.zero_()
        ctx.stream.synchronize()
        raise CtsiError(f"{cnt.value} device-side error(s) recorded by the HIP engine (last: split-K hand-off of conv tile "
                        f"{det.value} timed out); outputs computed since the last check are invalid")


def trilinear_depth(ctx: Ctx, z: torch.Tensor, d_out: int) -> torch.Tensor:
    """F.interpolate(z, size=(d_out, h, w), mode='trilinear', align_corners=False) for fp32 NCDHW z."""
    n, c, d, h, w = z.shape
    zz = z.detach().to(device=ctx.device, dtype=torch.float32).contiguous()
    out = torch.empty((n, c, d_out, h, w), dtype=torch.float32, device=ctx.device)
    ctx.lib.trilinear_depth_fwd(_ptr(zz), None, n, c, d, d_out, h, w, c, 0, _ptr(out), ctx.sptr)
    zz.record_stream(ctx.stream)
    return out


def nan_to_num_(ctx: Ctx, t: torch.Tensor) -> torch.Tensor:
    ctx.lib.nan_to_num_f32(_ptr(t), t.numel(), ctx.sptr)
    return t
