"""Optimizer step of the training path on the HIP engine.

The reference builds `torch.optim.Adam` / `torch.optim.AdamW` over per-module parameter groups with their own learning
rates (training/train.py:172-212) and steps it once per accumulation window (training/trainer.py:237-247, through a
GradScaler under AMP).  `FusedAdamW` / `FusedAdam` are drop-ins for those two classes -- same constructor arguments, same
`param_groups` / `state` / `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so LR schedulers,
GradScaler, `clip_grad_norm_` and checkpoints of the torch optimizers keep working -- whose `step()` is ONE
`ctsi_adamw_multi` launch over all tensors of all groups, followed by the engine's fast re-pack of the bf16 kernel images of
the programs that use these parameters (`engine.Program.fast_repack`).  There is no CPU path: parameters must live on a
ROCm device."""
from __future__ import annotations

import ctypes as C
import math
import struct
from typing import Iterable, Optional, Sequence

import torch

from .lib import CtsiError, get_lib


class FusedAdamW(torch.optim.Optimizer):
    decoupled = True     # AdamW: p *= 1 - lr * wd; FusedAdam below: grad += wd * p (torch.optim.Adam's L2 form)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *,
                 maximize: bool = False, engine_modules: Optional[Sequence[torch.nn.Module]] = None):
        """`engine_modules` (additive kwarg): modules (e.g. `[model.unet]`) whose cached engine programs are re-packed right
        behind the update; without it the programs notice the new parameter versions on their next use and re-pack then
        (the generic, slower path)."""
        if amsgrad:
            raise CtsiError("FusedAdamW: amsgrad is not supported by the HIP engine")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError(f"invalid optimizer hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize)
        super().__init__(params, defaults)
        self.engine_modules = list(engine_modules) if engine_modules is not None else []
        self._lib = get_lib()
        self._tables = None

    # ---- state (torch layout: state[p] = {step, exp_avg, exp_avg_sq}) ---------------------------------------------------
    def _init_state(self, p: torch.Tensor):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def state_dict(self):
        """torch's layout.  The step counters the parameters of a group share in here are written out as one tensor per
        parameter: torch's optimizers increment `step` per parameter, so aliased counters (aliasing survives the deepcopy
        in load_state_dict) would be bumped once per parameter."""
        sd = super().state_dict()
        sd["state"] = {k: ({**v, "step": v["step"].clone()} if isinstance(v, dict) and torch.is_tensor(v.get("step")) else v)
                       for k, v in sd["state"].items()}
        return sd

    def _build_tables(self, entries, dev):
        """entries: [(param, hyper-row index)] of the parameters that have a gradient this step."""
        chunk = self._lib.adamw_chunk_elems()
        chunks = []
        for i, (p, _) in enumerate(entries):
            chunks.extend((i, q * (chunk // 4)) for q in range((p.numel() + chunk - 1) // chunk))
        ck = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        n = len(entries)
        host = torch.zeros((n, 6), dtype=torch.int64)               # CtsiOptTensor rows: p, g, m, v, numel, (row | pad)
        for i, (p, row) in enumerate(entries):
            st = self.state[p]
            for name in ("exp_avg", "exp_avg_sq"):
                if st[name].device != p.device or st[name].dtype != torch.float32 or not st[name].is_contiguous():
                    st[name] = st[name].to(device=p.device, dtype=torch.float32).contiguous()   # (a loaded state dict)
            host[i, 0], host[i, 2], host[i, 3] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            host[i, 4], host[i, 5] = p.numel(), row                 # (little endian: the int `group` is the low word)
        self._tables = dict(key=tuple(id(p) for p, _ in entries), rows=tuple(r for _, r in entries), chunks=ck,
                            nchunks=len(chunks), dev=dev, host=host, gptr=None,
                            state_ptrs=tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                                              self.state[p]["exp_avg_sq"].data_ptr()) for p, _ in entries),
                            tensors=torch.empty((n, 6), dtype=torch.int64, device=dev),
                            groups=torch.empty(max(n, 1) * 64, dtype=torch.uint8, device=dev))

    def _hyper_rows(self):
        """Step bookkeeping + one CtsiOptGroup row per (parameter group, step count) present.  torch counts steps per
        parameter (a parameter without a gradient skips the step), so a group may hold several counts; the counters of
        parameters that move together are ONE shared CPU tensor (350 tensor increments per step would cost more host time
        than the kernel takes); a parameter that sits a step out gets its own copy first.  Returns (entries, rows)."""
        entries, rows, row_of = [], [], {}
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            live = [p for p in params if p.grad is not None]
            if not live:
                continue
            state = self.state
            first = state.get(live[0])
            shared = first["step"] if first else None
            uniform = (shared is not None and len(live) == len(params)
                       and all((st := state.get(p)) is not None and len(st) and st["step"] is shared for p in live))
            if uniform:                                  # the steady state: one counter, one row for the whole group
                shared += 1
                steps = [(live, float(shared))]
            else:
                live_counters = {id(state[q]["step"]) for q in live if state.get(q)}
                for p in params:                         # (parameters sitting this step out leave the shared counters ...)
                    st = state.get(p)
                    if p.grad is None and st and id(st["step"]) in live_counters:
                        st["step"] = st["step"].clone()
                bumped, by_t = {}, {}
                for p in live:                           # (... then the counters move on)
                    if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                        raise CtsiError("FusedAdamW runs on the HIP engine: parameters must be contiguous fp32 tensors on a "
                                        "ROCm device")
                    st = self._init_state(p)
                    key = id(st["step"])
                    if key not in bumped:
                        t_prev = float(st["step"])
                        share = next((c for c, tp in bumped.values() if tp == t_prev), None)
                        if share is not None:            # same count as a counter already seen: share it
                            st["step"] = share
                        else:
                            st["step"] += 1
                            bumped[key] = (st["step"], t_prev)
                    by_t.setdefault(float(st["step"]), []).append(p)
                steps = [(ps, t) for t, ps in by_t.items()]
            b1, b2 = group["betas"]
            lr, wd = float(group["lr"]), float(group["weight_decay"])
            for ps, t in steps:
                bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
                row_of[(gi, t)] = len(rows)
                rows.append(struct.pack("<11f5i", lr, b1, b2, float(group["eps"]), wd, lr / bc1, math.sqrt(bc2),
                                        1.0 - lr * wd, 1.0 - b1, 1.0 - b2, 1.0, int(self.decoupled),
                                        int(bool(group.get("maximize", False))), 0, 0, 0))
                entries.extend((p, row_of[(gi, t)]) for p in ps)
        return entries, rows

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, rows = self._hyper_rows()
        if not entries:
            return loss
        tb = self._tables
        key = tuple(id(p) for p, _ in entries)
        if (tb is None or tb["key"] != key or tb["rows"] != tuple(r for _, r in entries)
                or tb["state_ptrs"] != tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                                              self.state[p]["exp_avg_sq"].data_ptr()) for p, _ in entries)):
            for p, _ in entries:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise CtsiError("FusedAdamW runs on the HIP engine: parameters must be contiguous fp32 tensors on a ROCm "
                                    "device")
            self._build_tables(entries, entries[0][0].device)
            tb = self._tables
        dev = tb["dev"]
        # gradients: new tensors after every backward, usually at the addresses of the previous step (caching allocator)
        grads = []
        for p, _ in entries:
            g = p.grad
            if g.dtype is not torch.float32 or g.is_sparse or g.device != p.device or not g.is_contiguous():
                if g.is_sparse:
                    raise CtsiError("FusedAdamW does not support sparse gradients")
                g = g.to(device=p.device, dtype=torch.float32).contiguous()
            grads.append(g)
        gptr = tuple(g.data_ptr() for g in grads)
        if gptr != tb["gptr"]:
            tb["host"][:, 1] = torch.tensor(gptr, dtype=torch.int64)
            tb["tensors"].copy_(tb["host"])
            tb["gptr"] = gptr
        gbytes = b"".join(rows)
        tb["groups"][:len(gbytes)].copy_(torch.frombuffer(bytearray(gbytes), dtype=torch.uint8))
        stream = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            self._lib.adamw_multi(C.c_void_p(tb["tensors"].data_ptr()), C.c_void_p(tb["groups"].data_ptr()),
                                  C.c_void_p(tb["chunks"].data_ptr()), tb["nchunks"], C.c_void_p(stream.cuda_stream))
        for g in grads:
            g.record_stream(stream)
        # the update went through raw pointers: tell torch (and, through it, every engine program's fingerprint)
        bump = getattr(torch.autograd.graph, "increment_version", None)
        if bump is not None:
            bump([p for p, _ in entries])
        else:  # pragma: no cover  (older torch)
            for p, _ in entries:
                p.add_(0)
        self._repack_engine_programs()
        return loss

    def _repack_engine_programs(self):
        """Fast re-pack of the programs (training programs: they own their weight images) of `engine_modules`."""
        for mod in self.engine_modules:
            subs = mod.__dict__.get("_ctsi_submodules")
            if subs is None:
                subs = mod.__dict__["_ctsi_submodules"] = list(mod.modules())
            for m in subs:
                progs = m.__dict__.get("_ctsi_programs")
                if not progs:
                    continue
                for prog in list(progs.values()):
                    if not prog.weight_cache and not prog.needs_rebuild():
                        prog.ctx.enter()
                        try:
                            prog.fast_repack()
                        finally:
                            prog.ctx.leave()


class FusedAdam(FusedAdamW):
    """torch.optim.Adam (weight decay as an L2 term on the gradient), the reference's default optimizer
    (training/train.py:205-206)."""
    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
