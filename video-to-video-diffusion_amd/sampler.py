"""DDPM / DDIM samplers (mirror of reference inference/sampler.py) on the HIP engine.

One denoising step = one replay of a captured hipGraph holding every kernel of a U-Net evaluation,
the elementwise x_{t-1} update (ctsi_ddim_step / ctsi_ddpm_step) and the increment of the device-side
step counter.  Per-step scalars (timestep embedding rows, update coefficients) come from device
tables indexed by that counter, so the same graph serves all steps and the host never synchronises
inside the loop.  The reference's five isnan/isinf host checks per step are folded into the update
kernel as unconditional nan_to_num (identity on finite values).
"""
from __future__ import annotations

import logging
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import Ctx, UNetProgram, _ptr, check_device_errors, nan_to_num_, trilinear_depth
from .lib import CtsiError

logger = logging.getLogger(__name__)

try:  # progress bars are cosmetic
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


def _is_engine_unet(model) -> bool:
    return type(model).__name__ == "UNet3D" and hasattr(model, "program")


def _log_nonfinite(kind: str, table: torch.Tensor, steps: int, max_rows: int):
    """What the reference's five NaN/Inf checkpoints log (inference/sampler.py:268-275, 288-292, 307-311, 331-334), from
    the device-side counters the update kernel keeps: one host read after the loop instead of five syncs per step."""
    if kind != "ddim":
        return            # the reference's DDPM loop (models/diffusion.py:340-367) has no such checkpoints
    t = table.cpu()
    if not bool(t.any()):
        return
    init, cond = t[max_rows], t[max_rows + 1]
    if int(init[0]) or int(init[1]):
        logger.error(f"NaN/Inf in initial noise z! NaN: {int(init[0])}, Inf: {int(init[1])}")
    if int(cond[0]) or int(cond[1]):
        logger.error(f"NaN/Inf in conditioning! NaN: {int(cond[0])}, Inf: {int(cond[1])}")
    for i in range(steps):
        r = [int(v) for v in t[i]]
        if r[0] or r[1]:
            logger.error(f"[Step {i}/{steps}] NaN/Inf in noise_pred! NaN: {r[0]}, Inf: {r[1]}")
        if r[2] or r[3]:
            logger.error(f"[Step {i}/{steps}] NaN/Inf in z_0_pred! NaN: {r[2]}, Inf: {r[3]}")
        if r[4] or r[5]:
            logger.error(f"[Step {i}/{steps}] NaN/Inf in z after update! NaN: {r[4]}, Inf: {r[5]}")


def _run_sampler_generic(diffusion, model, shape, conditioning, ctx, z0, *, kind, t_desc, eta, noise_fn, progress,
                         trajectory):
    """Reverse loop over an ARBITRARY `model(z, t, c) -> eps` callable (the reference's samplers accept any,
    inference/sampler.py:211-219): the network evaluation is the caller's (any torch code on the ROCm device), the
    x_{t-1} update with its guards is the engine's ctsi_ddim_step / ctsi_ddpm_step.  Not captured: the callable is
    opaque.  The engine's own UNet3D takes the hipGraph path in run_sampler instead."""
    import ctypes as C
    lib, sptr = ctx.lib, ctx.sptr
    n, L, d, h, w = [int(v) for v in shape]
    steps = len(t_desc)
    with_noise = (kind == "ddpm") or eta > 0
    dev = ctx.device
    coef = (ddim_coef_rows(diffusion.alphas_cumprod, t_desc, eta) if kind == "ddim"
            else diffusion.ddpm_coef_rows(t_desc)).to(dev, torch.float32).contiguous()
    z_nd = torch.empty((n, d, h, w, L), dtype=torch.float32, device=dev)
    eps_nd = torch.empty_like(z_nd)
    step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    nonfinite = torch.zeros((steps + 2, 6), dtype=torch.int32, device=dev)
    cond = conditioning.to(dev)
    z = z0.to(dev, torch.float32).contiguous()
    with ctx.scope():
        lib.count_nonfinite_f32(_ptr(z), z.numel(), 1, C.c_void_p(nonfinite.data_ptr() + steps * 24), sptr)
        cf = cond.float().contiguous()
        lib.count_nonfinite_f32(_ptr(cf), cf.numel(), 0, C.c_void_p(nonfinite.data_ptr() + (steps + 1) * 24), sptr)
    it = range(steps)
    if progress and tqdm is not None:
        it = tqdm(it, desc=f"{kind.upper()} Sampling", total=steps)
    for i in it:
        t = torch.full((n,), int(t_desc[i]), device=dev, dtype=torch.long)
        eps = model(z, t, cond)
        if not (torch.is_tensor(eps) and tuple(eps.shape) == tuple(shape) and eps.is_cuda):
            raise CtsiError("the model callable must return a ROCm tensor of the latent's shape "
                            f"{tuple(shape)}, got {type(eps).__name__} {tuple(getattr(eps, 'shape', ()))}")
        eps = eps.detach().to(torch.float32).contiguous()
        noise = None
        if with_noise:
            noise = (noise_fn(i, tuple(shape)) if noise_fn is not None else torch.randn(tuple(shape), device=dev))
            noise = noise.to(dev, torch.float32).contiguous()
        with ctx.scope():
            lib.ncdhw_f32_to_ndhwc_f32(_ptr(z), _ptr(z_nd), n, L, d, h, w, sptr)
            lib.ncdhw_f32_to_ndhwc_f32(_ptr(eps), _ptr(eps_nd), n, L, d, h, w, sptr)
            if kind == "ddim":
                lib.ddim_step(_ptr(z_nd), _ptr(eps_nd), _ptr(noise), None, 0, 0, _ptr(coef), _ptr(step_ptr), n, L, d, h, w,
                              _ptr(nonfinite), sptr)
            else:
                lib.ddpm_step(_ptr(z_nd), _ptr(eps_nd), _ptr(noise), None, 0, 0, _ptr(coef), _ptr(step_ptr), n, L, d, h, w,
                              sptr)
            lib.step_advance(_ptr(step_ptr), sptr)
            z = torch.empty((n, L, d, h, w), dtype=torch.float32, device=dev)
            lib.ndhwc_f32_to_ncdhw_f32(_ptr(z_nd), _ptr(z), n, L, d, h, w, sptr)
        if trajectory is not None:
            trajectory.append(z.clone())
    _log_nonfinite(kind, nonfinite, steps, steps)
    return z


def ddim_coef_rows(alphas_cumprod: torch.Tensor, timesteps: Sequence[int], eta: float) -> torch.Tensor:
    """Coefficient rows for ctsi_ddim_step, computed with fp32 torch ops in the reference's order
    (sampler.py:295-325): [sqrt(1-a+1e-8), sqrt(a+1e-8)+1e-8, sqrt(a'+1e-8), sqrt(1-a'+1e-8), sigma]."""
    ac = alphas_cumprod.detach().float().cpu()
    n = len(timesteps)
    rows = torch.zeros(n, 8, dtype=torch.float32)
    for i, t_idx in enumerate(timesteps):
        a = ac[int(t_idx)]
        a_prev = ac[int(timesteps[i + 1])] if i < n - 1 else torch.tensor(1.0)
        rows[i, 0] = torch.sqrt(1 - a + 1e-8)
        rows[i, 1] = torch.sqrt(a + 1e-8) + 1e-8
        rows[i, 2] = torch.sqrt(a_prev + 1e-8)
        rows[i, 3] = torch.sqrt(1 - a_prev + 1e-8)
        if eta > 0:
            rows[i, 4] = eta * torch.sqrt((1 - a_prev + 1e-8) / (1 - a + 1e-8) * (1 - a / (a_prev + 1e-8)))
    return rows


def run_sampler_sharded(diffusion, unet, shape, conditioning, ctx, z0, *, kind, t_desc, eta, noise_fn, comm,
                        trajectory=None):
    """Depth-sharded reverse loop: this process owns depth slab `comm.rank` of the volume (parallel.RcclComm: RCCL
    issued by libctsi on the engine stream; parallel.DistComm under the gloo tests).  Every rank passes the full
    conditioning / initial noise and gets the full result back (all-gather along depth).  A batch runs volume by
    volume through the one-volume sharded program (every rank holds 1/world of ONE volume at a time).
    With a capture-safe transport and CTSI_SHARD_CAPTURE=1 the step -- kernels AND collectives -- is replayed as one
    hipGraph, like the single-GPU step."""
    import os
    from .engine import cached_program
    from .parallel import ShardSpec
    n, L, d, h, w = [int(v) for v in shape]
    spec = ShardSpec(comm.rank, comm.world, comm, d)
    dl = spec.depth_local
    with_noise = (kind == "ddpm") or eta > 0
    steps = len(t_desc)
    capture = bool(getattr(comm, "capturable", False)) and os.environ.get("CTSI_SHARD_CAPTURE") == "1"
    outs, trajs = [], [[] for _ in range(steps)]
    with ctx.scope():
        key = ("sampler-shard", ctx.device.index, 1, d, h, w, comm.rank, comm.world, kind, with_noise)

        def build():
            prog = UNetProgram(ctx, unet, 1, dl, h, w, diffusion.timesteps + 1, "fast", shard=spec)
            prog.add_sampler_step(kind, with_noise)
            return prog

        prog = cached_program(unet, key, build)
        coef = (ddim_coef_rows(diffusion.alphas_cumprod, t_desc, eta) if kind == "ddim"
                else diffusion.ddpm_coef_rows(t_desc))
        lo = spec.depth_start
        noises = {}
        for b in range(n):
            prog.load_latents(z0[b:b + 1], conditioning[b:b + 1])
            prog.set_schedule([int(t) for t in t_desc], coef.to(ctx.device))
            if capture and prog.graph is None:
                prog.capture()
                prog.step_ptr.zero_()
            for i in range(steps):
                if with_noise:
                    if i not in noises:   # one draw per step for the whole batch, as the unsharded loop makes it
                        noises[i] = (noise_fn(i, tuple(shape)) if noise_fn is not None
                                     else torch.randn(tuple(shape), device=ctx.device))
                    prog.noise.copy_(noises[i][b:b + 1, :, lo:lo + dl].to(ctx.device, torch.float32))
                prog.launch() if capture else prog.run()
                if trajectory is not None:
                    trajs[i].append(comm.gather_depth(comm.rank, prog.z_ncdhw(), counts=spec.depth_counts))
            outs.append(comm.gather_depth(comm.rank, prog.z_ncdhw(), counts=spec.depth_counts))
        if trajectory is not None:
            trajectory.extend(torch.cat(t, dim=0) for t in trajs)
        res = torch.cat(outs, dim=0)
        torch.cuda.current_stream(ctx.device).synchronize()
        check_device_errors(ctx)
        return res


def run_sampler(diffusion, model, shape, conditioning, device, *, kind: str, t_desc: Sequence[int],
                progress: bool, eta: float = 0.0, noise_fn=None, z_init: Optional[torch.Tensor] = None,
                trajectory: Optional[list] = None):
    """Shared reverse loop.  kind: 'ddim' | 'ddpm'; t_desc: descending timestep list."""
    if not _is_engine_unet(model) and not callable(model):
        raise CtsiError(f"the samplers need a model(z, t, c) callable; got {type(model).__name__}")
    unet = model
    device = torch.device(device)
    ctx = Ctx.get(device if device.type == "cuda" else conditioning.device)
    n, L, d, h, w = [int(v) for v in shape]
    steps = len(t_desc)
    with_noise = (kind == "ddpm") or eta > 0
    max_rows = (diffusion.timesteps + 1) * n
    # initial noise is drawn exactly where the reference draws it (on the caller's stream/generator)
    if z_init is not None:
        z0 = z_init
    elif noise_fn is not None:
        z0 = noise_fn(-1, tuple(shape)).to(ctx.device)
    else:
        z0 = torch.randn(tuple(shape), device=ctx.device)
    if not _is_engine_unet(model):
        return _run_sampler_generic(diffusion, model, shape, conditioning, ctx, z0, kind=kind, t_desc=t_desc, eta=eta,
                                    noise_fn=noise_fn, progress=progress, trajectory=trajectory)
    comm = getattr(unet, "depth_shard_comm", None)
    if comm is not None and comm.world > 1:
        return run_sampler_sharded(diffusion, unet, shape, conditioning, ctx, z0, kind=kind, t_desc=t_desc, eta=eta,
                                   noise_fn=noise_fn, comm=comm, trajectory=trajectory)
    with ctx.scope():
        key = ("sampler", ctx.device.index, n, d, h, w, max_rows, kind, with_noise, unet.attention_mode)
        from .engine import cached_program

        def build():
            prog = UNetProgram(ctx, unet, n, d, h, w, max_rows, unet.attention_mode)
            prog.add_sampler_step(kind, with_noise)
            return prog

        prog: UNetProgram = cached_program(unet, key, build)
        prog.load_latents(z0, conditioning)
        import ctypes as C
        prog.nonfinite.zero_()
        nf_tail = prog.nonfinite.data_ptr() + prog.max_rows * 24
        # sampler.py:268-275: checkpoint 1 sanitises the initial noise (identity on finite values), checkpoint 2 only
        # reports on the conditioning; both are counted on device and logged after the loop
        ctx.lib.count_nonfinite_f32(_ptr(prog.z), prog.z.numel(), 1, C.c_void_p(nf_tail), ctx.sptr)
        cnd = conditioning.detach().to(ctx.device, torch.float32).contiguous()
        ctx.lib.count_nonfinite_f32(_ptr(cnd), cnd.numel(), 0, C.c_void_p(nf_tail + 24), ctx.sptr)
        cnd.record_stream(ctx.stream)
        if kind == "ddim":
            coef = ddim_coef_rows(diffusion.alphas_cumprod, t_desc, eta)
        else:
            coef = diffusion.ddpm_coef_rows(t_desc)
        t_rows = [int(t) for t in t_desc for _ in range(n)]
        prog.set_schedule(t_rows, coef.to(ctx.device))
        if prog.graph is None:
            # one eager warm-up step is not needed: capture records launches without executing them
            prog.capture()
            prog.step_ptr.zero_()
        it = range(steps)
        if progress and tqdm is not None:
            it = tqdm(it, desc=f"{kind.upper()} Sampling", total=steps)
        for i in it:
            if with_noise:
                if noise_fn is not None:
                    prog.noise.copy_(noise_fn(i, tuple(shape)).to(ctx.device, torch.float32))
                else:
                    prog.noise.normal_()
            prog.launch()
            if trajectory is not None:
                trajectory.append(prog.z_ncdhw())
        out = prog.z_ncdhw()
        _log_nonfinite(kind, prog.nonfinite, steps, prog.max_rows)     # (one host read: the loop itself never synchronises)
        check_device_errors(ctx)
        return out


class DDPMSampler:
    """Ancestral sampling over all `diffusion.timesteps` steps (reference sampler.py:17-61)."""

    def __init__(self, diffusion, model):
        self.diffusion = diffusion
        self.model = model
        self.timesteps = diffusion.timesteps

    @torch.no_grad()
    def sample(self, shape, conditioning, device, progress=True, noise_fn=None, num_steps=None,
               trajectory=None):
        t_desc = list(reversed(range(self.timesteps)))[:num_steps]
        return run_sampler(self.diffusion, self.model, shape, conditioning, device, kind="ddpm", t_desc=t_desc,
                           progress=progress, noise_fn=noise_fn, trajectory=trajectory)

    @torch.no_grad()
    def sample_with_stitching(self, v_thick_full, vae, patch_size=(8, 192, 192),
                              target_patch_size=(48, 192, 192), stride=(4, 96, 96), device='cuda',
                              progress=True, dp_group=None):
        return _stitched(self, v_thick_full, vae, patch_size, target_patch_size, stride, device, progress,
                         lambda shp, cond: self.sample(shp, cond, device, progress=False), dp_group=dp_group)

    def _create_gaussian_weight(self, d, h, w):
        return gaussian_weight(d, h, w)


class DDIMSampler:
    """Deterministic (eta = 0) or stochastic DDIM over a strided timestep subset (sampler.py:201-336)."""

    def __init__(self, diffusion, model):
        self.diffusion = diffusion
        self.model = model
        self.timesteps = diffusion.timesteps

    def _get_timesteps(self, num_inference_steps):
        """arange(0, T, T // N) plus T-1 when the stride misses it, descending — N+1 entries whenever
        T % N == 0 and N < T (sampler.py:221-239)."""
        stride = self.timesteps // num_inference_steps
        ts = np.arange(0, self.timesteps, stride)
        if ts[-1] != self.timesteps - 1:
            ts = np.append(ts, self.timesteps - 1)
        return ts[::-1]

    @torch.no_grad()
    def sample(self, shape, conditioning, num_inference_steps, device, eta=0.0, progress=True, noise_fn=None,
               trajectory=None, z_init=None):
        t_desc = [int(t) for t in self._get_timesteps(num_inference_steps)]
        return run_sampler(self.diffusion, self.model, shape, conditioning, device, kind="ddim", t_desc=t_desc,
                           progress=progress, eta=float(eta), noise_fn=noise_fn, trajectory=trajectory, z_init=z_init)

    @torch.no_grad()
    def sample_with_stitching(self, v_thick_full, vae, num_inference_steps=20, patch_size=(8, 192, 192),
                              target_patch_size=(48, 192, 192), stride=(4, 96, 96), device='cuda', eta=0.0,
                              progress=True, window_batch=None, dp_group=None):
        """`dp_group` (additive kwarg, default None = every window on this process, as in the reference): a
        torch.distributed process group (or True for the default group) over which the windows are split; every rank
        of the group must make the call with the SAME volume.
        `window_batch` (additive kwarg): windows are independent, so for the deterministic sampler (eta == 0) up to
        that many are encoded / sampled / decoded as one batch -- a single 192x192 patch leaves most of an MI355X
        idle (its coarsest level has 28 conv tiles for 256 CUs).  None / 0 (default): as many as a fifth of the device
        memory holds (13 windows of 48 x 192 x 192 on a 288 GB part); 1: one by one, like the reference.  The initial noise of every window is still drawn
        with its own `torch.randn` call in window order, exactly as the reference's one-by-one loop draws it."""
        batched = None
        if window_batch is None:
            window_batch = 0          # 0 = as many windows per batch as the device memory holds (see _stitched)
        if float(eta) == 0.0 and window_batch != 1:
            batched = lambda shp, cond, z_init: self.sample(shp, cond, num_inference_steps, device, eta=0.0,
                                                            progress=False, z_init=z_init)
        return _stitched(self, v_thick_full, vae, patch_size, target_patch_size, stride, device, progress,
                         lambda shp, cond: self.sample(shp, cond, num_inference_steps, device, eta=eta,
                                                       progress=False),
                         batched_fn=batched, window_batch=window_batch, dp_group=dp_group)

    def _create_gaussian_weight(self, d, h, w):
        return gaussian_weight(d, h, w)


def gaussian_weight(d: int, h: int, w: int) -> torch.Tensor:
    """Separable blend window, sigma = size/6, centred at (n-1)/2 (sampler.py:174-198)."""
    return _axis_window(d)[:, None, None] * _axis_window(h)[None, :, None] * _axis_window(w)[None, None, :]


def _window_starts(full: int, size: int, step: int):
    return sorted(set(list(range(0, full - size + 1, step)) + [max(0, full - size)]))


def _axis_window(n: int) -> torch.Tensor:
    x = torch.arange(n).float() - (n - 1) / 2
    return torch.exp(-(x ** 2) / (2 * (n / 6) ** 2))


def _window_partition(windows, dp_group, sampler):
    """Which windows this process computes: all of them unless `dp_group` opts in to window data parallelism.
    Returns (my_windows, world, group)."""
    rank, world, group = 0, 1, None
    if dp_group is not None and dp_group is not False:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise CtsiError("sample_with_stitching(dp_group=...) needs an initialised torch.distributed process group")
        comm = getattr(getattr(sampler, "model", None), "depth_shard_comm", None)
        if comm is not None and getattr(comm, "world", 1) > 1:
            raise CtsiError("window data parallelism (dp_group=) cannot be combined with depth sharding "
                            "(unet.depth_shard_comm): depth-sharded ranks must all run the same window")
        group = None if dp_group is True else dp_group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    from .parallel import shard_units
    return [windows[i] for i in shard_units(len(windows), rank, world)], world, group


def _stitched(sampler, v_thick_full, vae, patch_size, target_patch_size, stride, device, progress, sample_fn,
              batched_fn=None, window_batch=1, dp_group=None):
    """Sliding-window inference (sampler.py:63-172, 338-453): per window encode -> sample -> decode on the
    engine, Gaussian-weighted accumulation (ctsi_blend_accumulate) and final normalisation
    (ctsi_blend_normalize) on device.

    depth_ratio == 1 reproduces the reference exactly (pinned by tests/golden 'stitch.tiny.out').  For
    depth_ratio != 1 the reference samples the latent at the *thick* patch depth and then fails with a shape
    mismatch when it adds the decoded patch to the thin accumulator; here the window's
    conditioning latent is upsampled along depth to the target depth first, exactly as
    VideoToVideoDiffusion.generate does for a whole volume (models/model.py:284-289), so every window
    produces a (target_d, h, w) patch that lands at depth int(d_start * depth_ratio).
    Window data parallelism is OPT-IN (`dp_group=`): by default every window is computed by the calling process,
    whatever the state of torch.distributed -- the reference's behaviour, and the only safe one when ranks validate
    different volumes or only rank 0 calls.  With `dp_group` (a process group, or True for the default group) the
    windows are data-parallel units (parallel.shard_units): every rank of the group must call with the same volume,
    blends its share, and the accumulators are all-reduced over that group before normalisation.  It cannot be
    combined with depth sharding (`unet.depth_shard_comm`): there every rank must run the same window."""
    b, c, d_thick, hf, wf = v_thick_full.shape
    pd, ph, pw = patch_size
    td, th, tw = target_patch_size
    if (th, tw) != (ph, pw):
        raise CtsiError(f"sample_with_stitching: target patch (h, w)=({th},{tw}) must equal the thick patch "
                        f"(h, w)=({ph},{pw}); only depth is interpolated")
    ratio = td / pd
    d_thin = int(d_thick * ratio)
    dev = torch.device(device)
    ctx = Ctx.get(dev)
    lib, sptr = ctx.lib, ctx.sptr
    acc = torch.zeros(b, c, d_thin, hf, wf, device=ctx.device)
    wsum = torch.zeros(b, c, d_thin, hf, wf, device=ctx.device)
    wd, wh, ww = (_axis_window(n).to(ctx.device) for n in (td, th, tw))
    windows = [(ds, hs, ws) for ds in _window_starts(d_thick, pd, stride[0])
               for hs in _window_starts(hf, ph, stride[1]) for ws in _window_starts(wf, pw, stride[2])]
    mine, world, pg = _window_partition(windows, dp_group, sampler)
    it = None
    if progress and tqdm is not None:
        it = tqdm(desc="Patch-based inference", total=len(mine))
    group = 1
    if batched_fn is not None:
        group = int(window_batch)
        if group <= 0:
            # as many windows per batch as a fifth of the DEVICE memory holds: the VAE decoder is the largest program, ~2.5 KB of
            # activations per output voxel (30 GB for a 48 x 512 x 512 volume, 4.4 GB per 48 x 192 x 192 window: 13 windows on a
            # 288 GB part).  Measured on 25 windows: 13 per batch 1.69 s, all 25 at once 1.72 s, 5 per batch 1.98 s -- past a dozen
            # windows the levels are full and more only costs memory.  The TOTAL memory (not the free memory of the moment) keeps
            # the batch shape, and with it the cached programs, the same from call to call.
            try:
                total = torch.cuda.get_device_properties(ctx.device).total_memory
            except Exception:
                total = 64 << 30
            per_window = 2500.0 * b * td * th * tw
            group = max(1, min(len(mine), int(0.2 * total / per_window)))
    ngroups = max(1, -(-len(mine) // group))            # balanced groups: 25 windows, window_batch 8 -> 7 + 6 + 6 + 6
    bounds = [round(i * len(mine) / ngroups) for i in range(ngroups + 1)]
    for gi in range(ngroups):
        wins = mine[bounds[gi]:bounds[gi + 1]]
        if not wins:
            continue
        patch = torch.cat([v_thick_full[:, :, ds:ds + pd, hs:hs + ph, ws:ws + pw] for (ds, hs, ws) in wins], dim=0)
        z_cond = vae.encode(patch.to(ctx.device).contiguous())
        if td != pd:
            with ctx.scope():
                z_cond = trilinear_depth(ctx, z_cond, td)
        if len(wins) == 1 or batched_fn is None:
            z = sample_fn(tuple(z_cond.shape), z_cond)
        else:   # one torch.randn per window, in window order (the reference's RNG stream)
            one = (b,) + tuple(z_cond.shape[1:])
            z_init = torch.cat([torch.randn(one, device=ctx.device) for _ in wins], dim=0)
            z = batched_fn(tuple(z_cond.shape), z_cond, z_init)
        out = vae.decode(z).contiguous()
        with ctx.scope():
            for k, (ds, hs, ws) in enumerate(wins):
                ok = out[k * b:(k + 1) * b]
                lib.blend_accumulate(_ptr(acc), _ptr(wsum), _ptr(ok), _ptr(wd), _ptr(wh), _ptr(ww), b * c, td, th, tw,
                                     d_thin, hf, wf, int(ds * ratio), hs, ws, sptr)
        if progress and tqdm is not None:
            it.update(len(wins))
    if it is not None:
        it.close()
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(acc, group=pg)
        dist.all_reduce(wsum, group=pg)
    with ctx.scope():
        lib.blend_normalize(_ptr(acc), _ptr(wsum), acc.numel(), sptr)
    return acc


class EDMSampler:
    def __init__(self, diffusion, model):
        self.diffusion = diffusion
        self.model = model
        raise NotImplementedError("EDM sampler not yet implemented")
