"""TEST INFRASTRUCTURE ONLY — not part of the product path.

CPU restatement (fp32, plain torch.nn.functional ops on state dicts) of the reference's sampling
path.  The convolution / GroupNorm / softmax arithmetic of the reference lives in PyTorch itself
(requirements.txt: torch>=2.0.0, Dockerfile pins pytorch 2.5.1; this image: torch 2.10.0), so the
restatement calls the same F.* primitives at the reference's call sites and re-states everything
the reference writes itself (block wiring, schedules, samplers, index formulas).

Pinned by: tests/golden/*.npz, produced by tests/golden/make_golden.py which imports the real
reference from /root/reference in the build container (tests/test_oracle_golden.py compares).
The reference ships no tests or golden vectors of its own for this path.

Functions take `sd`, a dict name -> tensor in the reference's state-dict layout, plus a key prefix.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------
def group_count(channels: int) -> int:
    """largest of 32..1 dividing channels  (ref models/unet3d.py:62-68, 108-114, 155-161, 334-340)"""
    for g in (32, 16, 8, 4, 2, 1):
        if channels % g == 0:
            return g
    return 1


def conv_block_groups(channels: int) -> int:
    """ref models/unet3d.py:58: 8 groups when divisible by 8, else group_count"""
    return min(8, channels) if channels % 8 == 0 else group_count(channels)


def conv3d(sd: SD, p: str, x, stride=1, padding=0):
    return F.conv3d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


CONVT_AS_CONV = False   # see conv_transpose_122


def conv_transpose_122(x, weight, bias):
    """nn.ConvTranspose3d(k=(3,4,4), stride=(1,2,2), padding=1) (ref models/unet3d.py:218-221, models/vae.py:86-90).
    Default: PyTorch's own F.conv_transpose3d.  With CONVT_AS_CONV the same sum is evaluated as a stride-1 Conv3d over
    the zero-inserted input with the flipped, (cin, cout)-transposed kernel and padding k-1-p = (1,2,2) -- the textbook
    identity, bit-for-bit the same products in a different summation order (tests/test_oracle_golden.py holds the two
    to 1e-6).  The GPU parity tests use it at the 512^2 sizes because MIOpen's fp32 backward-data path (which
    F.conv_transpose3d maps to) spends minutes searching for / running a solver there, while its forward path does not."""
    if not CONVT_AS_CONV:
        return F.conv_transpose3d(x, weight, bias, stride=(1, 2, 2), padding=(1, 1, 1))
    n, c, d, h, w = x.shape
    up = x.new_zeros((n, c, d, 2 * h - 1, 2 * w - 1))
    up[:, :, :, ::2, ::2] = x
    wf = weight.flip(2, 3, 4).transpose(0, 1).contiguous()      # (cout, cin, 3, 4, 4)
    # MIOpen has no implicit-GEMM instance for (3,4,4) kernels and falls back to im2col + GEMM: its workspace is
    # cin * 48 taps * output voxels * 4 B (576 GiB for the VAE's last upsample at 512^2) -> evaluate depth slabs
    per_slice = c * 48 * (2 * h) * (2 * w) * 4 * n
    cd = max(1, min(d, int((32 << 30) // max(per_slice, 1))))
    if cd >= d:
        return F.conv3d(up, wf, bias, padding=(1, 2, 2))
    up = F.pad(up, (0, 0, 0, 0, 1, 1))
    return torch.cat([F.conv3d(up[:, :, d0:d0 + min(cd, d - d0) + 2], wf, bias, padding=(0, 2, 2))
                      for d0 in range(0, d, cd)], dim=2)


def gn(sd: SD, p: str, x, groups: int):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def timestep_sincos(t: torch.Tensor, dim: int) -> torch.Tensor:
    """ref models/unet3d.py:25-32"""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=t.device) * -scale)
    args = t[:, None] * freqs[None, :]
    return torch.cat((args.sin(), args.cos()), dim=-1)


def time_embedding(sd: SD, p: str, t: torch.Tensor, dim: int) -> torch.Tensor:
    """ref models/unet3d.py:40-48"""
    e = timestep_sincos(t, dim)
    e = F.linear(e, sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    e = F.silu(e)
    return F.linear(e, sd[p + ".time_mlp.3.weight"], sd[p + ".time_mlp.3.bias"])


def unet_resblock(sd: SD, p: str, x, temb):
    """ref models/unet3d.py:116-133"""
    cout = sd[p + ".conv1.conv.weight"].shape[0]
    res = conv3d(sd, p + ".residual_conv", x) if (p + ".residual_conv.weight") in sd else x
    h = conv3d(sd, p + ".conv1.conv", x, padding=1)
    h = F.silu(gn(sd, p + ".conv1.norm", h, conv_block_groups(cout)))
    tb = F.linear(F.silu(temb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    h = h + tb[:, :, None, None, None]
    h = conv3d(sd, p + ".conv2.0", h, padding=1)
    h = gn(sd, p + ".conv2.1", h, group_count(cout))
    return F.silu(h + res)


def temporal_attention(sd: SD, p: str, x, heads: int):
    """ref models/unet3d.py:163-194 — including the second einsum exactly as written
    ('bhqk,bhvc->bhqc': k and v are summed independently)."""
    b, c, t, hh, ww = x.shape
    hd = c // heads
    xn = gn(sd, p + ".norm", x, group_count(c))
    qkv = conv3d(sd, p + ".qkv", xn)                       # (b, 3c, t, h, w)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]

    def tokens(u):  # 'b (head c) t h w -> (b h w) head t c'
        return u.reshape(b, heads, hd, t, hh, ww).permute(0, 4, 5, 1, 3, 2).reshape(b * hh * ww, heads, t, hd)

    q, k, v = tokens(q), tokens(k), tokens(v)
    attn = torch.einsum('bhqc,bhkc->bhqk', q, k) * (hd ** -0.5)
    attn = F.softmax(attn, dim=-1)
    out = torch.einsum('bhqk,bhvc->bhqc', attn, v)
    out = out.reshape(b, hh, ww, heads, t, hd).permute(0, 3, 5, 4, 1, 2).reshape(b, c, t, hh, ww)
    return conv3d(sd, p + ".proj_out", out) + x


def unet_forward(sd: SD, cfg: dict, x, t, c, prefix: str = "", taps: Optional[dict] = None):
    """ref models/unet3d.py:357-413.  cfg: model_channels, num_res_blocks, attention_levels,
    channel_mult, num_heads."""
    P = prefix
    mc, nrb = cfg["model_channels"], cfg["num_res_blocks"]
    mult, att, heads = list(cfg["channel_mult"]), list(cfg["attention_levels"]), cfg["num_heads"]
    levels = len(mult)
    temb = time_embedding(sd, P + "time_embed", t, mc)
    h = conv3d(sd, P + "conv_in", torch.cat([x, c], dim=1), padding=1)
    if taps is not None:
        taps["conv_in"] = h
    skips = []
    for lvl in range(levels):
        for b in range(nrb):
            h = unet_resblock(sd, f"{P}down_blocks.{lvl}.{b}.0", h, temb)
            if lvl in att:
                h = temporal_attention(sd, f"{P}down_blocks.{lvl}.{b}.1", h, heads)
        skips.append(h)
        if lvl < levels - 1:
            h = conv3d(sd, f"{P}down_samples.{lvl}.conv", h, stride=(1, 2, 2), padding=(1, 1, 1))
    h = unet_resblock(sd, P + "mid_block1", h, temb)
    h = temporal_attention(sd, P + "mid_attn", h, heads)
    h = unet_resblock(sd, P + "mid_block2", h, temb)
    if taps is not None:
        taps["mid"] = h
    for lvl in range(levels):
        src = levels - 1 - lvl
        for b in range(nrb + 1):
            if b == 0:
                h = torch.cat([h, skips.pop()], dim=1)
            h = unet_resblock(sd, f"{P}up_blocks.{lvl}.{b}.0", h, temb)
            if src in att:
                h = temporal_attention(sd, f"{P}up_blocks.{lvl}.{b}.1", h, heads)
        if lvl < levels - 1:
            h = conv_transpose_122(h, sd[f"{P}up_samples.{lvl}.conv.weight"], sd[f"{P}up_samples.{lvl}.conv.bias"])
    cfin = h.shape[1]
    h = F.silu(gn(sd, P + "conv_out.0", h, group_count(cfin)))
    return conv3d(sd, P + "conv_out.2", h, padding=1)


# ---- VAE (ref models/vae.py) ---------------------------------------------------------------------------
def vae_conv_block(sd: SD, p: str, x, stride=1, transposed=False):
    """ref models/vae.py:31-35, 72-76, 93-97 (GroupNorm always 8 groups)"""
    if transposed:
        h = conv_transpose_122(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    else:
        h = F.conv3d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=stride, padding=(1, 1, 1))
    return F.silu(gn(sd, p + ".norm", h, 8))


def vae_resblock(sd: SD, p: str, x):
    """ref models/vae.py:50-56"""
    h = vae_conv_block(sd, p + ".conv1", x)
    h = gn(sd, p + ".conv2.1", conv3d(sd, p + ".conv2.0", h, padding=1), 8)
    return F.silu(h + x)


def vae_encode(sd: SD, x, scaling_factor: float, prefix: str = ""):
    """ref models/vae.py:139-147, 244-247"""
    P = prefix + "encoder."
    h = vae_conv_block(sd, P + "conv_in", x)
    for stage in ("down1", "down2"):
        h = vae_resblock(sd, f"{P}{stage}.0", h)
        h = vae_resblock(sd, f"{P}{stage}.1", h)
        h = vae_conv_block(sd, f"{P}{stage}.2", h, stride=(1, 2, 2))
    h = vae_resblock(sd, P + "mid.0", h)
    h = vae_resblock(sd, P + "mid.1", h)
    h = conv3d(sd, P + "conv_out", h, padding=1)
    return conv3d(sd, P + "quant_conv", h) * scaling_factor


def vae_decode(sd: SD, z, scaling_factor: float, prefix: str = ""):
    """ref models/vae.py:190-204, 258-260"""
    P = prefix + "decoder."
    h = conv3d(sd, P + "post_quant_conv", z / scaling_factor)
    h = vae_conv_block(sd, P + "conv_in", h)
    h = vae_resblock(sd, P + "mid.0", h)
    h = vae_resblock(sd, P + "mid.1", h)
    h = vae_conv_block(sd, P + "up2_upsample", h, transposed=True)
    h = vae_resblock(sd, P + "up2_res.0", h)
    h = vae_resblock(sd, P + "up2_res.1", h)
    h = vae_conv_block(sd, P + "up3_upsample", h, transposed=True)
    h = vae_resblock(sd, P + "up3_res.0", h)
    h = vae_resblock(sd, P + "up3_res.1", h)
    return torch.tanh(conv3d(sd, P + "conv_out", h, padding=1))


# ---- schedules (ref models/diffusion.py:27-79) ---------------------------------------------------------
def diffusion_buffers(noise_schedule="cosine", timesteps=1000, beta_start=0.0001, beta_end=0.02) -> SD:
    if noise_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, timesteps)
    elif noise_schedule == "cosine":
        s = 0.008
        x = torch.linspace(0, timesteps, timesteps + 1)
        ac = torch.cos(((x / timesteps) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0.0001, 0.9999)
    else:
        raise ValueError(f"Unknown noise schedule: {noise_schedule}")
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas, "alphas": alphas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac), "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "posterior_variance": pv, "posterior_log_variance_clipped": torch.log(torch.clamp(pv, min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
    }


def ddim_timesteps(total: int, n_steps: int) -> np.ndarray:
    """ref inference/sampler.py:231-239"""
    step = total // n_steps
    ts = np.arange(0, total, step)
    if ts[-1] != total - 1:
        ts = np.append(ts, total - 1)
    return ts[::-1]


def _guard(v):
    return torch.nan_to_num(v, nan=0.0, posinf=1.0, neginf=-1.0)


def ddim_sample(model: Callable, buffers: SD, shape, cond, n_steps: int, eta: float = 0.0,
                noise_fn: Optional[Callable] = None, trajectory: Optional[list] = None):
    """ref inference/sampler.py:242-336.  model(z, t, cond) -> eps.  noise_fn(i, shape) replaces
    torch.randn (i = -1 for the initial draw)."""
    ac = buffers["alphas_cumprod"]
    ts = ddim_timesteps(ac.shape[0], n_steps)
    z = noise_fn(-1, tuple(shape)) if noise_fn else torch.randn(shape)
    z = _guard(z)
    b = shape[0]
    for i, t_idx in enumerate(ts):
        t = torch.full((b,), int(t_idx), dtype=torch.long, device=z.device)
        eps = _guard(model(z, t, cond))
        a = ac[int(t_idx)]
        a_prev = ac[int(ts[i + 1])] if i < len(ts) - 1 else torch.tensor(1.0)
        z0 = (z - torch.sqrt(1 - a + 1e-8) * eps) / (torch.sqrt(a + 1e-8) + 1e-8)
        z0 = torch.clamp(_guard(z0), -10.0, 10.0)
        dirz = torch.sqrt(1 - a_prev + 1e-8) * eps
        if eta > 0:
            sigma = eta * torch.sqrt((1 - a_prev + 1e-8) / (1 - a + 1e-8) * (1 - a / (a_prev + 1e-8)))
            noise = noise_fn(i, tuple(shape)) if noise_fn else torch.randn_like(z)
            z = torch.sqrt(a_prev + 1e-8) * z0 + dirz + sigma * noise
        else:
            z = torch.sqrt(a_prev + 1e-8) * z0 + dirz
        z = _guard(z)
        if trajectory is not None:
            trajectory.append(z.clone())
    return z


def ddpm_sample(model: Callable, buffers: SD, shape, cond, noise_fn: Optional[Callable] = None,
                num_steps: Optional[int] = None, trajectory: Optional[list] = None):
    """ref models/diffusion.py:270-367 (p_mean_variance, p_sample, p_sample_loop)"""
    total = buffers["betas"].shape[0]
    b = shape[0]
    z = noise_fn(-1, tuple(shape)) if noise_fn else torch.randn(shape)
    for i, t_idx in enumerate(list(reversed(range(total)))[:num_steps]):
        t = torch.full((b,), t_idx, dtype=torch.long, device=z.device)
        eps = model(z, t, cond)
        ex = lambda name: buffers[name][t_idx].float()
        z0 = (z - ex("sqrt_one_minus_alphas_cumprod") * eps) / ex("sqrt_alphas_cumprod")
        z0 = torch.clamp(z0, -1.0, 1.0)
        mean = ex("posterior_mean_coef1") * z0 + ex("posterior_mean_coef2") * z
        noise = noise_fn(i, tuple(shape)) if noise_fn else torch.randn_like(z)
        nz = 0.0 if t_idx == 0 else 1.0
        z = mean + nz * torch.exp(0.5 * ex("posterior_log_variance_clipped")) * noise
        if trajectory is not None:
            trajectory.append(z.clone())
    return z


def _per_sample(buffers: SD, name: str, t: torch.Tensor, ndim: int):
    """ref models/diffusion.py:369-385 (_extract): gather per-sample values, broadcastable over the latent"""
    return buffers[name].gather(-1, t).float().reshape(t.shape[0], *((1,) * (ndim - 1)))


def predict_z0_from_noise(buffers: SD, z_t, t, noise_pred):
    """ref models/diffusion.py:249-268"""
    return ((z_t - _per_sample(buffers, "sqrt_one_minus_alphas_cumprod", t, z_t.dim()) * noise_pred)
            / _per_sample(buffers, "sqrt_alphas_cumprod", t, z_t.dim()))


def p_mean_variance(model: Callable, buffers: SD, z_t, t, c, clip_denoised: bool = True):
    """ref models/diffusion.py:270-308 -> (mean, variance, log_variance), per-sample t"""
    z0 = predict_z0_from_noise(buffers, z_t, t, model(z_t, t, c))
    if clip_denoised:
        z0 = torch.clamp(z0, -1.0, 1.0)
    mean = (_per_sample(buffers, "posterior_mean_coef1", t, z_t.dim()) * z0
            + _per_sample(buffers, "posterior_mean_coef2", t, z_t.dim()) * z_t)
    return (mean, _per_sample(buffers, "posterior_variance", t, z_t.dim()),
            _per_sample(buffers, "posterior_log_variance_clipped", t, z_t.dim()))


def p_sample(model: Callable, buffers: SD, z_t, t, c, clip_denoised: bool = True, noise=None):
    """ref models/diffusion.py:310-338"""
    mean, _, logvar = p_mean_variance(model, buffers, z_t, t, c, clip_denoised)
    noise = torch.randn_like(z_t) if noise is None else noise
    nonzero = (t != 0).float().view(-1, *([1] * (z_t.dim() - 1)))
    return mean + nonzero * torch.exp(0.5 * logvar) * noise


def trilinear_depth(z, d_out: int):
    """ref models/model.py:284-289"""
    return F.interpolate(z, size=(d_out, z.shape[3], z.shape[4]), mode="trilinear", align_corners=False)


def generate(sd: SD, cfg: dict, v_in, sampler: str, n_steps: int, target_depth: Optional[int],
             noise_fn: Optional[Callable] = None):
    """ref models/model.py:230-343 (fp32; the discarded randn of :303 is the caller's business when a
    noise_fn is injected)."""
    v_in = torch.nan_to_num(v_in.float(), nan=0.0)
    sf = cfg["scaling_factor"]
    z_in = _guard(vae_encode(sd, v_in, sf, "vae."))
    z_c = _guard(trilinear_depth(z_in, target_depth)) if target_depth is not None else z_in
    model = lambda z, t, c: unet_forward(sd, cfg, z, t, c, "unet.")
    bufs = {k[len("diffusion."):]: v for k, v in sd.items() if k.startswith("diffusion.")}
    if sampler == "ddim":
        z0 = ddim_sample(model, bufs, tuple(z_c.shape), z_c, n_steps, noise_fn=noise_fn)
    elif sampler == "ddpm":
        z0 = ddpm_sample(model, bufs, tuple(z_c.shape), z_c, noise_fn=noise_fn)
    else:
        raise ValueError(f"Unknown sampler: {sampler}")
    return _guard(vae_decode(sd, _guard(z0), sf, "vae."))


def q_sample(bufs: SD, z0, t, noise):
    """ref models/diffusion.py:81-106"""
    a = bufs["sqrt_alphas_cumprod"][t].float().view(-1, 1, 1, 1, 1)
    s = bufs["sqrt_one_minus_alphas_cumprod"][t].float().view(-1, 1, 1, 1, 1)
    return a * z0 + s * noise


def training_loss(sd: SD, cfg: dict, z0, cond, t, noise, mask=None, prefix: str = "unet."):
    """ref models/diffusion.py:108-203 (MSE part; t and noise are given instead of drawn).  Differentiable with
    respect to every tensor of `sd` that requires grad."""
    bufs = {k[len("diffusion."):]: v for k, v in sd.items() if k.startswith("diffusion.")}
    B = z0.shape[0]
    z_t = q_sample(bufs, z0, t, noise)
    pred = unet_forward(sd, cfg, z_t, t, cond, prefix)
    ac = bufs["alphas_cumprod"][t]
    snr = ac / (1 - ac + 1e-8)
    w = torch.clamp(snr, max=5.0) / (snr + 1e-8)
    if mask is None:
        per = F.mse_loss(pred, noise, reduction="none").reshape(B, -1).mean(dim=1)
        return (per * w).mean()
    me = mask.unsqueeze(-1).unsqueeze(-1).expand_as(pred)
    masked = ((pred - noise) ** 2) * me
    nv = me.reshape(B, -1).sum(dim=1)
    if bool((nv == nv[0]).all()):
        return ((masked.sum() / me.sum()) * w).mean()
    per = [(masked[i].sum() / nv[i]) * w[i] if nv[i] > 0 else torch.tensor(0.0) for i in range(B)]
    return torch.stack(per).mean()


def model_training_forward(sd: SD, cfg: dict, v_in, v_gt, t, noise, mask=None):
    """ref models/model.py:158-228: frozen-VAE encode, depth upsample of the conditioning, training_loss."""
    sf = cfg["scaling_factor"]
    with torch.no_grad():
        z_in = vae_encode(sd, v_in, sf, "vae.")
        z_gt = vae_encode(sd, v_gt, sf, "vae.")
        z_mask = mask
        if z_in.shape[2] != z_gt.shape[2]:
            z_in = F.interpolate(z_in, size=z_gt.shape[2:], mode="trilinear", align_corners=False)
            if mask is not None:
                z_mask = F.interpolate(mask.float().unsqueeze(-1).unsqueeze(-1), size=(z_gt.shape[2], 1, 1),
                                       mode="nearest").squeeze(-1).squeeze(-1)
    return training_loss(sd, cfg, z_gt, z_in, t, noise, z_mask)


def gaussian_window(d: int, h: int, w: int):
    """ref inference/sampler.py:174-198, 455-479"""
    def ax(n):
        x = torch.arange(n).float() - (n - 1) / 2
        return torch.exp(-(x ** 2) / (2 * (n / 6) ** 2))
    return ax(d)[:, None, None] * ax(h)[None, :, None] * ax(w)[None, None, :]


def window_starts(full: int, size: int, stride: int):
    """ref inference/sampler.py:388-395"""
    return sorted(set(list(range(0, full - size + 1, stride)) + [max(0, full - size)]))


def ddim_stitched(sd: SD, cfg: dict, v_full, n_steps: int, patch, stride, noise_fn: Optional[Callable] = None,
                  target_d: Optional[int] = None):
    """ref inference/sampler.py:338-453.  target_d == patch depth (default) follows the reference line by line
    and is pinned by golden 'stitch.tiny.out'.  target_d != patch depth is the *intended* behaviour (the
    reference raises a shape error there): the window's conditioning latent is upsampled along depth the way
    models/model.py:284-289 does for whole volumes; unpinned by the reference."""
    b, c, dt, hf, wf = v_full.shape
    pd, ph, pw = patch
    td = pd if target_d is None else int(target_d)
    ratio = td / pd
    acc = torch.zeros(b, c, int(dt * ratio), hf, wf, device=v_full.device)
    wmap = torch.zeros_like(acc)
    win = gaussian_window(td, ph, pw).view(1, 1, td, ph, pw).to(v_full.device)
    model = lambda z, t, cnd: unet_forward(sd, cfg, z, t, cnd, "unet.")
    bufs = {k[len("diffusion."):]: v for k, v in sd.items() if k.startswith("diffusion.")}
    sf = cfg["scaling_factor"]
    for ds in window_starts(dt, pd, stride[0]):
        for hs in window_starts(hf, ph, stride[1]):
            for ws in window_starts(wf, pw, stride[2]):
                vp = v_full[:, :, ds:ds + pd, hs:hs + ph, ws:ws + pw]
                zc = vae_encode(sd, vp, sf, "vae.")
                if td != pd:
                    zc = trilinear_depth(zc, td)
                z0 = ddim_sample(model, bufs, tuple(zc.shape), zc, n_steps, noise_fn=noise_fn)
                out = vae_decode(sd, z0, sf, "vae.")
                d0 = int(ds * ratio)
                acc[:, :, d0:d0 + td, hs:hs + ph, ws:ws + pw] += out * win
                wmap[:, :, d0:d0 + td, hs:hs + ph, ws:ws + pw] += win
    return acc / (wmap + 1e-8)


def ssim_box(a, b, window: int = 11, max_val: float = 1.0) -> float:
    """ref utils/metrics.py:48-121 for (B,C,H,W) inputs; (B,C,D,H,W) = mean over slices."""
    if a.ndim == 5:
        vals = [ssim_box(a[:, :, i], b[:, :, i], window, max_val) for i in range(a.shape[2])]
        return sum(vals) / len(vals) if vals else 0.0
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    pool = lambda t: F.avg_pool2d(t, window, stride=1, padding=window // 2)
    mu1, mu2 = pool(a), pool(b)
    v1 = torch.clamp(pool(a * a) - mu1 * mu1, min=0.0)
    v2 = torch.clamp(pool(b * b) - mu2 * mu2, min=0.0)
    v12 = pool(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * v12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (v1 + v2 + c2) + 1e-8)
    m = torch.clamp(m, 0.0, 1.0)
    return 0.0 if bool(torch.isnan(m).any()) else float(m.mean())


def psnr(a, b, max_val: float) -> float:
    """ref utils/metrics.py:14-44"""
    mse = torch.clamp(torch.mean((a - b) ** 2), min=1e-8)
    val = 20 * torch.log10(torch.tensor(max_val) / torch.sqrt(mse))
    return float(torch.clamp(val, 0.0, 100.0))


# ---- deterministic, machine-independent pseudo-random data ----------------------------------------------
def _hash_uniform(numel: int, key: float) -> torch.Tensor:
    """u[i] = frac(sin(12.9898 i + key) * 43758.5453) in float64: white, reproducible on any machine
    (no RNG state), so the golden generator and the tests rebuild identical tensors."""
    i = torch.arange(numel, dtype=torch.float64)
    v = torch.sin(i * 12.9898 + key) * 43758.5453
    return v - torch.floor(v)


def formula_input(shape, k: int) -> torch.Tensor:
    """uniform(-sqrt(3), sqrt(3)) (unit variance) test input number k"""
    n = int(np.prod(shape))
    return ((2.0 * _hash_uniform(n, 78.233 * (k + 1)) - 1.0) * math.sqrt(3.0)).reshape(shape).float()


def formula_noise(step: int, shape) -> torch.Tensor:
    """approximately N(0,1) injected sampler noise for draw `step` (-1 = initial latent)"""
    n = int(np.prod(shape))
    acc = sum(_hash_uniform(n, 31.7 * (step + 3) + 7.13 * j) for j in range(4))
    return ((acc - 2.0) * math.sqrt(3.0)).reshape(shape).float()


def formula_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0) -> SD:
    """Pseudo-random weights in PyTorch's default-init ranges (uniform +-1/sqrt(fan_in) for conv/linear
    weights and biases, GroupNorm weight 1 +- 0.1, bias +-0.1), generated from a closed formula so that
    fixtures carry expected outputs only."""
    sd = {}
    for n, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        numel = int(np.prod(shape)) if len(shape) else 1
        u = 2.0 * _hash_uniform(numel, 3.11 * (n + 1) + 0.77 * seed) - 1.0
        is_norm = (".norm." in name or ".conv2.1." in name or "conv_out.0." in name) and len(shape) == 1
        if is_norm:
            v = (1.0 + 0.1 * u) if name.endswith(".weight") else 0.1 * u
        elif name.endswith(".weight"):
            if "up_samples" in name or "upsample" in name:   # ConvTranspose3d: (cin, cout, k...)
                fan_in = shape[0] * int(np.prod(shape[2:])) / 4.0
            else:
                fan_in = int(np.prod(shape[1:]))
            v = u / math.sqrt(max(fan_in, 1))
        elif name.endswith(".bias"):
            v = 0.1 * u
        else:
            v = u
        sd[name] = v.reshape(shape).float()
    return sd
