"""GaussianDiffusion — schedules and DDPM sampling (mirror of reference models/diffusion.py).

The ten (timesteps,) fp32 buffers are built with the same torch ops, in the same order, as the
reference (diffusion.py:27-79) so they are bit-identical and checkpoint-compatible.  The reverse
process runs on the HIP engine: `p_sample_loop` drives a captured U-Net step graph and the
ctsi_ddpm_step kernel; single steps with per-sample timesteps (`p_mean_variance`, `p_sample`,
`_predict_z_0_from_noise`) use ctsi_ddpm_posterior.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .lib import CtsiError


class GaussianDiffusion(nn.Module):
    def __init__(self, noise_schedule='cosine', timesteps=1000, beta_start=0.0001, beta_end=0.02):
        super().__init__()
        self.timesteps = timesteps
        self.noise_schedule = noise_schedule
        if noise_schedule == 'linear':
            betas = self._linear_beta_schedule(timesteps, beta_start, beta_end)
        elif noise_schedule == 'cosine':
            betas = self._cosine_beta_schedule(timesteps)
        else:
            raise ValueError(f"Unknown noise schedule: {noise_schedule}")

        alphas = 1.0 - betas
        abar = torch.cumprod(alphas, dim=0)
        abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
        post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
        for name, value in (
            ('betas', betas),
            ('alphas', alphas),
            ('alphas_cumprod', abar),
            ('alphas_cumprod_prev', abar_prev),
            ('sqrt_alphas_cumprod', torch.sqrt(abar)),
            ('sqrt_one_minus_alphas_cumprod', torch.sqrt(1.0 - abar)),
            ('posterior_variance', post_var),
            ('posterior_log_variance_clipped', torch.log(torch.clamp(post_var, min=1e-20))),
            ('posterior_mean_coef1', betas * torch.sqrt(abar_prev) / (1.0 - abar)),
            ('posterior_mean_coef2', (1.0 - abar_prev) * torch.sqrt(alphas) / (1.0 - abar)),
        ):
            self.register_buffer(name, value)

    def _linear_beta_schedule(self, timesteps, beta_start, beta_end):
        return torch.linspace(beta_start, beta_end, timesteps)

    def _cosine_beta_schedule(self, timesteps, s=0.008):
        # abar(x) = cos^2(((x/T) + s)/(1+s) * pi/2), normalised by abar(0); beta clipped to [1e-4, 0.9999]
        grid = torch.linspace(0, timesteps, timesteps + 1)
        abar = torch.cos(((grid / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        abar = abar / abar[0]
        return torch.clip(1 - (abar[1:] / abar[:-1]), 0.0001, 0.9999)

    def _extract(self, a, t, x_shape):
        out = a.gather(-1, t).float()
        return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))

    # ---- forward process (elementwise, torch ops on whatever device the tensors live) --------------
    def q_sample(self, z_0, t, noise=None):
        if noise is None:
            noise = torch.randn_like(z_0)
        z_t = (self._extract(self.sqrt_alphas_cumprod, t, z_0.shape) * z_0 +
               self._extract(self.sqrt_one_minus_alphas_cumprod, t, z_0.shape) * noise)
        return z_t, noise

    # ---- DDPM reverse process on the HIP engine -------------------------------------------------------
    def ddpm_coef_rows(self, t_desc):
        """Per-step coefficient rows for ctsi_ddpm_step, fp32 values taken from the registered buffers
        exactly as p_mean_variance / p_sample read them (diffusion.py:290-336)."""
        idx = torch.as_tensor(list(t_desc), dtype=torch.long, device=self.betas.device)
        rows = torch.zeros(len(idx), 8, dtype=torch.float32, device=self.betas.device)
        rows[:, 0] = self.sqrt_one_minus_alphas_cumprod[idx]
        rows[:, 1] = self.sqrt_alphas_cumprod[idx]
        rows[:, 2] = self.posterior_mean_coef1[idx]
        rows[:, 3] = self.posterior_mean_coef2[idx]
        rows[:, 4] = (idx != 0).float() * torch.exp(0.5 * self.posterior_log_variance_clipped[idx])
        return rows

    @torch.no_grad()
    def p_sample_loop(self, model, shape, c, device, progress=True, noise_fn=None, num_steps=None):
        """z_T ~ N(0, I); for t = T-1..0: one U-Net evaluation + ctsi_ddpm_step (clip to [-1, 1]).

        `noise_fn(i, shape)` (optional, additive kwarg) supplies the initial noise (i = -1) and the
        per-step noise tensors instead of torch.randn; `num_steps` truncates the loop to its first
        steps (both are test hooks; defaults reproduce the reference)."""
        from .sampler import run_sampler  # local import: sampler imports this module
        return run_sampler(self, model, shape, c, device, kind="ddpm",
                           t_desc=list(reversed(range(self.timesteps)))[:num_steps], progress=progress,
                           noise_fn=noise_fn)

    # ---- single reverse steps with per-sample timesteps (diffusion.py:249-338) -------------------------------------
    def _posterior_rows(self, t):
        """One coefficient row per SAMPLE for ctsi_ddpm_posterior, read from the registered buffers exactly where
        _predict_z_0_from_noise / p_mean_variance / p_sample read them."""
        idx = t.reshape(-1).to(device=self.betas.device, dtype=torch.long)
        return self.ddpm_coef_rows(idx.tolist())

    def _posterior(self, z_t, t, eps, noise, clip, want_z0, want_out):
        from .engine import Ctx, _ptr
        if not (z_t.is_cuda and eps.is_cuda):
            raise CtsiError("the reverse-process arithmetic runs on the HIP engine: move the tensors to a ROCm device "
                            "(there is no CPU path)")
        if tuple(eps.shape) != tuple(z_t.shape) or t.reshape(-1).shape[0] != z_t.shape[0]:
            raise ValueError(f"expected noise_pred of shape {tuple(z_t.shape)} and one timestep per sample, got "
                             f"{tuple(eps.shape)} and t of shape {tuple(t.shape)}")
        ctx = Ctx.get(z_t.device)
        n = int(z_t.shape[0])
        z = z_t.detach().to(torch.float32).contiguous()
        e = eps.detach().to(torch.float32).contiguous()
        nz = None if noise is None else noise.detach().to(device=ctx.device, dtype=torch.float32).contiguous()
        coef = self._posterior_rows(t).to(ctx.device, torch.float32).contiguous()
        z0 = torch.empty_like(z) if want_z0 else None
        out = torch.empty_like(z) if want_out else None
        with ctx.scope():
            ctx.lib.ddpm_posterior(_ptr(z), _ptr(e), _ptr(nz), _ptr(z0), _ptr(out), _ptr(coef), n, z.numel() // n,
                                   int(bool(clip)), ctx.sptr)
            for tns in (z, e, nz, coef):
                if tns is not None:
                    tns.record_stream(ctx.stream)
        return z0, out

    @torch.no_grad()
    def _predict_z_0_from_noise(self, z_t, t, noise_pred):
        """z_0 = (z_t - sqrt(1 - abar_t) * noise_pred) / sqrt(abar_t), per-sample t (diffusion.py:249-268)."""
        return self._posterior(z_t, t, noise_pred, None, False, True, False)[0]

    @torch.no_grad()
    def p_mean_variance(self, model, z_t, t, c, clip_denoised=True):
        """(mean, variance, log_variance) of q(z_{t-1} | z_t, z_0_pred) (diffusion.py:270-308).  `model` is any
        `model(z, t, c) -> eps` callable on the ROCm device (the engine's UNet3D evaluates per-sample timesteps); the
        posterior mean is one ctsi_ddpm_posterior launch.  variance / log_variance are the (B,1,1,1,1) buffer gathers
        the reference returns."""
        noise_pred = model(z_t, t, c)
        _, mean = self._posterior(z_t, t, noise_pred, None, clip_denoised, False, True)
        variance = self._extract(self.posterior_variance, t, z_t.shape)
        log_variance = self._extract(self.posterior_log_variance_clipped, t, z_t.shape)
        return mean, variance, log_variance

    @torch.no_grad()
    def p_sample(self, model, z_t, t, c, clip_denoised=True, noise=None):
        """One DDPM step z_t -> z_{t-1} (diffusion.py:310-338); per-sample `t` and `clip_denoised=False` as in the
        reference.  The noise is drawn with `torch.randn_like(z_t)` after the network evaluation, where the reference
        draws it (`noise=` injects it instead: tests).  A batch-uniform clipped step on the engine's own UNet3D takes
        the captured-graph path of the sampling loop (same arithmetic, ctsi_ddpm_step)."""
        from .sampler import _is_engine_unet, run_sampler
        tv = [int(v) for v in t.reshape(-1).tolist()]
        if len(set(tv)) == 1 and clip_denoised and noise is None and _is_engine_unet(model):
            return run_sampler(self, model, tuple(z_t.shape), c, z_t.device, kind="ddpm", t_desc=[tv[0]],
                               progress=False, z_init=z_t)
        noise_pred = model(z_t, t, c)
        if noise is None:
            noise = torch.randn_like(z_t)
        return self._posterior(z_t, t, noise_pred, noise, clip_denoised, False, True)[1]

    def training_loss(self, model, z_0, c, mask=None, vae=None, v_gt=None, use_ssim=False, ssim_weight=0.0,
                      t=None, noise=None):
        """Min-SNR-5 weighted epsilon-prediction loss (diffusion.py:108-247) with forward AND backward on the HIP
        engine: the returned scalar carries an autograd node whose backward launches the engine's gradient kernels
        and feeds the U-Net parameters' .grad.

        t ~ randint(0, T, (B,)) and noise ~ randn_like(z_0) are drawn with torch's generator in the reference's
        order; `t=` / `noise=` (additive kwargs) inject them instead (tests).  The three normalisations of the
        reference (no mask; mask with equal valid counts; mask with per-sample counts) are folded into one
        per-sample factor.  The optional MS-SSIM term (a gradient-free logging term: the reference decodes under no_grad)
        needs the third-party `pytorch_msssim`; without it the reference warns and returns the MSE loss, and so does this
        engine (see the end of the function)."""
        from .train_engine import UNetTrainProgram, train_step
        from .engine import Ctx, cached_program
        if not z_0.is_cuda:
            raise CtsiError("training_loss runs on the HIP engine: move the tensors to a ROCm device")
        B, L, d, h, w = z_0.shape
        device = z_0.device
        if t is None:
            t = torch.randint(0, self.timesteps, (B,), device=device, dtype=torch.long)
        if noise is None:
            noise = torch.randn_like(z_0)
        snr = self.alphas_cumprod[t] / (1 - self.alphas_cumprod[t] + 1e-8)
        snr_weight = torch.clamp(snr, max=5.0) / (snr + 1e-8)
        if mask is not None:
            m = mask.to(device=device, dtype=torch.float32)
            if m.dim() != 3 or m.shape[0] != B or m.shape[2] != d or m.shape[1] not in (1, L):
                raise ValueError(f"mask must be (B, C, T) = ({B}, 1 or {L}, {d}), got {tuple(m.shape)}")
            m = m.expand(B, L, d).contiguous()     # mask.unsqueeze(-1).unsqueeze(-1).expand_as(noise_pred)
            num_valid = m.reshape(B, -1).sum(dim=1) * (h * w)
            if bool((num_valid == num_valid[0]).all()):
                norm = (snr_weight.mean() / num_valid.sum()).expand(B)
            else:
                norm = torch.where(num_valid > 0, snr_weight / (num_valid.clamp(min=1) * B),
                                   torch.zeros_like(snr_weight))
        else:
            m = None
            norm = snr_weight / float(B * L * d * h * w)
        ctx = Ctx.get(device)
        with ctx.scope():
            prog = cached_program(model, ("unet-train", ctx.device.index, B, d, h, w),
                                  lambda: UNetTrainProgram(ctx, model, B, d, h, w))
            prog.set_diffusion(self)
        loss = train_step(prog, z_0.detach().float(), c.detach().float(), t, noise.float(), norm, m)
        loss_dict = {'mse': loss.item()}
        prog.check_errors()     # (the stream is synchronised by the .item() above: a sticky split-K hand-off error of this forward,
                                #  or of the previous step's backward, surfaces here)
        # Optional MS-SSIM term (diffusion.py:204-240).  The reference decodes the predicted z_0 under torch.no_grad(), so the
        # term carries NO gradient: total = (1 - w) * mse + w * (1 - ms_ssim) scales the MSE gradient by (1 - w) and adds a
        # constant.  ms_ssim itself is `pytorch_msssim.ms_ssim` (requirements.txt:24, `pytorch-msssim>=1.0.0`), a third-party
        # package; where it is absent the reference prints the warning below and returns the MSE loss -- so does this
        # engine (model.forward never enables the term: models/model.py:218-219).
        if use_ssim and ssim_weight > 0.0 and vae is not None and v_gt is not None:
            try:
                from pytorch_msssim import ms_ssim
                with torch.no_grad():
                    z_t, _ = self.q_sample(z_0.detach().float(), t, noise.float())
                    eps = torch.empty((B, L, d, h, w), dtype=torch.float32, device=device)
                    with ctx.scope():
                        ctx.lib.ndhwc_f32_to_ncdhw_f32(C.c_void_p(prog.eps.data_ptr()), C.c_void_p(eps.data_ptr()), B, L, d, h,
                                                       w, ctx.sptr)
                    v_pred = vae.decode(self._predict_z_0_from_noise(z_t, t, eps))
                    terms = []
                    for i in range(v_gt.shape[2]):
                        terms.append(1.0 - ms_ssim((v_pred[:, :, i] + 1.0) / 2.0, (v_gt[:, :, i].float() + 1.0) / 2.0,
                                                   data_range=1.0, size_average=True))
                    loss_ssim = torch.stack(terms).mean()
                loss_dict['ssim'] = loss_ssim.item()
                total = (1.0 - ssim_weight) * loss + ssim_weight * loss_ssim
                loss_dict['total'] = total.item()
                return total, loss_dict
            except ImportError:
                print("Warning: pytorch-msssim not installed. Falling back to MSE-only loss.")
            except Exception as e:
                print(f"Warning: MS-SSIM calculation failed: {e}. Using MSE-only loss.")
        loss_dict['total'] = loss_dict['mse']
        return loss, loss_dict
