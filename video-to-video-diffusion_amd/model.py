"""VideoToVideoDiffusion façade (mirror of reference models/model.py): config parsing, module
construction, `generate`, checkpoint layout.  Inference runs entirely on the HIP engine."""
from __future__ import annotations

import logging

import torch
import torch.nn as nn
import torch.nn.functional as F

from .diffusion import GaussianDiffusion
from .engine import Ctx, nan_to_num_, trilinear_depth
from .lib import CtsiError
from .unet3d import UNet3D
from .vae import VideoVAE

logger = logging.getLogger(__name__)


class VideoToVideoDiffusion(nn.Module):
    def __init__(self, config, load_pretrained=False):
        super().__init__()
        pre = config.get('pretrained', {})
        use_pretrained = pre.get('use_pretrained', False) or load_pretrained
        grad_ckpt = config.get('hardware', {}).get('gradient_checkpointing',
                                                   config.get('gradient_checkpointing', False))
        # VAE keys are looked up inside config['model'] first, then at top level (model.py:58-62, 86-90)
        mc = config.get('model', config)

        def vae_arg(key, default):
            return mc.get(key, config.get(key, default))

        pre_vae = pre.get('vae', {})
        if use_pretrained and pre_vae.get('enabled', False):
            if pre_vae.get('checkpoint_path'):
                defaults = dict(in_channels=1, vae_base_channels=128, latent_dim=8, vae_scaling_factor=1.0)
            elif pre_vae.get('model_name'):
                VideoVAE.from_pretrained(pre_vae['model_name'])  # raises NotImplementedError (vae.py:308-321)
                defaults = {}
            else:
                raise ValueError("VAE enabled but neither checkpoint_path nor model_name specified in config")
        else:
            defaults = dict(in_channels=3, vae_base_channels=64, latent_dim=4, vae_scaling_factor=0.18215)
        self.vae = VideoVAE(in_channels=vae_arg('in_channels', defaults['in_channels']),
                            latent_dim=vae_arg('latent_dim', defaults['latent_dim']),
                            base_channels=vae_arg('vae_base_channels', defaults['vae_base_channels']),
                            scaling_factor=vae_arg('vae_scaling_factor', defaults['vae_scaling_factor']),
                            gradient_checkpointing=grad_ckpt)
        # U-Net keys are read from the TOP level of the config only (model.py:103-112): a YAML that nests
        # them under `model:` silently gets the defaults below.  Kept so checkpoint['config'] rebuilds
        # the same network.
        self.unet = UNet3D(latent_dim=self.vae.latent_dim,
                           model_channels=config.get('unet_model_channels', 128),
                           num_res_blocks=config.get('unet_num_res_blocks', 2),
                           attention_levels=config.get('unet_attention_levels', [1, 2]),
                           channel_mult=tuple(config.get('unet_channel_mult', [1, 2, 4, 4])),
                           num_heads=config.get('unet_num_heads', 4),
                           time_embed_dim=config.get('unet_time_embed_dim', 512),
                           use_checkpoint=grad_ckpt)
        self.diffusion = GaussianDiffusion(noise_schedule=config.get('noise_schedule', 'cosine'),
                                           timesteps=config.get('diffusion_timesteps', 1000),
                                           beta_start=config.get('beta_start', 0.0001),
                                           beta_end=config.get('beta_end', 0.02))
        self.config = config
        self.use_pretrained = use_pretrained

    def invalidate_engine_cache(self):
        """Drop the engine's cached programs (packed bf16 weights, captured graphs) -- needed only after weight
        writes torch cannot observe (`p.data[...] = ...`, raw-pointer copies); optimizer steps, `load_state_dict`
        and replaced parameters are detected automatically (engine.Program._fingerprint)."""
        from .engine import invalidate_engine_cache
        invalidate_engine_cache(self)

    def encode_videos(self, v_in, v_gt=None):
        z_in = self.vae.encode(v_in)
        if v_gt is not None:
            return z_in, self.vae.encode(v_gt)
        return z_in

    def decode_latent(self, z):
        return self.vae.decode(z)

    def forward(self, v_in, v_gt, mask=None, t=None, noise=None):
        """Training forward (model.py:158-228): frozen-VAE encode of both volumes, depth upsample of the
        conditioning when the depths differ, then `diffusion.training_loss` on the HIP engine.  Returns
        (loss, metrics); `loss.backward()` runs the engine's backward.  `t=` / `noise=` are test hooks."""
        if not v_in.is_cuda:
            raise CtsiError("the training forward runs on the HIP engine: move the inputs to a ROCm device")
        ctx = Ctx.get(v_in.device)
        with torch.no_grad():
            z_in = self.vae.encode(v_in)
            z_gt = self.vae.encode(v_gt)
            if z_in.shape[2] != z_gt.shape[2]:
                with ctx.scope():
                    z_cond = trilinear_depth(ctx, z_in, int(z_gt.shape[2]))
                z_mask = None
                if mask is not None:   # nearest-neighbour pick of the (B, C, T) mask at the latent depth (model.py:205-212)
                    z_mask = F.interpolate(mask.float().unsqueeze(-1).unsqueeze(-1), size=(z_gt.shape[2], 1, 1),
                                           mode='nearest').squeeze(-1).squeeze(-1)
            else:
                z_cond, z_mask = z_in, mask
        loss, loss_dict = self.diffusion.training_loss(self.unet, z_gt, z_cond, mask=z_mask, vae=self.vae, v_gt=v_gt,
                                                       use_ssim=False, ssim_weight=0.0, t=t, noise=noise)
        return loss, {'loss': loss.item(), **loss_dict}

    @torch.no_grad()
    def generate(self, v_in, sampler, num_inference_steps=20, guidance_scale=1.0, target_depth=None,
                 noise_fn=None):
        """thick slices (B, C, T_in, H, W) -> thin slices (B, C, T_out, H, W), fp32.

        encode -> trilinear depth upsample of the conditioning -> DDIM/DDPM -> decode, with the
        reference's nan_to_num guards applied unconditionally on device (model.py:230-343).
        `guidance_scale` is accepted and ignored, as in the reference."""
        if sampler not in ('ddpm', 'ddim'):
            raise ValueError(f"Unknown sampler: {sampler}")
        if not v_in.is_cuda:
            raise CtsiError("generate runs on the HIP engine: move the input to a ROCm device")
        device = v_in.device
        ctx = Ctx.get(device)
        v_in = torch.nan_to_num(v_in.float(), nan=0.0)
        z_in = self.vae.encode(v_in)
        with ctx.scope():
            nan_to_num_(ctx, z_in)
            if target_depth is not None:
                z_cond = trilinear_depth(ctx, z_in, int(target_depth))
                nan_to_num_(ctx, z_cond)
            else:
                z_cond = z_in
        latent_shape = tuple(z_cond.shape)
        if noise_fn is None:
            torch.randn(latent_shape, device=device)  # model.py:303 draws (and discards) one latent
        if sampler == 'ddpm':
            z_0 = self.diffusion.p_sample_loop(self.unet, latent_shape, z_cond, device, progress=True,
                                               noise_fn=noise_fn)
        else:
            from .sampler import DDIMSampler
            z_0 = DDIMSampler(self.diffusion, self.unet).sample(latent_shape, z_cond, num_inference_steps,
                                                                device, noise_fn=noise_fn)
        with ctx.scope():
            nan_to_num_(ctx, z_0)
        v_out = self.vae.decode(z_0)
        with ctx.scope():
            nan_to_num_(ctx, v_out)
        return v_out

    def save_checkpoint(self, path, optimizer=None, scheduler=None, scaler=None, epoch=None, global_step=None,
                        current_phase=None, best_loss=None, **kwargs):
        """Same dict layout as the reference (model.py:362-387)."""
        ckpt = {'model_state_dict': self.state_dict(), 'config': self.config}
        for key, obj in (('optimizer_state_dict', optimizer), ('scheduler_state_dict', scheduler),
                         ('scaler_state_dict', scaler)):
            if obj is not None:
                ckpt[key] = obj.state_dict()
        for key, val in (('epoch', epoch), ('global_step', global_step), ('current_phase', current_phase),
                         ('best_loss', best_loss)):
            if val is not None:
                ckpt[key] = val
        ckpt.update(kwargs)
        torch.save(ckpt, path)
        print(f"Checkpoint saved to {path}")

    def count_parameters(self):
        total = sum(p.numel() for p in self.parameters())
        return {
            'total': total,
            'trainable': sum(p.numel() for p in self.parameters() if p.requires_grad),
            'vae': sum(p.numel() for p in self.vae.parameters()),
            'vae_trainable': sum(p.numel() for p in self.vae.parameters() if p.requires_grad),
            'unet': sum(p.numel() for p in self.unet.parameters() if p.requires_grad),
            'diffusion': 0,
        }
