"""Batch inference helpers (mirror of reference inference/generate.py:98-226; the video-file I/O
wrapper `generate_video` of the legacy RGB path is out of scope)."""
from __future__ import annotations

import torch

from .sampler import DDIMSampler, DDPMSampler


def _sample(model, z_cond, sampler_type, num_inference_steps, device, progress, noise_fn=None):
    if sampler_type == 'ddim':
        return DDIMSampler(model.diffusion, model.unet).sample(z_cond.shape, z_cond, num_inference_steps, device,
                                                               progress=progress, noise_fn=noise_fn)
    if sampler_type == 'ddpm':
        return DDPMSampler(model.diffusion, model.unet).sample(z_cond.shape, z_cond, device, progress=progress,
                                                               noise_fn=noise_fn)
    raise ValueError(f"Unknown sampler type: {sampler_type}")


@torch.no_grad()
def generate_batch(model, input_videos, sampler_type='ddim', num_inference_steps=20, device='cuda',
                   noise_fn=None):
    """encode -> sample at the input's latent shape (no depth change) -> decode (generate.py:98-155)."""
    if sampler_type not in ('ddim', 'ddpm'):
        raise ValueError(f"Unknown sampler type: {sampler_type}")
    model.eval()
    model.to(device)
    input_videos = input_videos.to(device)
    print(f"Generating batch of {input_videos.shape[0]} videos...")
    z_in = model.vae.encode(input_videos)
    z_0 = _sample(model, z_in, sampler_type, num_inference_steps, device, True, noise_fn)
    return model.vae.decode(z_0)


@torch.no_grad()
def interpolate_videos(model, video_a, video_b, num_interpolations=5, sampler_type='ddim',
                       num_inference_steps=20, device='cuda'):
    """Latent-space lerp between two clips used as conditioning (generate.py:158-226)."""
    model.eval()
    model.to(device)
    z_a = model.vae.encode(video_a.unsqueeze(0).to(device))
    z_b = model.vae.encode(video_b.unsqueeze(0).to(device))
    outs = []
    for alpha in torch.linspace(0, 1, num_interpolations).to(device):
        z_mix = (1 - alpha) * z_a + alpha * z_b
        kind = 'ddim' if sampler_type == 'ddim' else 'ddpm'
        z_0 = _sample(model, z_mix, kind, num_inference_steps, device, False)
        outs.append(model.vae.decode(z_0).squeeze(0))
    return outs
