"""Shared test utilities (formula inputs / noise / weights come from oracle.ref_ops)."""
import numpy as np
import torch

from oracle import ref_ops as R


formula_input = R.formula_input
formula_noise = R.formula_noise


def formula_sd(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    return R.formula_state_dict(shapes, seed)


def load_formula(module, seed):
    sd = formula_sd(module, seed)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf16_round(t):
    return t.to(torch.bfloat16).float()


TINY_UNET = dict(latent_dim=8, model_channels=32, num_res_blocks=1, attention_levels=[1], channel_mult=(1, 2),
                 num_heads=4, time_embed_dim=64)
MID_UNET = dict(latent_dim=4, model_channels=32, num_res_blocks=2, attention_levels=[1, 2], channel_mult=(1, 2, 4),
                num_heads=8, time_embed_dim=128)
LEGACY163_UNET = dict(latent_dim=4, model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=(1, 2, 4),
                      num_heads=8, time_embed_dim=1024)   # the 163,410,692-parameter flat-config U-Net
TINY_CFG = {'in_channels': 1, 'latent_dim': 8, 'vae_base_channels': 16, 'vae_scaling_factor': 1.0,
            'unet_model_channels': 32, 'unet_num_res_blocks': 1, 'unet_attention_levels': [1],
            'unet_channel_mult': [1, 2], 'unet_num_heads': 4, 'unet_time_embed_dim': 64,
            'noise_schedule': 'cosine', 'diffusion_timesteps': 1000}


def unet_cfg(kw):
    return dict(model_channels=kw["model_channels"], num_res_blocks=kw["num_res_blocks"],
                attention_levels=list(kw["attention_levels"]), channel_mult=list(kw["channel_mult"]),
                num_heads=kw["num_heads"])


def tiny_model_sd(pkg):
    """Formula-initialised full tiny model, as the golden generator builds it (diffusion buffers real)."""
    model = pkg.VideoToVideoDiffusion(TINY_CFG)
    sd = formula_sd(model, 11)
    for k, v in pkg.GaussianDiffusion('cosine', 1000).state_dict().items():
        sd["diffusion." + k] = v
    model.load_state_dict(sd, strict=True)
    model.eval()
    cfg = dict(model_channels=32, num_res_blocks=1, attention_levels=[1], channel_mult=[1, 2], num_heads=4,
               scaling_factor=1.0)
    return model, sd, cfg
