"""SliceInterpolationVAE / VideoVAE — host-side mirror of reference models/vae.py.

Module tree, parameter names and construction order follow the reference (state-dict compatible);
`encode` / `decode` / `forward` run on the HIP engine (engine.VAEEncodeProgram / VAEDecodeProgram).
4x spatial compression in H, W; depth is never resampled (vae.py:1-14).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .engine import Ctx, VAEDecodeProgram, VAEEncodeProgram, cached_program
from .lib import CtsiError
from .unet3d import _EngineOnly


class Conv3DBlock(_EngineOnly):
    """conv -> GroupNorm(8) -> SiLU (vae.py:22-35)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding)
        self.norm = nn.GroupNorm(8, out_channels)
        self.act = nn.SiLU()


class ResBlock3D(_EngineOnly):
    """x + (conv-GN-SiLU, conv-GN)(x), then SiLU (vae.py:38-56)."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = Conv3DBlock(channels, channels)
        self.conv2 = nn.Sequential(nn.Conv3d(channels, channels, kernel_size=3, padding=1),
                                   nn.GroupNorm(8, channels))
        self.act = nn.SiLU()


class DownsampleBlock(_EngineOnly):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=(3, 4, 4), stride=(1, 2, 2),
                              padding=(1, 1, 1))
        self.norm = nn.GroupNorm(8, out_channels)
        self.act = nn.SiLU()


class UpsampleBlock(_EngineOnly):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=(3, 4, 4), stride=(1, 2, 2),
                                       padding=(1, 1, 1))
        self.norm = nn.GroupNorm(8, out_channels)
        self.act = nn.SiLU()


class VideoEncoder(_EngineOnly):
    """(B, C, T, H, W) -> (B, latent_dim, T, H/4, W/4) (vae.py:100-147)."""

    def __init__(self, in_channels=3, latent_dim=4, base_channels=64):
        super().__init__()
        b = base_channels
        self.conv_in = Conv3DBlock(in_channels, b)
        self.down1 = nn.Sequential(ResBlock3D(b), ResBlock3D(b), DownsampleBlock(b, 2 * b))
        self.down2 = nn.Sequential(ResBlock3D(2 * b), ResBlock3D(2 * b), DownsampleBlock(2 * b, 4 * b))
        self.mid = nn.Sequential(ResBlock3D(4 * b), ResBlock3D(4 * b))
        self.conv_out = nn.Conv3d(4 * b, 8, kernel_size=3, padding=1)
        self.quant_conv = nn.Conv3d(8, latent_dim, kernel_size=1)


class VideoDecoder(_EngineOnly):
    """(B, latent_dim, T, h, w) -> (B, C, T, 4h, 4w), tanh-bounded (vae.py:150-204)."""

    def __init__(self, latent_dim=4, out_channels=3, base_channels=64):
        super().__init__()
        b = base_channels
        self.post_quant_conv = nn.Conv3d(latent_dim, 8, kernel_size=1)
        self.conv_in = Conv3DBlock(8, 4 * b)
        self.mid = nn.Sequential(ResBlock3D(4 * b), ResBlock3D(4 * b))
        self.up2_upsample = UpsampleBlock(4 * b, 2 * b)
        self.up2_res = nn.Sequential(ResBlock3D(2 * b), ResBlock3D(2 * b))
        self.up3_upsample = UpsampleBlock(2 * b, b)
        self.up3_res = nn.Sequential(ResBlock3D(b), ResBlock3D(b))
        self.conv_out = nn.Conv3d(b, out_channels, kernel_size=3, padding=1)


class SliceInterpolationVAE(nn.Module):
    """encode(x) = encoder(x) * scaling_factor; decode(z) = decoder(z / scaling_factor)."""

    def __init__(self, in_channels=3, latent_dim=4, base_channels=64, scaling_factor=0.18215,
                 gradient_checkpointing=False):
        super().__init__()
        self.latent_dim = latent_dim
        self.in_channels = in_channels
        self.gradient_checkpointing = gradient_checkpointing
        self.encoder = VideoEncoder(in_channels, latent_dim, base_channels)
        self.decoder = VideoDecoder(latent_dim, in_channels, base_channels)
        self.scaling_factor = scaling_factor

    def invalidate_engine_cache(self):
        """Drop the engine's cached programs (packed bf16 weights, captured graphs) -- needed only after weight
        writes torch cannot observe (`p.data[...] = ...`, raw-pointer copies); optimizer steps, `load_state_dict`
        and replaced parameters are detected automatically (engine.Program._fingerprint)."""
        from .engine import invalidate_engine_cache
        invalidate_engine_cache(self)

    def _check(self, t: torch.Tensor, what: str):
        if not t.is_cuda:
            raise CtsiError(f"SliceInterpolationVAE.{what} runs on the HIP engine: move the tensor to a ROCm "
                            "device (there is no CPU path)")
        if t.dim() != 5:
            raise ValueError(f"{what} expects a (B, C, T, H, W) tensor, got shape {tuple(t.shape)}")

    @torch.no_grad()
    def encode(self, x):
        self._check(x, "encode")
        n, c, d, h, w = x.shape
        if c != self.in_channels:
            raise ValueError(f"encode expects {self.in_channels} input channels, got {c}")
        ctx = Ctx.get(x.device)
        with ctx.scope():
            key = ("enc", ctx.device.index, n, d, h, w, float(self.scaling_factor))
            prog = cached_program(self, key, lambda: VAEEncodeProgram(ctx, self, n, d, h, w))
            return prog(x)

    @torch.no_grad()
    def decode(self, z):
        self._check(z, "decode")
        n, c, d, h, w = z.shape
        if c != self.latent_dim:
            raise ValueError(f"decode expects {self.latent_dim} latent channels, got {c}")
        ctx = Ctx.get(z.device)
        comm = getattr(self, "depth_shard_comm", None)
        with ctx.scope():
            if comm is not None and comm.world > 1:
                from .parallel import ShardSpec
                spec = ShardSpec(comm.rank, comm.world, comm, d)
                key = ("dec-shard", ctx.device.index, 1, d, h, w, comm.rank, comm.world, float(self.scaling_factor))
                prog = cached_program(self, key, lambda: VAEDecodeProgram(ctx, self, 1, spec.depth_local, h, w,
                                                                           shard=spec))
                # a depth-sharded program holds one volume: a batch is decoded volume by volume
                return torch.cat([prog(z[i:i + 1]) for i in range(n)], dim=0)
            else:
                key = ("dec", ctx.device.index, n, d, h, w, float(self.scaling_factor))
                prog = cached_program(self, key, lambda: VAEDecodeProgram(ctx, self, n, d, h, w))
            return prog(z)

    def encode_with_posterior(self, x):
        """(mu, logvar) = channel halves of the unscaled encoder output (vae.py:262-287)."""
        z = self.encode(x) / self.scaling_factor
        return torch.chunk(z, 2, dim=1)

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z

    def get_latent_shape(self, volume_shape):
        b, _, t, h, w = volume_shape
        return (b, self.latent_dim, t, h // 4, w // 4)

    @classmethod
    def from_pretrained(cls, model_name_or_path, method='auto', inflate_method='central', strict=True,
                        device='cpu', **kwargs):
        raise NotImplementedError("from_pretrained() is not available. Pretrained VAE loading has been removed. "
                                  "Load weights manually with torch.load() and load_state_dict().")


VideoVAE = SliceInterpolationVAE
