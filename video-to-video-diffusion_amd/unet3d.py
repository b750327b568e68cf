"""UNet3D — host-side mirror of the reference denoiser's interface (reference models/unet3d.py).

Only the *module tree* lives here: parameter names, shapes and construction order follow the
reference so that `load_state_dict(strict=True)` of a reference checkpoint works and
`torch.manual_seed(s)` before construction yields the same initial weights.  The arithmetic is not
implemented in Python: `UNet3D.forward` compiles the tree into a libctsi program (engine.UNetProgram)
and runs it on the HIP stream.  Calling it on CPU tensors raises — there is no fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .engine import Ctx, UNetProgram, cached_program
from .lib import CtsiError

_GROUP_CANDIDATES = (32, 16, 8, 4, 2, 1)


def _largest_group_count(channels: int) -> int:
    """Largest of 32,16,...,1 dividing `channels` (reference unet3d.py:62-68 and twins)."""
    return next((g for g in _GROUP_CANDIDATES if channels % g == 0), 1)


class _EngineOnly(nn.Module):
    """Sub-blocks carry parameters only; they execute as part of the enclosing network's program."""

    def forward(self, *args, **kwargs):  # pragma: no cover - defensive
        raise CtsiError(f"{type(self).__name__} is executed by the HIP engine as part of UNet3D / the VAE; "
                        "call the enclosing network instead")


class SinusoidalPositionEmbeddings(_EngineOnly):
    """Parameter-free placeholder keeping `time_mlp.{1,3}` at the reference's Sequential indices
    (unet3d.py:18-32; evaluated by ctsi_time_embed_fwd)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim


class TimeEmbedding(_EngineOnly):
    def __init__(self, dim: int, time_dim: int):
        super().__init__()
        self.time_mlp = nn.Sequential(SinusoidalPositionEmbeddings(dim), nn.Linear(dim, time_dim), nn.SiLU(),
                                      nn.Linear(time_dim, time_dim))


class Conv3DBlock(_EngineOnly):
    """conv -> GroupNorm -> SiLU (unet3d.py:51-74): 8 groups when C % 8 == 0, else the largest
    divisor from (32..1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding)
        groups = min(8, out_channels) if out_channels % 8 == 0 else _largest_group_count(out_channels)
        self.norm = nn.GroupNorm(groups, out_channels)
        self.act = nn.SiLU()


class ResBlock3D(_EngineOnly):
    """unet3d.py:77-133."""

    def __init__(self, in_channels, out_channels, time_dim):
        super().__init__()
        self.conv1 = Conv3DBlock(in_channels, out_channels)
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, out_channels))
        self.conv2 = nn.Sequential(nn.Conv3d(out_channels, out_channels, kernel_size=3, padding=1),
                                   nn.GroupNorm(_largest_group_count(out_channels), out_channels))
        self.residual_conv = (nn.Conv3d(in_channels, out_channels, kernel_size=1)
                              if in_channels != out_channels else nn.Identity())
        self.act = nn.SiLU()


class TemporalAttention(_EngineOnly):
    """unet3d.py:136-194 (attention along depth; see csrc/attention.hip for how it is evaluated)."""

    def __init__(self, channels, num_heads=4):
        super().__init__()
        self.num_heads = num_heads
        self.channels = channels
        self.head_dim = channels // num_heads
        assert channels % num_heads == 0, "channels must be divisible by num_heads"
        self.norm = nn.GroupNorm(_largest_group_count(channels), channels)
        self.qkv = nn.Conv3d(channels, channels * 3, kernel_size=1)
        self.proj_out = nn.Conv3d(channels, channels, kernel_size=1)


class Downsample3D(_EngineOnly):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, in_channels if out_channels is None else out_channels,
                              kernel_size=(3, 4, 4), stride=(1, 2, 2), padding=(1, 1, 1))


class Upsample3D(_EngineOnly):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.ConvTranspose3d(channels, channels, kernel_size=(3, 4, 4), stride=(1, 2, 2),
                                       padding=(1, 1, 1))


class UNet3D(nn.Module):
    """forward(x, t, c) -> predicted noise, all (B, latent_dim, T, h, w) fp32 NCDHW; t int64 (B,).

    Extra attribute `attention_mode` ('fast' | 'exact') selects how TemporalAttention's
    rowsum(softmax) factor is obtained; both reproduce the reference einsum (see attention.hip).
    """

    def __init__(self, latent_dim=4, model_channels=128, num_res_blocks=2, attention_levels=[1, 2],
                 channel_mult=(1, 2, 4, 4), num_heads=4, time_embed_dim=512, use_checkpoint=False):
        super().__init__()
        self.latent_dim = latent_dim
        self.model_channels = model_channels
        self.num_res_blocks = num_res_blocks
        self.attention_levels = attention_levels
        self.channel_mult = channel_mult
        self.num_levels = len(channel_mult)
        self.use_checkpoint = use_checkpoint
        self.attention_mode = "fast"

        self.time_embed = TimeEmbedding(model_channels, time_embed_dim)
        self.conv_in = nn.Conv3d(latent_dim * 2, model_channels, kernel_size=3, padding=1)

        def stage(cin, cout, with_attn):
            layers = [ResBlock3D(cin, cout, time_embed_dim)]
            if with_attn:
                layers.append(TemporalAttention(cout, num_heads))
            return nn.ModuleList(layers)

        self.down_blocks = nn.ModuleList()
        self.down_samples = nn.ModuleList()
        ch = model_channels
        for level, mult in enumerate(channel_mult):
            width = model_channels * mult
            blocks = nn.ModuleList()
            for _ in range(num_res_blocks):
                blocks.append(stage(ch, width, level in attention_levels))
                ch = width
            self.down_blocks.append(blocks)
            self.down_samples.append(Downsample3D(ch, ch) if level < self.num_levels - 1 else nn.Identity())

        self.mid_block1 = ResBlock3D(ch, ch, time_embed_dim)
        self.mid_attn = TemporalAttention(ch, num_heads)
        self.mid_block2 = ResBlock3D(ch, ch, time_embed_dim)

        self.up_blocks = nn.ModuleList()
        self.up_samples = nn.ModuleList()
        for level, mult in enumerate(reversed(channel_mult)):
            width = model_channels * mult
            src_level = self.num_levels - 1 - level
            blocks = nn.ModuleList()
            for i in range(num_res_blocks + 1):
                cin = ch + model_channels * channel_mult[src_level] if i == 0 else ch
                blocks.append(stage(cin, width, src_level in attention_levels))
                ch = width
            self.up_blocks.append(blocks)
            self.up_samples.append(Upsample3D(ch) if level < self.num_levels - 1 else nn.Identity())

        self.conv_out = nn.Sequential(nn.GroupNorm(_largest_group_count(ch), ch), nn.SiLU(),
                                      nn.Conv3d(ch, latent_dim, kernel_size=3, padding=1))

    @staticmethod
    def _get_num_groups(channels):
        return _largest_group_count(channels)

    # ---- engine plumbing ---------------------------------------------------------------------------
    def invalidate_engine_cache(self):
        """Drop the engine's cached programs (packed bf16 weights, captured graphs) -- needed only after weight
        writes torch cannot observe (`p.data[...] = ...`, raw-pointer copies); optimizer steps, `load_state_dict`
        and replaced parameters are detected automatically (engine.Program._fingerprint)."""
        from .engine import invalidate_engine_cache
        invalidate_engine_cache(self)

    def program(self, ctx: Ctx, n: int, d: int, h: int, w: int, max_rows: int) -> UNetProgram:
        key = ("unet", ctx.device.index, n, d, h, w, max_rows, self.attention_mode)
        return cached_program(self, key, lambda: UNetProgram(ctx, self, n, d, h, w, max_rows,
                                                             self.attention_mode))

    @torch.no_grad()
    def forward(self, x, t, c):
        if not (x.is_cuda and c.is_cuda):
            raise CtsiError("UNet3D.forward runs on the HIP engine: move the tensors to a ROCm device "
                            "(there is no CPU path; the oracle under oracle/ is test infrastructure only)")
        n, L, d, h, w = x.shape
        if L != self.latent_dim or tuple(c.shape) != tuple(x.shape):
            raise ValueError(f"expected x and c of shape (B, {self.latent_dim}, T, h, w), got {tuple(x.shape)} "
                             f"and {tuple(c.shape)}")
        ctx = Ctx.get(x.device)
        with ctx.scope():
            prog = self.program(ctx, n, d, h, w, max_rows=n)
            prog_plain = prog
            if getattr(prog, "sampler_kind", None) is not None:
                raise CtsiError("internal: plain forward reuses a sampler program")
            prog_plain.load_latents(x, c)
            prog_plain.set_schedule([int(v) for v in t.reshape(-1).tolist()])
            prog_plain.launch()
            return prog_plain.eps_ncdhw()
