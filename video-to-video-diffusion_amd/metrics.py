"""Validation metrics on device (mirror of reference utils/metrics.py:14-193).

`Trainer.validate*` and the evaluation scripts call these right after `generate`; here the per-slice squared error
and the 11x11 box-window SSIM map are reduced by one HIP launch over the whole volume (ctsi_slice_metrics), and only
the per-slice scalars come back to the host, where the reference's clamps / dB conversion are applied.
Return types and edge cases follow the reference: python floats, PSNR clamped to [0, 100] with the MSE floored at
1e-8, SSIM 0.0 when NaNs are present, `calculate_video_metrics` returning zeros + empty lists on NaN inputs.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from .engine import Ctx, _ptr
from .lib import CtsiError


def _as_device_f32(t: torch.Tensor) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise CtsiError("metrics run on the HIP engine: pass ROCm tensors (there is no CPU path in the product; "
                        "the oracle under oracle/ is test infrastructure)")
    return t.detach().to(torch.float32).contiguous()


def _slice_stats(a5: torch.Tensor, b5: torch.Tensor, window: int, max_val: float) -> np.ndarray:
    """(n,c,d,h,w) fp32 device tensors -> float64 array (d, 4): mse, mean ssim, nan count a, nan count b."""
    n, c, d, h, w = a5.shape
    ctx = Ctx.get(a5.device)
    out = torch.empty((d, 4), dtype=torch.float64, device=a5.device)
    with ctx.scope():
        ws = torch.empty(max(ctx.lib.slice_metrics_workspace_doubles(n, c, d, h, w), 1), dtype=torch.float64,
                         device=a5.device)
        ctx.lib.slice_metrics(_ptr(a5), _ptr(b5), n, c, d, h, w, int(window), float(max_val), _ptr(ws), _ptr(out),
                              ctx.sptr)
    return out.cpu().numpy()


def _psnr_from_mse(mse: float, max_val: float) -> float:
    mse = max(float(np.float32(mse)), 1e-8)
    return float(min(max(20.0 * math.log10(max_val / math.sqrt(mse)), 0.0), 100.0))


def calculate_psnr(img1, img2, max_val=1.0) -> float:
    """PSNR in dB over all elements of two equally shaped tensors (metrics.py:14-45)."""
    a, b = _as_device_f32(img1), _as_device_f32(img2)
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch: {tuple(a.shape)} vs {tuple(b.shape)}")
    cols = a.shape[-1] if a.dim() else 1
    rows = a.numel() // max(cols, 1)
    st = _slice_stats(a.reshape(1, 1, 1, rows, cols), b.reshape(1, 1, 1, rows, cols), 1, max_val)
    if st[0, 2] > 0 or st[0, 3] > 0:
        return float("nan")
    return _psnr_from_mse(st[0, 0], max_val)


def calculate_ssim(img1, img2, window_size=11, max_val=1.0) -> float:
    """Mean box-window SSIM of (B,C,H,W) images or, slice by slice, of (B,C,D,H,W) volumes (metrics.py:48-121)."""
    a, b = _as_device_f32(img1), _as_device_f32(img2)
    if a.shape != b.shape or a.dim() not in (4, 5):
        raise ValueError(f"expected two (B,C,H,W) or (B,C,D,H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    if a.dim() == 4:
        a, b = a.unsqueeze(2), b.unsqueeze(2)
    st = _slice_stats(a, b, window_size, max_val)
    vals = [0.0 if (r[2] > 0 or r[3] > 0) else float(r[1]) for r in st]
    return sum(vals) / len(vals) if vals else 0.0


def calculate_video_metrics(video1, video2, max_val=1.0) -> Dict[str, object]:
    """Per-frame and mean PSNR / SSIM of (B,C,T,H,W) or (C,T,H,W) volumes (metrics.py:124-193)."""
    a, b = _as_device_f32(video1), _as_device_f32(video2)
    if a.dim() == 4:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
    if a.shape != b.shape or a.dim() != 5:
        raise ValueError(f"expected two (B,C,T,H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    st = _slice_stats(a, b, 11, max_val)
    if st[:, 2].sum() > 0 or st[:, 3].sum() > 0:      # NaN anywhere in either volume
        return {'psnr': 0.0, 'ssim': 0.0, 'psnr_per_frame': [], 'ssim_per_frame': []}
    psnr_values: List[float] = [_psnr_from_mse(r[0], max_val) for r in st]
    ssim_values: List[float] = [float(r[1]) for r in st]
    return {'psnr': float(np.mean(psnr_values)) if psnr_values else 0.0,
            'ssim': float(np.mean(ssim_values)) if ssim_values else 0.0,
            'psnr_per_frame': psnr_values, 'ssim_per_frame': ssim_values}
