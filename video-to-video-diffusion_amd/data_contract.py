"""Tensor contract between the reference's data pipeline and the engine.

The data pipeline itself (DICOM / HF download, preprocessing, augmentation: reference data/*.py) is CPU I/O and is NOT
rebuilt.  What the engine must stay compatible with is the format those loaders hand to the model:

  * the per-patient cache file  `case_*.pt` = dict {'input' | 'thick': (1, D_thick, H, W), 'target' | 'thin':
    (1, D_thin, H, W), 'category', 'patient_id'}  (data/patch_slice_interpolation_dataset.py:257-258);
  * the dataset item  {'x_lr', 'x_hr', 'input', 'target', 'category', 'patient_id'} with x_lr == input of shape
    (1, 8, 192, 192) and x_hr == target of shape (1, 48, 192, 192), values in [-1, 1] (:267-274);
  * the collated batch the trainer consumes: `v_in = batch['input']`, `v_gt = batch['target']`, (B, 1, D, H, W)
    (training/trainer.py:209-210), then `loss, metrics = model(v_in, v_gt)`.

These helpers read / build / validate exactly that, so a reference DataLoader can feed `VideoToVideoDiffusion.forward`
and `generate` of this package unchanged.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

ITEM_KEYS = ("x_lr", "x_hr", "input", "target", "category", "patient_id")


def read_patient_cache(path) -> Dict[str, object]:
    """Load one preprocessed patient file; accepts both key spellings of the reference's caches."""
    d = torch.load(path, weights_only=False)
    thick = d.get("input", d.get("thick"))
    thin = d.get("target", d.get("thin"))
    if thick is None or thin is None:
        raise ValueError(f"{path}: expected 'input'/'thick' and 'target'/'thin' volumes, found keys {sorted(d)}")
    for name, v in (("thick", thick), ("thin", thin)):
        if v.dim() != 4 or v.shape[0] != 1:
            raise ValueError(f"{path}: {name} volume must be (1, D, H, W), got {tuple(v.shape)}")
    return {"thick": thick.float(), "thin": thin.float(), "category": d.get("category", "unknown"),
            "patient_id": d.get("patient_id", str(path))}


def aligned_patch(thick: torch.Tensor, thin: torch.Tensor, z_thin: int, y0: int, x0: int, depth_thin: int = 48,
                  depth_thick: int = 8, patch_hw: Tuple[int, int] = (192, 192)) -> Tuple[torch.Tensor, torch.Tensor]:
    """The patch pair the reference's dataset cuts at a given position (its positions are random; here they are
    arguments): thin slices [z_thin, z_thin + depth_thin), the thick slices covering the same physical range
    (index * D_thick / D_thin, at least one) resampled trilinearly to `depth_thick`, same spatial window; a thin
    volume shallower than the patch is padded with -1 (air)."""
    _, d_thick, h, w = thick.shape
    d_thin = thin.shape[1]
    ph, pw = patch_hw
    if h < ph or w < pw:
        raise ValueError(f"volume ({h}, {w}) smaller than the patch ({ph}, {pw})")
    z1 = min(z_thin + depth_thin, d_thin)
    thin_p = thin[:, z_thin:z1, y0:y0 + ph, x0:x0 + pw]
    k0 = max(0, int(z_thin * d_thick / d_thin))
    k1 = min(d_thick, max(int(z1 * d_thick / d_thin), k0 + 1))
    sub = thick[:, k0:k1, y0:y0 + ph, x0:x0 + pw]
    thick_p = F.interpolate(sub.unsqueeze(0), size=(depth_thick, ph, pw), mode="trilinear", align_corners=False).squeeze(0)
    if thin_p.shape[1] < depth_thin:
        thin_p = F.pad(thin_p, (0, 0, 0, 0, 0, depth_thin - thin_p.shape[1]), value=-1.0)
    return thick_p, thin_p


def make_item(thick_patch: torch.Tensor, thin_patch: torch.Tensor, category="unknown", patient_id="") -> Dict[str, object]:
    return {"x_lr": thick_patch, "x_hr": thin_patch, "input": thick_patch, "target": thin_patch, "category": category,
            "patient_id": patient_id}


def unpack_batch(batch: Dict[str, object], device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(v_in, v_gt) of a collated batch, as the reference's trainer takes them (`batch['input']`, `batch['target']`;
    the 'x_lr' / 'x_hr' and 'thick' / 'thin' spellings are accepted too), checked against the engine's contract:
    fp32 (B, 1, D, H, W), H and W multiples of 4 (two stride-2 VAE stages), same B, H, W for both."""
    v_in = batch.get("input", batch.get("x_lr", batch.get("thick")))
    v_gt = batch.get("target", batch.get("x_hr", batch.get("thin")))
    if v_in is None or v_gt is None:
        raise ValueError(f"batch needs 'input' and 'target' (or x_lr/x_hr, thick/thin); keys: {sorted(batch)}")
    for name, v in (("input", v_in), ("target", v_gt)):
        if not torch.is_tensor(v) or v.dim() != 5 or v.shape[1] != 1:
            raise ValueError(f"batch['{name}'] must be a (B, 1, D, H, W) tensor, got {tuple(getattr(v, 'shape', ()))}")
        if v.shape[3] % 4 or v.shape[4] % 4:
            raise ValueError(f"batch['{name}']: H, W = {tuple(v.shape[3:])} must be multiples of 4")
    if v_in.shape[0] != v_gt.shape[0] or tuple(v_in.shape[3:]) != tuple(v_gt.shape[3:]):
        raise ValueError(f"input {tuple(v_in.shape)} and target {tuple(v_gt.shape)} disagree in batch or slice size")
    v_in, v_gt = v_in.float(), v_gt.float()
    if device is not None:
        v_in, v_gt = v_in.to(device, non_blocking=True), v_gt.to(device, non_blocking=True)
    return v_in, v_gt
