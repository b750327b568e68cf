"""Drop-in for the reference's `models` package (reference models/__init__.py:1-6)."""
import importlib as _il

_pkg = _il.import_module("video-to-video-diffusion_amd")
VideoVAE = _pkg.VideoVAE
UNet3D = _pkg.UNet3D
GaussianDiffusion = _pkg.GaussianDiffusion
VideoToVideoDiffusion = _pkg.VideoToVideoDiffusion
# additive (no reference counterpart in this package): the on-device twins of torch.optim.AdamW / Adam, which the
# reference's training/train.py:205-208 constructs
FusedAdamW, FusedAdam = _pkg.FusedAdamW, _pkg.FusedAdam

__all__ = ['VideoVAE', 'UNet3D', 'GaussianDiffusion', 'VideoToVideoDiffusion', 'FusedAdamW', 'FusedAdam']
