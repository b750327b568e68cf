"""Drop-in for the reference's `models` package (reference models/__init__.py:1-6)."""
import importlib as _il

_pkg = _il.import_module("video-to-video-diffusion_amd")
VideoVAE = _pkg.VideoVAE
UNet3D = _pkg.UNet3D
GaussianDiffusion = _pkg.GaussianDiffusion
VideoToVideoDiffusion = _pkg.VideoToVideoDiffusion

__all__ = ['VideoVAE', 'UNet3D', 'GaussianDiffusion', 'VideoToVideoDiffusion']
