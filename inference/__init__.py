"""Drop-in for the reference's `inference` package (reference inference/__init__.py:1-3)."""
import importlib as _il

_pkg = _il.import_module("video-to-video-diffusion_amd")
DDIMSampler = _pkg.DDIMSampler
DDPMSampler = _pkg.DDPMSampler

__all__ = ['DDIMSampler', 'DDPMSampler']
