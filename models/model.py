"""Drop-in for the reference's `models.model` module: re-exports the HIP-engine mirror."""
import importlib as _il

_mod = _il.import_module("video-to-video-diffusion_amd.model")
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
