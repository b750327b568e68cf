"""Drop-in for the reference's `inference.sampler` module: re-exports the HIP-engine mirror."""
import importlib as _il

_mod = _il.import_module("video-to-video-diffusion_amd.sampler")
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
