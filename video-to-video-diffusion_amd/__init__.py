"""MI355X-native engine for the DDPM/DDIM sampling path of Kkuntal990/video-to-video-diffusion.

Host-side mirror of the reference interface (same class names, signatures and state-dict layout)
over libctsi.so, the hand-written HIP (gfx950) kernel library declared in include/ctsi.h.
The directory name contains hyphens; import it with importlib or through the top-level
`models` / `inference` packages, which re-export the reference's import surface.
"""
from .lib import CtsiError, build, get_lib  # noqa: F401
from .vae import SliceInterpolationVAE, VideoVAE  # noqa: F401
from .unet3d import UNet3D  # noqa: F401
from .diffusion import GaussianDiffusion  # noqa: F401
from .model import VideoToVideoDiffusion  # noqa: F401
from .sampler import DDIMSampler, DDPMSampler, EDMSampler  # noqa: F401
from .generate import generate_batch, interpolate_videos  # noqa: F401
from .optim import FusedAdam, FusedAdamW  # noqa: F401
from . import parallel  # noqa: F401,E402


def enable_depth_sharding(model, group=None, transport=None):
    """Shard every following sampling run depth-wise over the ranks of `group` (torch.distributed must be
    initialised, one process per GPU; backend "nccl" is RCCL on ROCm).  `model` is a
    VideoToVideoDiffusion (U-Net loop and VAE decode are sharded), a UNet3D or a VideoVAE."""
    import os
    import torch.distributed as dist
    # Transport of the sync points.  Default: torch.distributed's own collectives (`DistComm`; backend "nccl" IS RCCL on
    # ROCm) -- the path every multi-process test of this repo runs.  `transport="rccl"` (or CTSI_SHARD_TRANSPORT=rccl)
    # selects the C-ABI transport (csrc/comm.hip: one ncclGroup per sync point on the engine stream, capture-safe); it has
    # run with ONE rank on hardware and, rank by rank, against the recording stub (tests/test_rccl_stub.py), but no
    # multi-GPU node was available to the build: `bench.py --mode shard` checks it against the unsharded engine on the
    # first multi-GPU run (`parity` in its JSON) before timing with it.
    transport = (transport or os.environ.get("CTSI_SHARD_TRANSPORT", "dist")).lower()
    if transport not in ("dist", "rccl"):
        raise CtsiError(f"unknown depth-sharding transport '{transport}' (dist | rccl)")
    if transport == "rccl" and dist.is_initialized() and dist.get_backend(group) == "nccl":
        comm = parallel.RcclComm.from_process_group(group)
    else:
        comm = parallel.DistComm(group)
    for m in (getattr(model, "unet", None), getattr(model, "vae", None), model):
        if m is not None and type(m).__name__ in ("UNet3D", "SliceInterpolationVAE"):
            m.depth_shard_comm = comm
    return comm
