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
from . import parallel  # noqa: F401,E402


def enable_depth_sharding(model, group=None):
    """Shard every following sampling run depth-wise over the ranks of `group` (torch.distributed must be
    initialised, one process per GPU; backend "nccl" is RCCL on ROCm).  `model` is a
    VideoToVideoDiffusion (U-Net loop and VAE decode are sharded), a UNet3D or a VideoVAE."""
    import torch.distributed as dist
    # GPU ranks: RCCL issued by libctsi on the engine stream (C ABI, capture-safe); anything else (the gloo CPU tests of
    # the host logic): the same sync points over torch.distributed
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        comm = parallel.RcclComm.from_process_group(group)
    else:
        comm = parallel.DistComm(group)
    for m in (getattr(model, "unet", None), getattr(model, "vae", None), model):
        if m is not None and type(m).__name__ in ("UNet3D", "SliceInterpolationVAE"):
            m.depth_shard_comm = comm
    return comm
