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
