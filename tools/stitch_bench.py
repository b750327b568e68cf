"""Sliding-window inference throughput (reference use case: (8,192,192) thick patches -> (48,192,192), stride (4,96,96))
on a 8 x 512 x 512 thick volume = 25 windows, DDIM-10, one window at a time vs batched windows."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("video-to-video-diffusion_amd")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.VideoToVideoDiffusion(bench.EFFECTIVE_CFG).eval().to(dev)
sampler = pkg.DDIMSampler(model.diffusion, model.unet)
v = (torch.rand(1, 1, 8, 512, 512) * 2 - 1).to(dev)
for wb in [int(x) for x in (sys.argv[1:] or ["1", "5", "8", "13", "0"])]:      # 0 = automatic (all windows that fit)
    for rep in range(2):
        torch.manual_seed(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sampler.sample_with_stitching(v, model.vae, 10, patch_size=(8, 192, 192), target_patch_size=(48, 192, 192),
                                            stride=(4, 96, 96), device=dev, progress=False, window_batch=wb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"window_batch={wb}: {dt:.3f} s for 25 windows ({dt / 25 * 1e3:.1f} ms/window), out {tuple(out.shape)}, finite {bool(torch.isfinite(out).all())}")
