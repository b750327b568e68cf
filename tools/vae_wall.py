"""Where the wall-clock of one vae.decode / vae.encode call goes (load, launches, result copy), eager vs hipGraph.
    python tools/vae_wall.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("video-to-video-diffusion_amd")
E = importlib.import_module("video-to-video-diffusion_amd.engine")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.VideoToVideoDiffusion(bench.EFFECTIVE_CFG).eval().to(dev)
ctx = E.Ctx.get(dev)
z = torch.randn(1, 8, 48, 128, 128, device=dev)
v = torch.rand(1, 1, 8, 512, 512, device=dev) * 2 - 1


def wall(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sum(ts) / len(ts)


for leg, fn, arg in (("decode", model.vae.decode, z), ("encode", model.vae.encode, v)):
    print(leg, "vae.%s() wall ms (min, mean of 5):" % leg, "%.2f %.2f" % wall(lambda: fn(arg)))
    prog = [p for k, p in model.vae.__dict__["_ctsi_programs"].items() if k[0] == leg[:3]][-1]
    with ctx.scope():
        print("   prog.run() eager          :", "%.2f %.2f" % wall(prog.run))
        prof = prog.profile_ops(repeats=2)
        print("   sum of per-op HIP events  : %.2f" % sum(p[3] for p in prof))
        prog.capture()
        print("   prog.launch() hipGraph    :", "%.2f %.2f" % wall(prog.launch))
        print("   ensure_fresh()            :", "%.3f %.3f" % wall(prog.ensure_fresh))
    print("   vae.%s() with the graph   :" % leg, "%.2f %.2f" % wall(lambda: fn(arg)))
