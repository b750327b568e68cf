"""Per-op timing table of one U-Net evaluation (HIP events around every launch).
    python tools/profile_ops.py [--hw 512] [--net unet|dec|enc]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=512)
ap.add_argument("--depth", type=int, default=48)
ap.add_argument("--net", default="unet")
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--enc-depth", type=int, default=8, help="--net enc: input slices (8 = thick volume, 48 = training target)")
args = ap.parse_args()
pkg = importlib.import_module("video-to-video-diffusion_amd")
E = importlib.import_module("video-to-video-diffusion_amd.engine")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.VideoToVideoDiffusion(bench.EFFECTIVE_CFG).eval().to(dev)
ctx = E.Ctx.get(dev)
with ctx.scope():
    if args.net == "unet":
        nb = args.batch
        prog = E.UNetProgram(ctx, model.unet, nb, args.depth, args.hw // 4, args.hw // 4, max_rows=4 * nb)
        prog.load_latents(torch.randn(nb, 8, args.depth, args.hw // 4, args.hw // 4, device=dev),
                          torch.randn(nb, 8, args.depth, args.hw // 4, args.hw // 4, device=dev))
        prog.set_schedule([500] * nb)
    elif args.net == "dec":       # (random data, not the zero-initialised input buffer: the conv kernels are power-bound and
        prog = E.VAEDecodeProgram(ctx, model.vae, 1, args.depth, args.hw // 4, args.hw // 4)   # run ~10 % faster on zeros)
        prog.load(torch.randn(1, 8, args.depth, args.hw // 4, args.hw // 4, device=dev))
    else:
        prog = E.VAEEncodeProgram(ctx, model.vae, args.batch, args.enc_depth, args.hw, args.hw)
        prog(torch.rand(args.batch, 1, args.enc_depth, args.hw, args.hw, device=dev) * 2 - 1)
    prog.run()
    prof = prog.profile_ops(repeats=args.repeats)
torch.cuda.synchronize()
tot = sum(p[3] for p in prof)
print(f"{'op':24s} {'kernel':28s} {'GFLOP':>9s} {'ms':>8s} {'TFLOP/s':>8s}")
agg = {}
for name, kern, fl, ms in prof:
    if fl > 0:
        print(f"{name:24s} {kern:28s} {fl / 1e9:9.1f} {ms:8.3f} {fl / ms / 1e9:8.1f}")
    a = agg.setdefault(name if fl == 0 else kern, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += ms
    a[2] += fl
print("---- aggregate ----")
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:30s} n={n:4d} {ms:9.3f} ms {100 * ms / tot:5.1f}%  {fl / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s")
print(f"total {tot:.3f} ms, {sum(p[2] for p in prof) / 1e12:.2f} TFLOP")
