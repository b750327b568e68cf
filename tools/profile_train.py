"""Per-op-name time breakdown of one config-3 training micro-step (U-Net forward + backward), HIP events per launch.
usage: python tools/profile_train.py [--batch 4] [--hw 48] [--depth 48]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--hw", type=int, default=48)
ap.add_argument("--depth", type=int, default=48)
ap.add_argument("--repeats", type=int, default=2)
ap.add_argument("--by-flops", action="store_true", help="split the conv rows by the layer's GFLOP as well (tells the channel counts apart)")
a = ap.parse_args()
pkg = importlib.import_module("video-to-video-diffusion_amd")
E = importlib.import_module("video-to-video-diffusion_amd.engine")
T = importlib.import_module("video-to-video-diffusion_amd.train_engine")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.VideoToVideoDiffusion(bench.EFFECTIVE_CFG).to(dev)
ctx = E.Ctx.get(dev)
shape = (a.batch, 8, a.depth, a.hw, a.hw)
with ctx.scope():
    prog = T.UNetTrainProgram(ctx, model.unet, a.batch, a.depth, a.hw, a.hw)
    prog.set_diffusion(model.diffusion)
    z0, cond, noise = (torch.randn(shape, device=dev) for _ in range(3))
    t = torch.randint(0, 1000, (a.batch,), device=dev)
    norm = torch.full((a.batch,), 1.0 / z0[0].numel() / a.batch, device=dev)
    prog.run_forward(z0, cond, t, noise, norm)
    prog.run_backward(torch.ones(1, device=dev))
    prof = prog.profile_ops(repeats=a.repeats)
agg = {}
for i, (name, kern, fl, ms) in enumerate(prof):
    phase = "F" if i < prog.n_fwd else "B"
    key = (phase, name if not kern.startswith("conv_mfma") else name.split(".")[-1] + ":" + kern)
    if a.by_flops and (kern.startswith("conv_mfma") or "wgrad" in name):
        key = (phase, key[1] + f" {fl / 1e9:.0f}G")
    g = agg.setdefault(key, [0, 0.0, 0.0])
    g[0] += 1; g[1] += fl; g[2] += ms
tot = sum(v[2] for v in agg.values())
print(f"total {tot:.2f} ms, {prog.flops / 1e12:.2f} TFLOP, pool {prog.pool.total_bytes / 2**30:.2f} GiB")
for (phase, name), v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:60 if a.by_flops else 40]:
    tf = f"{v[1] / (v[2] * 1e-3) / 1e12:7.0f} TF" if v[1] else "          "
    print(f"{phase} {name:42s} n={v[0]:4d} {v[2]:8.3f} ms {100 * v[2] / tot:5.1f}% {tf}")
