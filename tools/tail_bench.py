"""The six fused residual tails of a config-2 U-Net evaluation (1x1x1 residual conv + GroupNorm of c2 + SiLU, models/unet3d.py:
102, 112-133) on their real shapes: streaming kernel (csrc/conv1_stream.hip) against the gather kernel's fused tail, HIP-event
time per launch and algorithmic bytes / time, both programs alternated in one process.
    python tools/tail_bench.py [--variants] [--nt N]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (c1, c2, cout, d, h, w): down L1, down L2, up L3, up L2, up L1, up L0 of the effective 264.66 M-param U-Net @ (48,128,128)
SHAPES = [(128, 0, 256, 48, 64, 64), (256, 0, 512, 48, 32, 32), (512, 512, 512, 48, 16, 16), (512, 512, 512, 48, 32, 32),
          (512, 256, 256, 48, 64, 64), (256, 128, 128, 48, 128, 128)]


def build(E, ctx, shape, stream):
    c1, c2, cout, d, h, w = shape
    dev = ctx.device
    os.environ["CTSI_CONV1_STREAM"] = "1" if stream else "0"
    prog = E.Program(ctx)
    x1 = prog.act(1, c1, d, h, w)
    x1.t.normal_()
    x2 = None
    if c2:
        x2 = prog.act(1, c2, d, h, w)
        x2.t.normal_()
    hh = prog.act(1, cout, d, h, w)
    hh.t.normal_()
    wt = torch.randn(cout, c1 + c2, 1, 1, 1, device=dev) * (1.0 / (c1 + c2) ** 0.5)
    b = torch.randn(cout, device=dev) * 0.1
    gnm = torch.nn.GroupNorm(32, cout).to(dev)
    prog.zero_gn_op()
    slot = prog.gn_finalize(hh, 32, prog.gn_colsum(hh))
    prog.conv("res1x1+gn", lambda: wt, lambda: b, x1, x2, k=(1, 1, 1), p=(0, 0, 0), cout=cout, out=hh,
              fuse_gn=(hh, slot, gnm, True))
    prog.finalize_layout()
    return prog


VARIANTS = [("stream", {}), ("stream-b512", {"CTSI_CONV1_STREAM_BLOCKS": "512"}), ("stream-b128", {"CTSI_CONV1_STREAM_BLOCKS": "128"})]
KEYS = ["CTSI_CONV1_STREAM_BLOCKS"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--variants", action="store_true", help="time the launch-time variants of the streaming kernel too")
    a = ap.parse_args()
    if a.nt:
        os.environ["CTSI_CONV1_STREAM_NT"] = str(a.nt)
    E = importlib.import_module("video-to-video-diffusion_amd.engine")
    ctx = E.Ctx.get(torch.device("cuda", 0))
    variants = VARIANTS if a.variants else VARIANTS[:1]
    tot = [0.0] * (1 + len(variants))
    with ctx.scope():
        for shape in SHAPES:
            c1, c2, cout, d, h, w = shape
            vox = d * h * w
            nbytes = 2.0 * vox * (c1 + c2 + 2 * cout)
            progs = [build(E, ctx, shape, s) for s in (False, True)]
            runs = [(progs[0], "gather", {})] + [(progs[1], nm, env) for nm, env in variants]
            ms = [[] for _ in runs]
            for rep in range(3):
                for i, (pr, nm, env) in enumerate(runs):
                    for k in KEYS:
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    for _ in range(2):
                        pr.run()
                    prof = pr.profile_ops(repeats=20)
                    ms[i].append([r for r in prof if r[0] == "res1x1+gn"][0][3])
            best = [min(m) for m in ms]
            kern = [[m[2] for m in pr.op_meta if m[0] == "res1x1+gn"][0] for pr in progs]
            for i, b in enumerate(best):
                tot[i] += b
            print(f"{c1 + c2:5d} -> {cout:4d} @{d}x{h}x{w}: {kern[0]} | {kern[1]}: " +
                  "  ".join(f"{nm} {b * 1e3:6.1f} us {nbytes / b / 1e9:4.2f} TB/s" for (_, nm, _), b in zip(runs, best)), flush=True)
    print("sum of the six tails (ms): " + "  ".join(f"{nm} {t:.3f}" for (_, nm, _), t in zip(runs, tot)))


if __name__ == "__main__":
    main()
