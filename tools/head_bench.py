"""Few-cout conv heads (csrc/conv3_head2.hip vs csrc/conv3_head.hip) on their real shapes: HIP-event time per launch, algorithmic
bytes / time, variants of the second form side by side (CTSI_HEAD2_VARIANT, CTSI_HEAD2_BLOCKS are read at the first launch, so
every variant runs in its own process).
    python tools/head_bench.py [dec|unet] [variant] [blocks]"""
import importlib
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"dec": (1, 128, 48, 512, 512, 1, 1), "unet": (1, 128, 48, 128, 128, 8, 0), "dec300": (1, 128, 300, 512, 512, 1, 1)}


def one(which):
    E = importlib.import_module("video-to-video-diffusion_amd.engine")
    dev = torch.device("cuda", 0)
    ctx = E.Ctx.get(dev)
    n, c, d, h, w, cout, act = SHAPES[which]
    with ctx.scope():
        prog = E.Program(ctx)
        x = prog.act(n, c, d, h, w)
        x.t.normal_()
        wt = torch.randn(cout, c, 3, 3, 3, device=dev) * 0.03
        b = torch.randn(cout, device=dev) * 0.1
        vox = d * h * w
        y = prog.persistent((n, cout, d, h, w), torch.float32, zero=True)
        strides = (cout * vox, vox, h * w, w, 1) if cout == 1 else (vox * cout, 1, h * w * cout, w * cout, cout)
        prog.conv("head", lambda: wt, lambda: b, x, None, cout=cout, f32_out=y, f32_strides=strides, act=act)
        prog.finalize_layout()
        for _ in range(3):
            prog.run()
        prof = prog.profile_ops(repeats=20)
        ms = prof[0][3]
        nbytes = 2.0 * n * c * vox + 4.0 * n * cout * vox
        print(f"{which:6s} variant {os.environ.get('CTSI_HEAD2_VARIANT', '0'):2s} blocks {os.environ.get('CTSI_HEAD2_BLOCKS', '1024'):5s} "
              f"old-kernel {os.environ.get('CTSI_CONV_NO_HEAD2', '0')}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s  "
              f"checksum {float(y.double().sum()):.6e}")


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        one(sys.argv[1])
    else:
        which = sys.argv[1] if len(sys.argv) > 1 else "all"
        runs = []
        if which in ("dec", "all"):
            runs += [("dec", dict(CTSI_CONV_NO_HEAD2="1"))] + [("dec", dict(CTSI_HEAD2_VARIANT=str(v))) for v in range(2)]
            runs += [("dec", dict(CTSI_HEAD2_VARIANT=str(v), CTSI_HEAD2_BLOCKS=str(b))) for v in (0, 1) for b in (512, 2048, 4096)]
        if which in ("unet", "all"):
            runs += [("unet", dict(CTSI_CONV_NO_HEAD2="1"))] + [("unet", dict(CTSI_HEAD2_VARIANT=str(v))) for v in range(2)]
            runs += [("unet", dict(CTSI_HEAD2_VARIANT=str(v), CTSI_HEAD2_BLOCKS=str(b))) for v in (0, 1) for b in (512, 1024, 2048)]
        for w_, env in runs:
            subprocess.call([sys.executable, os.path.abspath(__file__), w_, "x"], env=dict(os.environ, **env))
