"""Micro-benchmark of single conv layers (HIP events), optionally with the timing-only K-loop ablation.
    python tools/conv_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
import ctypes as C

dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
CASES = [  # name, cin, cout, (d,h,w), kind
    ("L0 128->128", 128, 128, (48, 128, 128), "k3"),
    ("L1 256->256", 256, 256, (48, 64, 64), "k3"),
    ("L2 512->512", 512, 512, (48, 32, 32), "k3"),
    ("L3 512->512", 512, 512, (48, 16, 16), "k3"),
    ("L0 384->128", 384, 128, (48, 128, 128), "k3"),
]
stats = os.environ.get("CONV_STATS", "1") == "1"
for name, cin, cout, (d, h, w), kind in CASES:
    with ctx.scope():
        prog = E.Program(ctx)
        x = prog.act(1, cin, d, h, w)
        x.t.normal_()
        if os.environ.get("CONV_ZERO") == "1":     # clock-vs-data check (MI355X_MICROARCH.md, DVFS give-back item 1)
            x.t.zero_()
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * (0.0 if os.environ.get("CONV_ZERO") == "1" else 0.02)
        b = torch.randn(cout, device=dev)
        prog.zero_gn_op()
        y, st = prog.conv(name, lambda: wt, lambda: b, x, None, cout=cout, want_stats=stats)
        prog.finalize_layout()
        for _ in range(3):
            prog.run()
        prof = prog.profile_ops(repeats=10)
    ms = [p for p in prof if p[2] > 0][0][3]
    fl = prog.conv_flops[0][1]
    kern = [p for p in prof if p[2] > 0][0][1]
    print(f"{name:14s} {fl/1e9:8.1f} GF  {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  {kern}")
