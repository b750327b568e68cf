"""Soak check of the split-K hand-off (conv3_halo_k32_kernel<SK>): many launches of the benchmark's two split-K layer classes
and the split-K Downsample form inside one process; every launch must reproduce the first one bit for bit (the sum of the two halves does not depend on
which block finishes first) and agree with the unsplit kernel to fp32 reassociation.
    python tools/splitk_soak.py [--launches 200]"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
CASES = [("16-wide 512->512", 512, 512, (48, 16, 16), "1", False), ("32-wide 1024->512", 1024, 512, (48, 32, 32), "512", False),
         ("Downsample (3,4,4)/(1,2,2) 512->512 -> 48x16x16", 512, 512, (48, 32, 32), "1", True)]
for name, cin, cout, (d, h, w), mode, down in CASES:
    outs = {}
    for sk in (mode, "0"):
        os.environ["CTSI_CONV_K32_SPLITK"] = sk
        with ctx.scope():
            torch.manual_seed(0)
            prog = E.Program(ctx)
            x = prog.act(1, cin, d, h, w)
            x.t.normal_()
            wt = torch.randn((cout, cin, 3, 4, 4) if down else (cout, cin, 3, 3, 3), device=dev) * 0.02
            b = torch.randn(cout, device=dev)
            prog.zero_gn_op()
            kw = dict(k=(3, 4, 4), s=(2, 2)) if down else {}
            y, st = prog.conv("c", lambda: wt, lambda: b, x, None, cout=cout, want_stats=True, **kw)
            prog.finalize_layout()
            prog.run()
            first = y.t.clone()
            bad = 0
            for _ in range(a.launches if sk != "0" else 1):
                prog.run()
                bad += int(not torch.equal(y.t, first))
            torch.cuda.synchronize()
            kern = [m for m in prog.op_meta if m[1] > 0][0][2]
            outs[sk] = (first.float(), kern, bad)
    os.environ.pop("CTSI_CONV_K32_SPLITK", None)
    ys, y0 = outs[mode][0], outs["0"][0]
    rel = float((ys - y0).norm() / y0.norm())
    print(f"{name}: {outs[mode][1]} x {a.launches} launches, {outs[mode][2]} differ from the first; vs {outs['0'][1]}: rel-L2 {rel:.2e}, "
          f"finite {bool(torch.isfinite(ys).all())}")
    assert outs[mode][2] == 0 and rel < 3e-3 and outs[mode][1].endswith("s")
E.check_device_errors(ctx)          # no split-K consumer's bounded wait expired
print("ok")
