"""Interleaved A/B timing of one conv layer under several plan-selection environments (variants alternate inside one
process: separate processes drift by several percent with clock / thermal state).
    python tools/ab_variants.py --cin 512 --cout 512 --dhw 48 16 16 CTSI_CONV_H32W16=1 CTSI_CONV_H32W16=2
    python tools/ab_variants.py --cin 128 --cout 8 --f32 --dhw 48 128 128 CTSI_CONV_NO_HEAD3=1 -
(a variant is a comma-separated list of NAME=value settings; "-" = no setting)"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--cin", type=int, default=512); ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--dhw", type=int, nargs=3, default=[48, 32, 32]); ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--n", type=int, default=1); ap.add_argument("--f32", action="store_true")
ap.add_argument("--transposed", action="store_true", help="ConvTranspose3d (3,4,4) / (1,2,2); --dhw is the INPUT grid")
ap.add_argument("--zero", action="store_true", help="all-zero operands: cycles at the unthrottled clock (MI355X_MICROARCH.md, DVFS give-back)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
d, h, w = a.dhw
progs = {}
names = set()
for v in a.variants:
    for kv in v.split(","):
        if "=" in kv:
            names.add(kv.split("=")[0])
with ctx.scope():
    wt = torch.randn(a.cout, a.cin, 3, 3, 3, device=dev) * (0.0 if a.zero else 0.02)
    if a.transposed:
        wt = torch.randn(a.cin, a.cout, 3, 4, 4, device=dev) * (0.0 if a.zero else 0.02)
    b = torch.randn(a.cout, device=dev)
    for v in a.variants:
        for nme in names:
            os.environ.pop(nme, None)
        for kv in v.split(","):
            if "=" in kv:
                k_, val = kv.split("=")
                os.environ[k_] = val
        prog = E.Program(ctx)
        x = prog.act(a.n, a.cin, d, h, w)
        x.t.normal_()
        if a.zero:
            x.t.zero_()
        prog.zero_gn_op()
        if a.f32:
            y = prog.persistent((a.n, d, h, w, a.cout), torch.float32)
            vox = d * h * w
            prog.conv("c", lambda: wt, lambda: b, x, None, cout=a.cout, f32_out=y,
                      f32_strides=(vox * a.cout, 1, h * w * a.cout, w * a.cout, a.cout))
        elif a.transposed:
            prog.conv("c", lambda: wt, lambda: b, x, None, cout=a.cout, want_stats=True, transposed=True, k=(3, 4, 4), s=(2, 2))
        else:
            prog.conv("c", lambda: wt, lambda: b, x, None, cout=a.cout, want_stats=True)
        prog.finalize_layout()
        for _ in range(3):
            prog.run()
        progs[v] = prog
    res = {v: [] for v in progs}
    for r in range(a.rounds):
        for v, prog in progs.items():
            # the per-launch environment switches (stagger, n-major, ring) are read at launch time
            for nme in names:
                os.environ.pop(nme, None)
            for kv in v.split(","):
                if "=" in kv:
                    k_, val = kv.split("=")
                    os.environ[k_] = val
            prof = prog.profile_ops(repeats=10)
            ms = [p for p in prof if p[2] > 0][0]
            res[v].append((ms[2] / ms[3] / 1e9, ms[3]))
for v, prog in progs.items():
    kern = [m for m in prog.op_meta if m[1] > 0][0][2]
    xs = [t for t, _ in res[v]]
    ms = sum(m for _, m in res[v]) / len(res[v])
    print(f"{v:34s} {kern:22s} TFLOP/s per round: {' '.join('%6.0f' % t for t in xs)}   mean {sum(xs) / len(xs):7.1f}  {ms:.4f} ms")
