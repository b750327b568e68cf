"""Interleaved A/B timing of one conv layer under two plan-selection environments (clock / thermal drift between
separate processes is several percent on this part, so variants are alternated inside one process).
    python tools/conv_ab.py CTSI_CONV_M512 0 1 --cin 512 --cout 512 --dhw 48 32 32
(tools/ab_variants.py is the general form: any number of variants, several variables each)"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
ap = argparse.ArgumentParser()
ap.add_argument("var"); ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--cin", type=int, default=512); ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--dhw", type=int, nargs=3, default=[48, 32, 32]); ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--n", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
d, h, w = a.dhw
progs = {}
with ctx.scope():
    wt = torch.randn(a.cout, a.cin, 3, 3, 3, device=dev) * 0.02
    b = torch.randn(a.cout, device=dev)
    for val in (a.a, a.b):
        os.environ[a.var] = val
        prog = E.Program(ctx)
        x = prog.act(a.n, a.cin, d, h, w)
        x.t.normal_()
        prog.zero_gn_op()
        prog.conv("c", lambda: wt, lambda: b, x, None, cout=a.cout, want_stats=True)
        prog.finalize_layout()
        for _ in range(3):
            prog.run()
        progs[val] = prog
    res = {v: [] for v in progs}
    for r in range(a.rounds):
        for val, prog in progs.items():
            prof = prog.profile_ops(repeats=10)
            ms = [p for p in prof if p[2] > 0][0]
            res[val].append(ms[2] / ms[3] / 1e9)
for val, prog in progs.items():
    kern = [m for m in prog.op_meta if m[1] > 0][0][2]
    xs = res[val]
    print(f"{a.var}={val:4s} {kern:24s} TFLOP/s per round: {' '.join('%6.0f' % v for v in xs)}   mean {sum(xs) / len(xs):7.1f}")
