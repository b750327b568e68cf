import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
E = importlib.import_module("video-to-video-diffusion_amd.engine")
dev = torch.device("cuda", 0); ctx = E.Ctx.get(dev)
for (cin, cout, dhw) in [(512, 512, (48, 32, 32)), (256, 256, (48, 64, 64)), (512, 512, (48, 16, 16)), (1024, 512, (48, 32, 32))]:
    d, h, w = dhw
    with ctx.scope():
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02; b = torch.randn(cout, device=dev)
        prog = E.Program(ctx); x = prog.act(1, cin, d, h, w); x.t.normal_(); prog.zero_gn_op()
        prog.conv("c", lambda: wt, lambda: b, x, None, cout=cout, want_stats=True); prog.finalize_layout()
        for _ in range(3): prog.run()
        res = {"0": [], "1": []}
        for r in range(6):
            for v in ("0", "1"):
                os.environ["CTSI_CONV_NMAJOR"] = v
                prof = prog.profile_ops(repeats=10); ms = [p for p in prof if p[2] > 0][0]
                res[v].append(ms[2] / ms[3] / 1e9)
    kern = [m for m in prog.op_meta if m[1] > 0][0][2]
    print(cin, cout, dhw, kern, "n_major=0:", " ".join("%5.0f" % v for v in res["0"]), "| n_major=1:", " ".join("%5.0f" % v for v in res["1"]))
