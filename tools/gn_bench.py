"""gn_apply micro-benchmark: GB/s on the L0 / L1 tensors of the config-2 step, in place (GN_OUT_OF_PLACE=1: into a
second buffer), with and without residual.  Round-2 finding: 4.5-5.0 TB/s (1 read + 1 write) / 5.3-5.8 TB/s (2 reads + 1
write) whatever the grid size, loads in flight, store policy or buffer aliasing: the HBM read/write mix bounds it."""
import importlib, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
for (c, d, h, w) in ((128, 48, 128, 128), (256, 48, 64, 64), (128, 48, 512, 512), (256, 48, 256, 256)):
    for res in (False, True):
        with ctx.scope():
            prog = E.Program(ctx)
            x = prog.act(1, c, d, h, w); x.t.normal_()
            r = prog.act(1, c, d, h, w); r.t.normal_()
            gn = torch.nn.GroupNorm(32, c).to(dev)
            st = prog.gn_colsum(x)
            slot = prog.gn_finalize(x, 32, st)
            y = prog.act(1, c, d, h, w)
            inplace = os.environ.get("GN_OUT_OF_PLACE") is None
            for _ in range(6):
                prog.gn_apply(x, slot, gn, silu_pre=not res, residual=r if res else None, silu_post=res, out=x if inplace else y)
            prog.finalize_layout()
            prog.run()
            prof = prog.profile_ops(repeats=5)
        ms = [p[3] for p in prof if p[0] == "gn.apply"]
        nb = (3 if res else 2) * 2.0 * c * d * h * w
        print(f"c={c} {d}x{h}x{w} residual={res}: {sum(ms)/len(ms)*1e3:7.1f} us  {nb/(sum(ms)/len(ms)*1e-3)/1e9:7.0f} GB/s")
