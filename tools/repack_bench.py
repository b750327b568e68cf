"""Where the fast re-pack of the config-3 training program goes: HIP events around every launch of engine.Program.fast_repack's
recorded list, grouped by entry point and by image size.    python tools/repack_bench.py"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("video-to-video-diffusion_amd")
E = importlib.import_module("video-to-video-diffusion_amd.engine")
T = importlib.import_module("video-to-video-diffusion_amd.train_engine")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pkg.VideoToVideoDiffusion(bench.EFFECTIVE_CFG).to(dev)
ctx = E.Ctx.get(dev)
os.environ["CTSI_NO_REPACK_GRAPH"] = "1"
with ctx.scope():
    prog = T.UNetTrainProgram(ctx, model.unet, 4, 48, 48, 48)
    prog.fast_repack()
    f = prog._fast
    lib, sptr = ctx.lib, ctx.sptr
    evs = []
    for _ in range(len(f["packs"]) + 2):
        e = C.c_void_p()
        lib.event_create(C.byref(e))
        evs.append(e)
    for rep in range(2):
        lib.event_record(evs[0], sptr)
        lib.copy_scale_multi(E._ptr(f["segs"]), E._ptr(f["pieces"]), f["npieces"], sptr)
        lib.event_record(evs[1], sptr)
        for i, (fn, args) in enumerate(f["packs"]):
            fn(*args)
            lib.event_record(evs[i + 2], sptr)
torch.cuda.synchronize()
ms = C.c_float()
lib.event_elapsed_ms(evs[0], evs[1], C.byref(ms))
print(f"copy_scale_multi: {f['nseg']} segments, {f['npieces']} pieces, {ms.value * 1e3:.1f} us")
agg = {}
for i, (fn, args) in enumerate(f["packs"]):
    lib.event_elapsed_ms(evs[i + 1], evs[i + 2], C.byref(ms))
    a = agg.setdefault(fn.__name__, [0, 0.0])
    a[0] += 1
    a[1] += ms.value
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} n={v[0]:4d} {v[1]:8.3f} ms")
print(f"total pack launches {len(f['packs'])}: {tot:.3f} ms (eager, event-timed; slow-path entries: {len(f['slow'])})")
