"""Micro-benchmark of ctsi_wgrad on the config-3 / config-2 layer shapes (HIP events, TFLOP/s).
usage: python tools/wgrad_bench.py [--reps 5]      env CTSI_WGRAD_GATHER=1 forces the one-tap-per-block kernel"""
import argparse, ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
L = importlib.import_module("video-to-video-diffusion_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--zero", action="store_true", help="all-zero operands: cycles at the unthrottled clock (DVFS give-back check)")
a = ap.parse_args()
ctx = E.Ctx.get(torch.device("cuda", 0))
lib = ctx.lib
SHAPES = [("L0 128->128 n4 48^3", 128, 128, (4, 48, 48, 48), 3), ("L1 256->256 n4 48x24^2", 256, 256, (4, 48, 24, 24), 3),
          ("L2 512->512 n4 48x12^2", 512, 512, (4, 48, 12, 12), 3), ("L3 512->512 n4 48x6^2", 512, 512, (4, 48, 6, 6), 3),
          ("L0 384->128 (128 src) n4", 128, 128, (4, 48, 48, 48), 3), ("L1 1x1 256->256", 256, 256, (4, 48, 24, 24), 1),
          ("cfg2 L0 128->128 n1 48x128^2", 128, 128, (1, 48, 128, 128), 3)]
for name, cg, cr, (n, d, h, w), k in SHAPES:
    vox = n * d * h * w
    r = torch.randn(vox * cr, device="cuda").to(torch.bfloat16)
    g = torch.randn(vox * cg, device="cuda").to(torch.bfloat16)
    if a.zero:
        r.zero_(); g.zero_()
    p = 1 if k == 3 else 0
    desc = L.WgradDesc(k, k, k, 1, 1, p, p, p, n, d, h, w, d, h, w, cr, cr, cg, cg)
    wsb = lib.wgrad_workspace_bytes(C.byref(desc))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    T = k ** 3
    dw = torch.empty(cr * cg * T, device="cuda")
    fl = lib.wgrad_flops(C.byref(desc))
    with ctx.scope():
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
        lib.wgrad(C.byref(desc), E._ptr(r), E._ptr(g), E._ptr(ws), ws.numel() * ws.element_size(), E._ptr(dw), cg * T, T, 1, 1.0, ctx.sptr)
        lib.event_record(e0, ctx.sptr)
        for _ in range(a.reps):
            lib.wgrad(C.byref(desc), E._ptr(r), E._ptr(g), E._ptr(ws), ws.numel() * ws.element_size(), E._ptr(dw), cg * T, T, 1, 1.0, ctx.sptr)
        lib.event_record(e1, ctx.sptr)
    torch.cuda.synchronize()
    ms = C.c_float()
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    ms = ms.value / a.reps
    print(f"{name:34s} {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s   workspace {wsb / 2**20:7.1f} MiB")
