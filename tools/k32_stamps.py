"""In-kernel timeline of conv3_halo_k32_kernel (CTSI_DEBUG_FLAGS & 4096: wave 0 of every block records s_memtime at entry, loop
start, loop end, after the epilogue's first barrier, after the LDS staging, after the row stores were issued, at the end, plus
its HW_ID / XCC_ID): where a tile's time goes and how long a CU idles between two blocks.
    python tools/k32_stamps.py --cin 128 --cout 128 --dhw 48 128 128"""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
E = importlib.import_module("video-to-video-diffusion_amd.engine")
L = importlib.import_module("video-to-video-diffusion_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=128); ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--dhw", type=int, nargs=3, default=[48, 128, 128])
ap.add_argument("--no-stats", action="store_true")
ap.add_argument("--flags", type=int, default=0, help="extra CTSI_DEBUG_FLAGS bits (knock-outs)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = E.Ctx.get(dev)
d, h, w = a.dhw
with ctx.scope():
    wt = torch.randn(a.cout, a.cin, 3, 3, 3, device=dev) * 0.02
    b = torch.randn(a.cout, device=dev)
    prog = E.Program(ctx)
    x = prog.act(1, a.cin, d, h, w); x.t.normal_()
    prog.zero_gn_op()
    prog.conv("c", lambda: wt, lambda: b, x, None, cout=a.cout, want_stats=not a.no_stats)
    prog.finalize_layout()
    for _ in range(3):
        prog.run()
    torch.cuda.synchronize()
    os.environ["CTSI_DEBUG_FLAGS"] = str(4096 + a.flags)
    prog.run()
    torch.cuda.synchronize()
    os.environ.pop("CTSI_DEBUG_FLAGS")
nb = 4096
buf = np.zeros((nb, 8), dtype=np.uint64)
lib = L.get_lib()._dll
lib.ctsi_debug_k32_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(nb))
used = buf[:, 0] != 0
s = buf[used].astype(np.int64)
print(f"{used.sum()} blocks recorded")
t = s[:, :7] - s[:, :1]
names = ["entry->loop start (prologue)", "loop", "loop end->barrier", "LDS staging + sums", "row stores issued", "barrier + cross-wave column sums"]
for k in range(6):
    dk = t[:, k + 1] - t[:, k]
    print(f"{names[k]:32s} mean {dk.mean():9.0f}  p10 {np.percentile(dk, 10):9.0f}  p90 {np.percentile(dk, 90):9.0f} ticks")
tot = t[:, 6]
print(f"{'block total':32s} mean {tot.mean():9.0f}")
hw = s[:, 7]
cu = ((hw >> 32) & 0xf) * 1024 + (hw & 0xffffffff & 0xff00) // 256 + ((hw >> 13) & 7) * 16   # (xcc, se, sh|cu): unique-ish key
key = ((hw >> 32) << 16) | (hw & 0xff00)
gaps = []
for k_ in np.unique(key):
    idx = np.where(key == k_)[0]
    o = idx[np.argsort(s[idx, 0])]
    for i0, i1 in zip(o[:-1], o[1:]):
        gaps.append(s[i1, 0] - s[i0, 6])
gaps = np.array(gaps)
print(f"{len(np.unique(key))} distinct (xcc, se, cu) keys; gap between a block's end and the next block's entry on the same CU: "
      f"mean {gaps.mean():.0f}  p10 {np.percentile(gaps, 10):.0f}  p90 {np.percentile(gaps, 90):.0f} ticks")
print(f"whole launch: {s[:, 6].max() - s[:, 0].min()} ticks")
