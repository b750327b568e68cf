"""How long the fp32 torch oracle takes on the GPU box for the full-size parity tests (MIOpen fp32 conv3d).
usage: MIOPEN_FIND_MODE=FAST python tools/oracle_timing.py"""
import os, sys, time
import torch
import torch.nn.functional as F

dev = "cuda:0"
def t(fn, name):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize()
    print(f"{name}: {time.time()-t0:.2f} s", flush=True); return r

print("MIOPEN_FIND_MODE =", os.environ.get("MIOPEN_FIND_MODE"), flush=True)
for (cin, cout, d, h, w) in ((128, 128, 48, 128, 128), (512, 512, 48, 32, 32), (128, 128, 48, 512, 512), (256, 256, 48, 256, 256), (8, 128, 48, 128, 128), (128, 1, 48, 512, 512)):
    x = torch.randn(1, cin, d, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
    for rep in range(2):
        y = t(lambda: F.conv3d(x, wt, None, padding=1), f"  conv3d fp32 {cin}->{cout} @{d}x{h}x{w} rep{rep}")
    del x, wt, y
    torch.cuda.empty_cache()
x = torch.randn(1, 256, 48, 256, 256, device=dev)
wt = torch.randn(256, 128, 3, 4, 4, device=dev) * 0.02
for rep in range(2):
    y = t(lambda: F.conv_transpose3d(x, wt, None, stride=(1, 2, 2), padding=(1, 1, 1)), f"  convT fp32 256->128 @48x256x256 rep{rep}")
del x, wt, y
x = torch.randn(1, 128, 48, 512, 512, device=dev)
wt = torch.randn(128, 128, 3, 4, 4, device=dev) * 0.02
for rep in range(2):
    y = t(lambda: F.conv3d(x, wt, None, stride=(1, 2, 2), padding=(1, 1, 1)), f"  strided conv fp32 128->128 @48x512x512 rep{rep}")
del wt, y
for rep in range(2):
    y = t(lambda: F.silu(F.group_norm(x, 32)), f"  gn+silu 128 @48x512x512 rep{rep}")
with torch.autocast("cuda", dtype=torch.bfloat16):
    wt = torch.randn(128, 128, 3, 3, 3, device=dev) * 0.02
    for rep in range(2):
        y = t(lambda: F.conv3d(x, wt, None, padding=1), f"  conv3d bf16-autocast 128->128 @48x512x512 rep{rep}")
