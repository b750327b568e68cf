"""Helpers for the `-m gpu` tests: drive single libctsi ops through engine.Program."""
import ctypes as C
import importlib

import torch

E = importlib.import_module("video-to-video-diffusion_amd.engine")
_ptr = E._ptr


def ctx():
    return E.Ctx.get(torch.device("cuda", 0))


def to_act(prog, x_ncdhw, c_pad=None):
    """fp32 NCDHW (any device) -> bf16 NDHWC Act on the engine stream."""
    n, c, d, h, w = x_ncdhw.shape
    ct = c if c_pad is None else c_pad
    buf = prog.persistent((n * d * h * w * ct,), torch.bfloat16, zero=True)
    xx = x_ncdhw.to(prog.ctx.device, torch.float32).contiguous()
    prog.lib.ncdhw_f32_to_ndhwc_bf16(_ptr(xx), _ptr(buf), n, c, d, h, w, ct, 0, prog.ctx.sptr)
    prog.keep.append(xx)
    return E.Act(buf, n, ct, d, h, w)


def from_act(prog, a):
    out = torch.empty((a.n, a.c, a.d, a.h, a.w), dtype=torch.float32, device=prog.ctx.device)
    prog.lib.ndhwc_bf16_to_ncdhw_f32(_ptr(a.t), _ptr(out), a.n, a.c, a.d, a.h, a.w, prog.ctx.sptr)
    return out


def run_conv(x1, x2, weight, bias, *, transposed=False, k=(3, 3, 3), s=(1, 1), p=(1, 1, 1), f32=False, act=0,
             c1_pad=None, cin_w=None, want_stats=False, groups=None):
    """Returns (y fp32 NCDHW on cpu, sums or None)."""
    c = ctx()
    with c.scope():
        prog = E.Program(c)
        a1 = to_act(prog, x1, c1_pad)
        a2 = to_act(prog, x2) if x2 is not None else None
        cout = weight.shape[1] if transposed else weight.shape[0]
        kw = dict(transposed=transposed, k=k, s=s, p=p, cout=cout, cin_w=cin_w, want_stats=want_stats, act=act)
        slot = None
        prog.zero_gn_op()
        if f32:
            # probe output dims with a throw-away plan
            desc = E.ConvDesc(int(transposed), k[0], k[1], k[2], s[0], s[1], p[0], p[1], p[2], a1.n, a1.c,
                              0 if a2 is None else a2.c, cout, a1.d, a1.h, a1.w, 0)
            plan = C.c_void_p()
            prog.lib.conv_plan_create(C.byref(plan), C.byref(desc))
            do, ho, wo = C.c_int(), C.c_int(), C.c_int()
            prog.lib.conv_plan_out_dims(plan, C.byref(do), C.byref(ho), C.byref(wo))
            prog.lib.conv_plan_destroy(plan)
            do, ho, wo = do.value, ho.value, wo.value
            y = prog.persistent((a1.n, cout, do, ho, wo), torch.float32, zero=True)
            vox = do * ho * wo
            prog.conv("t", lambda: weight, (lambda: bias) if bias is not None else None, a1, a2, f32_out=y,
                      f32_strides=(cout * vox, vox, ho * wo, wo, 1), **kw)
            out_act = None
        else:
            out_act, st = prog.conv("t", lambda: weight, (lambda: bias) if bias is not None else None, a1, a2, **kw)
            if want_stats:
                slot = prog.gn_finalize(out_act, groups, st)
        prog.finalize_layout()
        prog.run()
        res = y.clone() if f32 else from_act(prog, out_act)
        sums = None
        if slot is not None:
            sums = prog._gn_sums[slot:slot + out_act.n * groups * 2].clone().reshape(out_act.n, groups, 2)
    torch.cuda.synchronize()
    return res.cpu(), (sums.cpu() if sums is not None else None)
