"""Training forward/backward of the U-Net on the HIP engine.

Reference: `GaussianDiffusion.q_sample / training_loss` (models/diffusion.py:81-247) called from
`VideoToVideoDiffusion.forward` (models/model.py:158-228); the backward is what autograd derives from
models/unet3d.py.  Here one `UNetTrainProgram` holds the forward launches of engine.UNetProgram (without
buffer reuse: 288 GB of HBM keep every activation, so no recomputation / gradient checkpointing is needed) plus
the backward launches, built by replaying a tape of the forward layers in reverse:

  conv            dgrad = one of the *forward* conv kernels on re-laid-out weights (3^3 / 1^3: flipped + transposed;
                  strided Conv3d <-> ConvTranspose3d use each other's plans with the weights as they are),
                  wgrad = ctsi_wgrad (MFMA, transposing LDS reads), bias = ctsi_channel_sum
  GroupNorm chain ctsi_gn_bwd
  attention       the depth-sum form of engine.Program.attention, un-folded into its two 1x1x1 convs so that
                  proj_out / qkv(V) get their own gradients (the q and k thirds get exactly zero: rowsum(softmax) == 1)
  time embedding  ctsi_linear_bwd

`train_step` wraps both passes in a torch.autograd.Function over the U-Net parameters, so `loss.backward()`,
GradScaler, gradient accumulation, clip_grad_norm_ and any torch optimizer work as with the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from .engine import Act, Ctx, Program, _ptr
from .lib import CtsiError, WgradDesc


class UNetTrainProgram(Program):
    def __init__(self, ctx: Ctx, unet, n: int, d: int, h: int, w: int):
        super().__init__(ctx)
        self.weight_cache = False    # weights change every optimizer step: private images, repacked in place
        if unet.attention_mode != "fast":
            raise CtsiError("training supports attention_mode='fast' only")
        self.unet = unet
        self.n, self.d, self.h, self.w = n, d, h, w
        L = unet.latent_dim
        self.L = L
        self.Lp = (L + 7) // 8 * 8
        dev = ctx.device
        vox = d * h * w
        self.tape: List[Callable[[], None]] = []
        self._deferred_lin: List[tuple] = []     # small pointwise layers whose weight / bias gradients go into ONE launch at the end
        self.grads: Dict[int, torch.Tensor] = {}     # id(param) -> fp32 gradient buffer (maybe padded)
        self._need = dict(wgrad=16, gn=16, chsum=16)
        self._ws: Dict[str, Optional[torch.Tensor]] = dict(wgrad=None, gn=None, chsum=None)
        self.params = [p for p in unet.parameters()]
        self.track_module(unet)
        # every parameter-gradient buffer is a piece of ONE fp32 arena, so that the hand-over to autograd is one copy instead of
        # one per parameter (333 launches, 1.6 ms of a config-3 micro-step); the slack takes the row padding of the conv layouts
        self._garena = self.persistent((sum(p.numel() for p in self.params) + (16 << 20),), torch.float32, zero=True)
        self._garena_used = 0
        for p in self.params:
            if not p.is_cuda:
                raise CtsiError("training runs on the HIP engine: move the model to a ROCm device first")

        # ---- inputs ------------------------------------------------------------------------------------
        self.xin = Act(self.persistent((n * vox * 2 * L,), torch.bfloat16, zero=True), n, 2 * L, d, h, w, 0)
        self.z0 = self.persistent((n, L, d, h, w), torch.float32, zero=True)
        self.noise = self.persistent((n, L, d, h, w), torch.float32, zero=True)
        self.cond = self.persistent((n, L, d, h, w), torch.float32, zero=True)
        self.t_rows = self.persistent((n,), torch.int32, zero=True)
        self.norm = self.persistent((n,), torch.float32, zero=True)
        self.mask = self.persistent((n, L, d), torch.float32, zero=True)
        self.use_mask = False
        self.gscale = self.persistent((1,), torch.float32, zero=True)
        self.eps = self.persistent((n, d, h, w, L), torch.float32)
        self.loss_out = self.persistent((1 + n,), torch.float32, zero=True)
        self.loss_ws = self.persistent((self.lib.mse_loss_workspace_doubles(n),), torch.float64)
        self.sqrt_ac = self.persistent((1,), torch.float32)      # replaced by set_diffusion()
        self.sqrt_1mac = self.persistent((1,), torch.float32)

        # ---- time embedding --------------------------------------------------------------------------------
        te = unet.time_embed.time_mlp
        self.dim = unet.model_channels
        self.time_dim = te[1].out_features
        self.blocks = [m for m in unet.modules() if type(m).__name__ == "ResBlock3D"]
        self.tb_off = {}
        off = 0
        for m in self.blocks:
            self.tb_off[id(m)] = off
            off += m.time_mlp[1].out_features
        self.total_out = off
        self.w1 = self.dev_f32(lambda: te[1].weight)
        self.b1 = self.dev_f32(lambda: te[1].bias)
        self.w2 = self.dev_f32(lambda: te[3].weight)
        self.b2 = self.dev_f32(lambda: te[3].bias)
        self.w_all = self.dev_f32(lambda: torch.cat([m.time_mlp[1].weight for m in self.blocks], 0),
                                  parts=[(lambda m=m: m.time_mlp[1].weight) for m in self.blocks])
        self.b_all = self.dev_f32(lambda: torch.cat([m.time_mlp[1].bias for m in self.blocks], 0),
                                  parts=[(lambda m=m: m.time_mlp[1].bias) for m in self.blocks])
        self.tbias = self.persistent((n, self.total_out), torch.float32, zero=True)
        self.d_tbias = self.persistent((n, self.total_out), torch.float32, zero=True)
        self.te_scratch = self.persistent((n * (self.dim + 2 * self.time_dim),), torch.float32)
        self.g_w_all = self.grad_alloc((self.total_out, self.time_dim))
        self.g_b_all = self.grad_alloc((self.total_out,))
        self.g_temb = self.persistent((n, self.time_dim), torch.float32, zero=True)
        self.g_lin1 = self.persistent((n, self.time_dim), torch.float32, zero=True)

        lib, sptr = self.lib, ctx.sptr
        prog = self

        def run_inputs():
            lib.q_sample(_ptr(prog.z0), _ptr(prog.noise), _ptr(prog.sqrt_ac), _ptr(prog.sqrt_1mac), _ptr(prog.t_rows),
                         prog.xin.ip, n, L, d, h, w, 2 * L, 0, sptr)
            lib.ncdhw_f32_to_ndhwc_bf16(_ptr(prog.cond), prog.xin.ip, n, L, d, h, w, 2 * L, L, sptr)
            lib.time_embed_train_fwd(_ptr(prog.t_rows), n, prog.dim, prog.time_dim, _ptr(prog.w1), _ptr(prog.b1),
                                     _ptr(prog.w2), _ptr(prog.b2), _ptr(prog.w_all), _ptr(prog.b_all), prog.total_out,
                                     _ptr(prog.te_scratch), _ptr(prog.tbias), sptr)

        self._emit(run_inputs, "train.inputs")
        self.zero_gn_op()

        # ---- forward network (same wiring as engine.UNetProgram) -------------------------------------------------
        x = self.t_conv("conv_in", unet.conv_in, self.xin, None, need_dx=False)
        skips: List[Act] = []
        for level_blocks, down in zip(unet.down_blocks, unet.down_samples):
            for block_list in level_blocks:
                for layer in block_list:
                    x = self._layer(layer, x, None)
            skips.append(x)
            if not isinstance(down, nn.Identity):
                x = self.t_conv("down", down.conv, x, None, k=(3, 4, 4), s=(2, 2))
        x = self._layer(unet.mid_block1, x, None)
        x = self._layer(unet.mid_attn, x, None)
        x = self._layer(unet.mid_block2, x, None)
        for level_blocks, up in zip(unet.up_blocks, unet.up_samples):
            for j, block_list in enumerate(level_blocks):
                skip = skips.pop() if j == 0 else None
                for layer in block_list:
                    x = self._layer(layer, x, skip)
                    skip = None
            if not isinstance(up, nn.Identity):
                x = self.t_conv("up", up.conv, x, None, transposed=True, k=(3, 4, 4), s=(2, 2))
        gn, conv = unet.conv_out[0], unet.conv_out[2]
        st = self.gn_colsum(x)
        slot = self.gn_finalize(x, gn.num_groups, st)
        y = self.t_gn(x, slot, gn, silu_pre=True)
        self.d_eps = Act(self.persistent((n * vox * self.Lp,), torch.bfloat16, zero=True), n, self.Lp, d, h, w, 0)
        self.t_conv("conv_out", conv, y, None, f32_out=self.eps, f32_strides=(vox * L, 1, h * w * L, w * L, L),
                    gy=self.d_eps)

        def run_loss():
            lib.mse_loss_fwd(_ptr(prog.eps), _ptr(prog.noise), _ptr(prog.mask) if prog.use_mask else None,
                             _ptr(prog.norm), n, L, d, h, w, _ptr(prog.loss_ws), _ptr(prog.loss_out), sptr)

        self._emit(run_loss, "loss.fwd")
        self.n_fwd = len(self.ops)
        self.generation = 0   # bumped by every run_forward (see there)

        # ---- backward: loss, then the tape in reverse, then the time embedding ---------------------------------
        def run_loss_bwd():
            lib.mse_loss_bwd(_ptr(prog.eps), _ptr(prog.noise), _ptr(prog.mask) if prog.use_mask else None,
                             _ptr(prog.norm), _ptr(prog.gscale), n, L, d, h, w, prog.d_eps.ip, prog.Lp, sptr)

        self._emit(run_loss_bwd, "loss.bwd")
        for fn in reversed(self.tape):
            fn()
        self._emit_deferred_lin()
        self._time_embed_bwd()
        self.finalize_layout()
        for key, need in self._need.items():
            self._ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)

    def _emit_deferred_lin(self):
        """ONE launch for the weight and bias gradients of every deferred pointwise layer (the 22 attention projections):
        device tables of layers and of 64 x 64 dW tiles, built once."""
        import struct
        if not self._deferred_lin:
            return
        ents, blks, fl = [], [], 0.0
        for i, (xa, ga, gw, gw_off, gb, gb_off, b_scale, rows, cin, cout, dw_stride) in enumerate(self._deferred_lin):
            dbp = 0 if gb is None else gb.data_ptr() + 4 * gb_off
            ents.append(struct.pack("<QQQQiiiifi", xa.ip.value, ga.ip.value, gw.data_ptr() + 4 * gw_off, dbp, rows, cin, cout,
                                    dw_stride, b_scale, 0))
            for ct in range((cout + 63) // 64):
                for it in range((cin + 63) // 64):
                    blks.append(struct.pack("<iiii", i, ct, it, 0))
            fl += 2.0 * rows * cin * cout
        dev = self.ctx.device
        et = torch.frombuffer(bytearray(b"".join(ents)), dtype=torch.uint8).to(dev)
        bt = torch.frombuffer(bytearray(b"".join(blks)), dtype=torch.uint8).to(dev)
        self.keep.extend([et, bt])
        lib, sptr, nb = self.lib, self.ctx.sptr, len(blks)

        def run():
            lib.linear_wgrad_multi(_ptr(et), _ptr(bt), nb, sptr)

        self.flops += fl
        self._emit(run, "attn.wgrad+bgrad.batched", fl, "linear_wgrad_multi")
        for (xa, ga, *_rest) in self._deferred_lin:
            self.release(ga)

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _ws_ptr(self, key):
        return C.c_void_p(self._ws[key].data_ptr())

    def grad_alloc(self, shape) -> torch.Tensor:
        """A zeroed fp32 gradient buffer: a 256-byte-aligned piece of the arena (a separate allocation once the arena is full)."""
        numel = 1
        for v in shape:
            numel *= int(v)
        if self._garena_used + numel > self._garena.numel():
            return self.persistent(tuple(shape), torch.float32, zero=True)
        g = self._garena[self._garena_used:self._garena_used + numel].view(tuple(shape))
        self._garena_used += (numel + 63) // 64 * 64
        return g

    def grad_buf(self, p: torch.Tensor, rows_pad: Optional[int] = None) -> torch.Tensor:
        g = self.grads.get(id(p))
        if g is None:
            shape = list(p.shape)
            if rows_pad is not None and rows_pad > shape[0]:
                shape[0] = rows_pad
            g = self.grad_alloc(shape)
            self.grads[id(p)] = g
        return g

    def act_like(self, a: Act) -> Act:
        return self.act(a.n, a.c, a.d, a.h, a.w, halo=0)

    def into_grad(self, a: Act, produce: Callable[[Act], None]):
        """Run `produce(dst)` so that it writes a gradient contribution of `a`; the first contribution lands in
        a.grad directly, later ones go through a temporary and ctsi_add_bf16."""
        if a.grad is None:
            a.grad = self.act_like(a)
            produce(a.grad)
            return
        tmp = self.act_like(a)
        produce(tmp)
        lib, sptr = self.lib, self.ctx.sptr
        gp, tp, cnt = a.grad.ip, tmp.ip, a.n * a.vox * a.c

        def run():
            lib.add_bf16(gp, tp, cnt, sptr)

        self._emit(run, "grad.add")
        self.release(tmp)

    def _layer(self, layer, x: Act, skip: Optional[Act]) -> Act:
        kind = type(layer).__name__
        if kind == "ResBlock3D":
            y = self.t_resblock(layer, x, skip)
        elif kind == "TemporalAttention":
            y = self.t_attention(layer, x)
        else:
            raise CtsiError(f"unsupported U-Net layer {kind}")
        return y

    # ---- conv ----------------------------------------------------------------------------------------------------
    def t_conv(self, name, m, x1: Act, x2: Optional[Act], *, transposed=False, k=(3, 3, 3), s=(1, 1), want_stats=False,
               need_dx=True, f32_out=None, f32_strides=None, gy: Optional[Act] = None, ret_stats=False,
               bias_from_gn=False):
        p = (1, 1, 1) if k != (1, 1, 1) else (0, 0, 0)
        cout = m.out_channels
        out, st = self.conv(name, lambda: m.weight, lambda: m.bias, x1, x2, transposed=transposed, k=k, s=s, p=p,
                            cout=cout, want_stats=want_stats, f32_out=f32_out, f32_strides=f32_strides)

        def bwd():
            g = gy if gy is not None else out.grad
            if g is None:
                raise CtsiError(f"internal: no gradient reached the output of {name}")
            self._conv_bwd(name, m.weight, None if bias_from_gn else m.bias, x1, x2, g, transposed, k, s, p, cout,
                           need_dx)
            if gy is None:
                self.release(out.grad)
                out.grad = None

        self.tape.append(bwd)
        return (out, st) if ret_stats else out

    def _conv_bwd(self, name, wparam, bparam, x1, x2, g: Act, transposed, k, s, p, cout, need_dx,
                  w_rows=None, b_scale=1.0, gw: Optional[torch.Tensor] = None, gb: Optional[torch.Tensor] = None,
                  gw_off=0, gb_off=0, w_src: Optional[Callable[[], torch.Tensor]] = None, defer_wb: bool = False):
        """Emit bias / weight / data gradient launches of one conv whose output gradient is `g`.
        gw/gb (+ element offsets) override the destination (used for the V slice of attention's qkv);
        w_src overrides the weight the data gradient uses (same slice)."""
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        T = k[0] * k[1] * k[2]
        cin = x1.c + (0 if x2 is None else x2.c)
        cw_out = wparam.shape[1] if transposed else wparam.shape[0]
        if gw is None:
            pad = g.c if (not transposed and g.c > wparam.shape[0]) else None
            gw = self.grad_buf(wparam, rows_pad=pad)
        if gb is None and bparam is not None:
            gb = self.grad_buf(bparam, rows_pad=g.c if g.c > bparam.shape[0] else None)
        rows = g.n * g.vox
        if defer_wb:
            # a small 1x1x1 layer (rows of a depth-summed tensor): weight and bias gradient join the batched launch at the end of
            # the backward pass (ctsi_linear_wgrad_multi); x1 and g must stay alive until then
            if transposed or T != 1 or x2 is not None:
                raise CtsiError("internal: only plain pointwise layers can defer their weight gradient")
            self._deferred_lin.append((x1, g, gw, gw_off, gb, gb_off, float(b_scale), rows, x1.c, g.c, wparam.shape[1]))
        # bias: channel sums of the output gradient
        if gb is not None and not defer_wb:
            self._need["chsum"] = max(self._need["chsum"], 4 * lib.channel_sum_workspace_floats(rows, g.c))
            gp, gbp, gc = g.ip, C.c_void_p(gb.data_ptr() + 4 * gb_off), g.c

            def run_b():
                lib.channel_sum(gp, rows, gc, gc, prog._ws_ptr("chsum"), gbp, b_scale, sptr)

            self._emit(run_b, name + ".bgrad")
        # weight
        srcs = [(x1, 0)] + ([(x2, x1.c)] if x2 is not None else [])
        for xa, coff in ([] if defer_wb else srcs):
            if transposed:   # weight (cin, cout, T): R = layer input, G = output gradient
                r_act, g_act = xa, g
                sr, sg = cw_out * T, T
                off = gw_off + coff * cw_out * T
            else:            # weight (cout, cin, T): R = output gradient, G = layer input
                r_act, g_act = g, xa
                sr, sg = wparam.shape[1] * T, T
                off = gw_off + coff * T
            desc = WgradDesc(k[0], k[1], k[2], s[0], s[1], p[0], p[1], p[2], r_act.n, r_act.d, r_act.h, r_act.w,
                             g_act.d, g_act.h, g_act.w, r_act.c, r_act.c, g_act.c, g_act.c)
            self.keep.append(desc)
            self._need["wgrad"] = max(self._need["wgrad"], lib.wgrad_workspace_bytes(C.byref(desc)))
            fl = lib.wgrad_flops(C.byref(desc))
            self.flops += fl
            rp, gp2, dwp = r_act.ip, g_act.ip, C.c_void_p(gw.data_ptr() + 4 * off)

            def run_w(desc=desc, rp=rp, gp2=gp2, dwp=dwp, sr=sr, sg=sg):
                lib.wgrad(C.byref(desc), rp, gp2, prog._ws_ptr("wgrad"), prog._ws["wgrad"].numel(), dwp, sr, sg, 1, 1.0, sptr)

            self._emit(run_w, name + ".wgrad", fl, "conv_wgrad")
        # data
        if not need_dx:
            return
        if transposed:      # ConvTranspose3d layer: dx = strided Conv3d of g with the same weight tensor
            self.into_grad(x1, lambda dst: self.conv(name + ".dgrad", lambda: wparam, None, g, None, k=k, s=s, p=p,
                                                     cout=x1.c, out=dst))
        elif s != (1, 1):   # strided Conv3d layer: dx = ConvTranspose3d of g with the same weight tensor
            self.into_grad(x1, lambda dst: self.conv(name + ".dgrad", lambda: wparam, None, g, None, transposed=True,
                                                     k=k, s=s, p=p, cout=x1.c, out=dst))
        else:               # stride-1 'same' conv: flipped, transposed weights, one launch per concatenated source
            co_w, ci_w = wparam.shape[0], wparam.shape[1]
            if w_src is not None:
                co_w = cout
            for xa, coff in srcs:
                def wfn(coff=coff, cnt=xa.c):
                    src = (w_src() if w_src is not None else wparam).detach().contiguous()
                    outw = torch.empty((cnt, co_w) + tuple(wparam.shape[2:]), dtype=torch.float32,
                                       device=self.ctx.device)
                    lib.weight_dgrad_layout(_ptr(src), _ptr(outw), co_w, ci_w, T, coff, cnt, sptr)
                    src.record_stream(self.ctx.stream)
                    return outw

                # (fast_repack: the same re-layout from a pointer table, into a buffer the program keeps)
                wfn.fast_layout = ((w_src if w_src is not None else (lambda: wparam)), co_w, ci_w, T, coff, xa.c)

                self.into_grad(xa, lambda dst, wfn=wfn, cnt=xa.c: self.conv(
                    name + ".dgrad", wfn, None, g, None, k=k, s=(1, 1), p=p, cout=cnt, out=dst,
                    cin_w=(co_w if co_w != g.c else None)))

    # ---- GroupNorm chain ---------------------------------------------------------------------------------------------
    def t_gn(self, x: Act, slot: int, gn: nn.GroupNorm, *, silu_pre: bool, tb_off: Optional[int] = None,
             residual: Optional[Act] = None, silu_post: bool = False, conv_bias: Optional[torch.Tensor] = None) -> Act:
        """`conv_bias`: bias parameter of the convolution that produced x; its gradient (sum of dx) then comes out of
        the GroupNorm backward's statistics and that conv's backward skips its own channel-sum pass."""
        out = self.gn_apply(x, slot, gn, silu_pre=silu_pre, tbias=self.tbias if tb_off is not None else None,
                            tbias_off=tb_off or 0, tbias_stride=self.total_out, residual=residual, silu_post=silu_post)
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        gamma = self.dev_f32(lambda: gn.weight)
        beta = self.dev_f32(lambda: gn.bias)

        def bwd():
            gy = out.grad
            if gy is None:
                raise CtsiError("internal: no gradient reached a GroupNorm output")
            self._gn_bwd(x, gy, False, slot, gn, gamma, beta, silu_pre, tb_off, residual, silu_post, None,
                         dxsum=None if conv_bias is None else self.grad_buf(conv_bias))
            self.release(gy)
            out.grad = None

        self.tape.append(bwd)
        return out

    def _gn_bwd(self, x: Act, gy: Act, bcast: bool, slot: int, gn, gamma, beta, silu_pre, tb_off, residual, silu_post,
                add: Optional[Act], dxsum: Optional[torch.Tensor] = None):
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        n, c, d, h, w = x.n, x.c, x.d, x.h, x.w
        groups, eps = gn.num_groups, float(gn.eps)
        self._need["gn"] = max(self._need["gn"], 4 * lib.gn_bwd_workspace_floats(n, c, d, h, w, groups))
        if x.grad is not None:
            raise CtsiError("internal: a normalised tensor has a second consumer")
        x.grad = self.act_like(x)
        # g_buf (gradient of the GroupNorm output) is the residual's gradient when no SiLU sits in between
        tmp = None
        if residual is not None and not silu_pre and residual.grad is None:
            residual.grad = self.act_like(residual)
            gbuf = residual.grad
            add_to_res = False
        elif residual is None and not silu_post:
            gbuf = None         # nobody but pass 3 needs the GroupNorm output's gradient: it re-derives it from dy (no buffer, no write)
            add_to_res = False
        else:
            tmp = self.act_like(x)
            gbuf = tmp
            add_to_res = residual is not None
        if residual is not None and silu_pre:
            raise CtsiError("internal: residual after a pre-SiLU is not a combination the U-Net uses")
        dgam, dbet = self.grad_buf(gn.weight), self.grad_buf(gn.bias)
        xp, gyp, gp, bp = x.ip, gy.ip, _ptr(gamma), _ptr(beta)
        rp = C.c_void_p(0) if residual is None else residual.ip
        ap = C.c_void_p(0) if add is None else add.ip
        gbp, dxp, dgp, dbp = (C.c_void_p(0) if gbuf is None else gbuf.ip), x.grad.ip, _ptr(dgam), _ptr(dbet)
        dtp = C.c_void_p(0 if tb_off is None else self.d_tbias.data_ptr() + 4 * tb_off)
        tstride = self.total_out
        dxsp = _ptr(dxsum)

        def run():
            lib.gn_bwd(xp, gyp, int(bcast), C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), gp, bp, n, c, d, h, w,
                       groups, eps, int(silu_pre), rp, int(silu_post), ap, gbp, dxp, prog._ws_ptr("gn"), dgp, dbp, dtp,
                       tstride, dxsp, sptr)

        self._emit(run, "gn.bwd")
        if add_to_res:
            rgp, cnt = residual.grad.ip, n * x.vox * c

            def run_add():
                lib.add_bf16(rgp, gbp, cnt, sptr)

            self._emit(run_add, "grad.add")
        if tmp is not None:
            self.release(tmp)

    # ---- ResBlock3D (models/unet3d.py:116-133) ------------------------------------------------------------------------
    def t_resblock(self, m, x: Act, skip: Optional[Act]) -> Act:
        if isinstance(m.residual_conv, nn.Identity):
            if skip is not None:
                raise CtsiError("identity residual with a concatenated input")
            r = x
        else:
            r = self.t_conv("res1x1", m.residual_conv, x, skip, k=(1, 1, 1))
        c1, st = self.t_conv("rb.conv1", m.conv1.conv, x, skip, want_stats=True, ret_stats=True, bias_from_gn=True)
        slot = self.gn_finalize(c1, m.conv1.norm.num_groups, st)
        h1 = self.t_gn(c1, slot, m.conv1.norm, silu_pre=True, tb_off=self.tb_off[id(m)], conv_bias=m.conv1.conv.bias)
        c2, st = self.t_conv("rb.conv2", m.conv2[0], h1, None, want_stats=True, ret_stats=True, bias_from_gn=True)
        slot = self.gn_finalize(c2, m.conv2[1].num_groups, st)
        out = self.t_gn(c2, slot, m.conv2[1], silu_pre=False, residual=r, silu_post=True, conv_bias=m.conv2[0].bias)
        return out

    # ---- TemporalAttention in its depth-sum form (see engine.Program.attention) ---------------------------------------------
    def t_attention(self, m, x: Act) -> Act:
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        n, c, d, h, w = x.n, x.c, x.d, x.h, x.w
        tps = lib.attn_depthsum_tiles(c, h, w)
        self._colsum_need = max(self._colsum_need, 2 * n * tps * c)
        depthsum = self.persistent((n * h * w * c,), torch.float32)
        xp, dsp = x.ip, _ptr(depthsum)

        def run_ds():
            lib.attn_depthsum(xp, dsp, _ptr(prog._colsum), n, c, d, h, w, sptr)

        self._emit(run_ds, "attn.depthsum")
        slot = self.gn_finalize(x, m.norm.num_groups, dict(tps=tps, cpad=c, nclass=1))
        gamma = self.dev_f32(lambda: m.norm.weight)
        beta = self.dev_f32(lambda: m.norm.bias)
        xs = self.act(n, c, 1, h, w, halo=0)
        groups, eps = m.norm.num_groups, float(m.norm.eps)
        gp, bp, xsp = _ptr(gamma), _ptr(beta), xs.ip

        def run_ns():
            lib.attn_normsum(dsp, C.c_void_p(prog._gn_sums.data_ptr() + slot * 8), gp, bp, xsp, n, c, d, h, w, groups,
                             eps, sptr)

        self._emit(run_ns, "attn.normsum")
        wv = lambda: m.qkv.weight[2 * c:3 * c]
        def bv():                      # the V bias is added once per depth slice of the sum
            return float(d) * m.qkv.bias[2 * c:3 * c]

        bv.parts, bv.scale = [lambda: m.qkv.bias[2 * c:3 * c]], float(d)     # (fast_repack: a scaled parameter view)
        u, _ = self.conv("attn.v", wv, bv, xs, None, k=(1, 1, 1), p=(0, 0, 0), cout=c)
        pterm, _ = self.conv("attn.proj", lambda: m.proj_out.weight, lambda: m.proj_out.bias, u, None, k=(1, 1, 1),
                             p=(0, 0, 0), cout=c)
        out = self.act(n, c, d, h, w, halo=0)
        pp, op_ = pterm.ip, out.ip

        def run_ba():
            lib.attn_broadcast_add(xp, pp, None, m.num_heads, op_, n, c, d, h, w, sptr)

        self._emit(run_ba, "attn.broadcast_add")

        def bwd():
            gy = out.grad
            if gy is None:
                raise CtsiError("internal: no gradient reached an attention output")
            # dP = sum over depth of gy  (fp32 -> bf16 (n,1,h,w,c))
            dsum = self.pool.get(n * h * w * c, torch.float32)
            dP = self.act(n, c, 1, h, w, halo=0)
            gyp, dsp2, dPp = gy.ip, _ptr(dsum), dP.ip

            def run_dsum():
                lib.attn_depthsum(gyp, dsp2, _ptr(prog._colsum), n, c, d, h, w, sptr)
                lib.f32_to_bf16(dsp2, dPp, n * h * w * c, sptr)

            self._emit(run_dsum, "attn.bwd.depthsum")
            # proj_out: dW_p, db_p, du
            defer = not os.environ.get("CTSI_TRAIN_NO_LIN_BATCH")     # (A/B timing, tests: one wgrad + bgrad launch pair per layer)
            self._conv_bwd("attn.proj", m.proj_out.weight, m.proj_out.bias, u, None, dP, False, (1, 1, 1), (1, 1),
                           (0, 0, 0), c, True, defer_wb=defer)
            du = u.grad
            # V third of qkv: dW_v, db_v (x D: the bias is added once per depth slice), d(xs)
            gqkv_w, gqkv_b = self.grad_buf(m.qkv.weight), self.grad_buf(m.qkv.bias)
            self._conv_bwd("attn.v", m.qkv.weight, m.qkv.bias, xs, None, du, False, (1, 1, 1), (1, 1), (0, 0, 0), c,
                           True, b_scale=float(d), gw=gqkv_w, gb=gqkv_b, gw_off=2 * c * c, gb_off=2 * c, w_src=wv, defer_wb=defer)
            dxs = xs.grad
            # GroupNorm under the depth sum: dy is d(xs) broadcast over depth; + gy (the identity path)
            self._gn_bwd(x, dxs, True, slot, m.norm, gamma, beta, False, None, None, False, gy)
            for a in ((dxs, gy) if defer else (dP, du, dxs, gy)):      # (deferred: dP and du are read by the batched launch)
                self.release(a)
            self.pool.put(dsum)
            u.grad = xs.grad = out.grad = None

        self.tape.append(bwd)
        return out

    # ---- time embedding backward ------------------------------------------------------------------------------------------
    def _time_embed_bwd(self):
        lib, sptr, prog = self.lib, self.ctx.sptr, self
        te = self.unet.time_embed.time_mlp
        n, dim, td, tot = self.n, self.dim, self.time_dim, self.total_out
        gw1, gb1 = self.grad_buf(te[1].weight), self.grad_buf(te[1].bias)
        gw2, gb2 = self.grad_buf(te[3].weight), self.grad_buf(te[3].bias)
        sincos = self.te_scratch
        lin1 = self.te_scratch[n * dim:]
        temb = self.te_scratch[n * dim + n * td:]

        def run():
            lib.linear_bwd(_ptr(temb), _ptr(prog.w_all), _ptr(prog.d_tbias), n, td, tot, 1, _ptr(prog.g_w_all),
                           _ptr(prog.g_b_all), _ptr(prog.g_temb), sptr)
            lib.linear_bwd(_ptr(lin1), _ptr(prog.w2), _ptr(prog.g_temb), n, td, td, 1, _ptr(gw2), _ptr(gb2),
                           _ptr(prog.g_lin1), sptr)
            lib.linear_bwd(_ptr(sincos), _ptr(prog.w1), _ptr(prog.g_lin1), n, dim, td, 0, _ptr(gw1), _ptr(gb1), None,
                           sptr)

        self._emit(run, "time_embed.bwd")
        # the per-block Linear gradients are row slices of the stacked buffers
        for m in self.blocks:
            off = self.tb_off[id(m)]
            co = m.time_mlp[1].out_features
            self.grads[id(m.time_mlp[1].weight)] = self.g_w_all[off:off + co]
            self.grads[id(m.time_mlp[1].bias)] = self.g_b_all[off:off + co]

    def needs_rebuild(self) -> bool:
        """The autograd bridge hands gradients to the Parameter objects captured at build time: a replaced
        parameter (load_state_dict(assign=True), module.weight = ...) needs a new program, not just a repack."""
        cur = list(self.unet.parameters())
        return len(cur) != len(self.params) or any(a is not b for a, b in zip(cur, self.params))

    # ---- execution ---------------------------------------------------------------------------------------------------------
    def set_diffusion(self, diffusion):
        self.sqrt_ac = diffusion.sqrt_alphas_cumprod.detach().to(self.ctx.device, torch.float32).contiguous()
        self.sqrt_1mac = diffusion.sqrt_one_minus_alphas_cumprod.detach().to(self.ctx.device, torch.float32).contiguous()

    def run_forward(self, z0, cond, t, noise, norm, mask=None) -> torch.Tensor:
        """Runs the forward launches and overwrites the tape (saved activations, t, noise) of this program: the
        generation counter lets a backward that belongs to an EARLIER forward of the same shape fail loudly instead of
        differentiating through the wrong activations (l1 = model(a); l2 = model(b); (l1 + l2).backward())."""
        self.ensure_fresh()
        self.generation += 1
        self.z0.copy_(z0)
        self.cond.copy_(cond)
        self.noise.copy_(noise)
        self.t_rows.copy_(t.to(torch.int32))
        self.norm.copy_(norm.to(torch.float32))
        self.use_mask = mask is not None
        if mask is not None:
            self.mask.copy_(mask.to(torch.float32))
        for op in self.ops[:self.n_fwd]:
            op()
        return self.loss_out[0].clone()

    def run_backward(self, grad_out: torch.Tensor, generation: Optional[int] = None) -> List[torch.Tensor]:
        if generation is not None and generation != self.generation:
            raise CtsiError(
                "backward of a training forward whose saved activations were overwritten: another forward of the same "
                f"shape ran on this program in between (tape generation {generation}, now {self.generation}).  Call "
                "backward() before the next forward of that shape, or evaluate the second batch under torch.no_grad() "
                "through model.generate / unet(x, t, c), which do not touch the training tape.")
        self.gscale.copy_(grad_out.reshape(1).to(torch.float32))
        for op in self.ops[self.n_fwd:]:
            op()
        out = []
        for p in self.params:
            g = self.grads.get(id(p))
            if g is None:
                raise CtsiError("internal: a U-Net parameter received no gradient buffer")
            out.append(g[:p.shape[0]] if g.shape[0] != p.shape[0] else g)
        return out


def _clone_grads(grads: List[torch.Tensor]) -> List[torch.Tensor]:
    """Copies of the program's gradient buffers for autograd to own (the program overwrites its buffers in the next backward).
    Buffers that share a storage -- the program's gradient arena -- are copied as ONE range and returned as views of the copy
    at the same offsets; a storage whose members cover less than half of the range they span is copied tensor by tensor."""
    if os.environ.get("CTSI_TRAIN_NO_GRAD_ARENA"):          # A/B timing: one copy per parameter, as before round 4
        return [g.clone() for g in grads]
    groups: Dict[int, List[int]] = {}
    for i, g in enumerate(grads):
        groups.setdefault(g.untyped_storage().data_ptr(), []).append(i)
    out: List[Optional[torch.Tensor]] = [None] * len(grads)
    for idx in groups.values():
        g0 = grads[idx[0]]
        lo = min(grads[i].storage_offset() for i in idx)
        hi = max(grads[i].storage_offset() + grads[i].numel() for i in idx)
        dense = all(grads[i].is_contiguous() and grads[i].dtype == g0.dtype for i in idx)
        if len(idx) == 1 or not dense or (hi - lo) > 2 * sum(grads[i].numel() for i in idx):
            for i in idx:
                out[i] = grads[i].clone()
            continue
        flat = torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,)).clone()
        for i in idx:
            out[i] = flat.as_strided(grads[i].shape, grads[i].stride(), grads[i].storage_offset() - lo)
    return out


class _TrainStep(torch.autograd.Function):
    """loss = training loss of one batch; backward runs the engine's backward launches and hands the parameter
    gradients to autograd (which accumulates them into .grad like any other op)."""

    @staticmethod
    def forward(fctx, prog: UNetTrainProgram, z0, cond, t, noise, norm, mask, *params):
        ectx = prog.ctx
        with ectx.scope():
            loss = prog.run_forward(z0, cond, t, noise, norm, mask)
        fctx.prog = prog
        fctx.generation = prog.generation
        return loss

    @staticmethod
    def backward(fctx, grad_out):
        prog = fctx.prog
        with prog.ctx.scope():
            grads = _clone_grads(prog.run_backward(grad_out, fctx.generation))
        return (None,) * 7 + tuple(grads)


def train_step(prog: UNetTrainProgram, z0, cond, t, noise, norm, mask=None) -> torch.Tensor:
    return _TrainStep.apply(prog, z0, cond, t, noise, norm, mask, *prog.params)
