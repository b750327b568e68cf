"""GPU: the optimizer step on the device (optim.FusedAdamW / FusedAdam, csrc/optim.hip) against torch.optim.AdamW / Adam --
the classes the reference constructs (training/train.py:205-208) -- and the engine's fast re-pack behind it against the
generic re-pack.  Tolerance: parameters and both moments within 1e-6 (relative to the tensor's largest magnitude) of torch's
after 3 steps: the kernel performs torch's single-tensor arithmetic operation by operation in fp32."""
import importlib

import pytest
import torch

from tests.helpers import TINY_CFG, formula_input, formula_noise, rel_l2, tiny_model_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed):
    shapes = [(128, 64, 3, 3, 3), (128,), (7,), (33, 5), (1,), (512, 257), (64, 32, 3, 4, 4), (3, 3)]
    return [torch.nn.Parameter(formula_input(s, seed + i).to(DEV) * (0.5 + 0.1 * i)) for i, s in enumerate(shapes)]


@pytest.mark.parametrize("kind", ["adamw", "adam"])
def test_fused_optimizer_matches_torch(pkg, kind):
    ours, ref = _params(100), _params(100)
    groups = lambda ps: [dict(params=ps[:3], lr=3e-3, name="a"), dict(params=ps[3:], lr=1e-3 * 0.1, name="b")]
    kw = dict(betas=(0.9, 0.98), eps=1e-7, weight_decay=0.05)
    if kind == "adamw":
        o1, o2 = pkg.FusedAdamW(groups(ours), **kw), torch.optim.AdamW(groups(ref), **kw)
    else:
        o1, o2 = pkg.FusedAdam(groups(ours), **kw), torch.optim.Adam(groups(ref), **kw)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ours, ref)):
            g = formula_input(tuple(a.shape), 500 + 10 * step + i).to(DEV) * 0.3
            a.grad, b.grad = g.clone(), g.clone()
        if step == 2:
            ours[4].grad = ref[4].grad = None        # a parameter without a gradient is skipped, like torch does
        v0 = ours[0]._version
        o1.step()
        o2.step()
        assert ours[0]._version > v0                 # the raw-pointer update is visible to torch's version counters
        o1.zero_grad(set_to_none=True)
        o2.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    for a, b in zip(ours, ref):
        scale = float(b.detach().abs().max())
        assert float((a.detach() - b.detach()).abs().max()) <= 1e-6 * scale, tuple(a.shape)
        sa, sb = o1.state[a], o2.state[b]
        for k in ("exp_avg", "exp_avg_sq"):
            # (the moments: a few fp32 ulps of fma-contraction freedom per step in either implementation)
            assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 * float(sb[k].abs().max() + 1e-30), (k, tuple(a.shape))
        assert float(sa["step"]) == float(sb["step"])
    # torch layout of the state: a torch optimizer loads ours and continues identically, and vice versa
    sd = o1.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["name"] == "a"
    o3 = (torch.optim.AdamW if kind == "adamw" else torch.optim.Adam)(groups(ours), **kw)
    o3.load_state_dict(sd)
    o4 = (pkg.FusedAdamW if kind == "adamw" else pkg.FusedAdam)(groups(ref), **kw)
    o4.load_state_dict(o2.state_dict())
    for i, (a, b) in enumerate(zip(ours, ref)):
        g = formula_input(tuple(a.shape), 900 + i).to(DEV) * 0.3
        a.grad, b.grad = g.clone(), g.clone()
    o3.step()
    o4.step()
    torch.cuda.synchronize()
    for a, b in zip(ours, ref):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max())
    with pytest.raises(pkg.CtsiError):           # no CPU path
        _cpu_param_step(pkg)


def _cpu_param_step(pkg):
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    pkg.FusedAdamW([p], lr=1e-3).step()


def test_fast_repack_equals_generic_repack_and_training_continues(pkg):
    """After FusedAdamW.step(engine_modules=[unet]) the training program's bf16 weight images and fp32 operands are what the
    generic re-pack produces from the same parameters (bit for bit), nothing is left to re-pack, and the next micro-step's
    loss equals the one a torch.optim.AdamW twin model reaches through the generic path."""
    model, sd, cfg = tiny_model_sd(pkg)
    twin, _, _ = tiny_model_sd(pkg)
    model.to(DEV)
    twin.to(DEV)
    v_in = formula_input((2, 1, 2, 16, 16), 16).clamp(-1, 1).to(DEV)
    v_gt = formula_input((2, 1, 4, 16, 16), 19).clamp(-1, 1).to(DEV)
    t, nz = torch.tensor([612, 77], device=DEV), formula_noise(-1, (2, 8, 4, 4, 4)).to(DEV)
    for p in list(model.vae.parameters()) + list(twin.vae.parameters()):
        p.requires_grad_(False)
    o1 = pkg.FusedAdamW(model.unet.parameters(), lr=2e-3, weight_decay=0.01, engine_modules=[model.unet])
    o2 = torch.optim.AdamW(twin.unet.parameters(), lr=2e-3, weight_decay=0.01)
    losses = []
    for it in range(3):
        l1, _ = model(v_in, v_gt, t=t, noise=nz)
        l2, _ = twin(v_in, v_gt, t=t, noise=nz)
        losses.append((float(l1), float(l2)))
        l1.backward()
        l2.backward()
        o1.step()
        o2.step()
        o1.zero_grad(set_to_none=True)
        o2.zero_grad(set_to_none=True)
        prog = [pr for k, pr in model.unet.__dict__["_ctsi_programs"].items() if k[0] == "unet-train"][0]
        assert prog._fast is not None and prog._fingerprint() == prog._versions      # nothing left to re-pack
        if it == 0:
            assert prog._fast["nseg"] > 20 and len(prog._fast["packs"]) > 20
            assert len(prog._fast["slow"]) == 0, "every operand of the training program is expressible as a table entry"
            fast = [h["holder"][0].clone() for h in prog._pack_meta] + [e["buf"].clone() for e in prog._f32_meta]
            with prog.ctx.scope():
                prog.repack()                                                         # the generic path, same parameters
            torch.cuda.synchronize()
            slow = [h["holder"][0] for h in prog._pack_meta] + [e["buf"] for e in prog._f32_meta]
            assert all(torch.equal(a, b) for a, b in zip(fast, slow))
    print("losses (fused / torch twin):", losses)
    assert losses[0][0] == pytest.approx(losses[0][1], rel=1e-6)
    for a, b in losses[1:]:
        assert abs(a - b) <= 2e-2 * abs(b)          # bf16 engine: the trajectories agree to the training tolerance
    assert losses[2][0] < losses[0][0]              # and the optimizer optimizes
    for p1, p2 in zip(model.unet.parameters(), twin.unet.parameters()):
        assert rel_l2(p1.detach(), p2.detach()) < 5e-2


def test_fused_optimizer_under_gradscaler_and_lr_scheduler(pkg):
    """The reference's loop steps its optimizer through a GradScaler under AMP (training/trainer.py:237-247) and drives the
    learning rate with a torch scheduler: both work on the drop-in unchanged, and match torch.optim.AdamW under the same."""
    ours, ref = _params(300)[:4], _params(300)[:4]
    o1, o2 = pkg.FusedAdamW(ours, lr=1e-2, weight_decay=0.02), torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.02)
    s1 = torch.optim.lr_scheduler.StepLR(o1, step_size=1, gamma=0.5)
    s2 = torch.optim.lr_scheduler.StepLR(o2, step_size=1, gamma=0.5)
    g1, g2 = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    for step in range(3):
        for ps, opt, sc, sch in ((ours, o1, g1, s1), (ref, o2, g2, s2)):
            loss = sum((p * formula_input(tuple(p.shape), 700 + 10 * step + i).to(DEV)).sum() for i, p in enumerate(ps))
            opt.zero_grad(set_to_none=True)
            sc.scale(loss).backward()
            sc.step(opt)
            sc.update()
            sch.step()
    torch.cuda.synchronize()
    assert o1.param_groups[0]["lr"] == pytest.approx(o2.param_groups[0]["lr"]) == pytest.approx(1e-2 * 0.125)
    for a, b in zip(ours, ref):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max())
    # an overflowing step is skipped by the scaler for both
    ours[0].grad = torch.full_like(ours[0], float("inf"))
    before = ours[0].detach().clone()
    g1._per_optimizer_states.clear()
    g1.unscale_(o1)
    g1.step(o1)
    assert torch.equal(ours[0].detach(), before)
