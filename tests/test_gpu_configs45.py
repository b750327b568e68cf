"""GPU: BASELINE configs 4 and 5 at their REAL per-GPU workload, on one device.

config 4  one 8->48 @512^2 volume depth-sharded over 8 GPUs: latent (1,8,48,128,128), 6 slices per rank.  The eight
          per-rank programs -- exactly what each RCCL rank runs, interior/boundary overlap split on, 4-12.6 MB halo
          messages, 3-launch convs on 3072-block grids -- are driven in lock-step on one device (parallel.LocalComm; only
          the transport differs) and checked against the unsharded engine AND the fp32 oracle; likewise the sharded
          production VAE decode at 512^2 (67 MB halos at full resolution).
config 5  4 volumes of 512^2 per GPU in ONE captured hipGraph step: every sample of the batch must equal its own B = 1
          run; plus generate() of a batch of 2 end to end.
Tolerances as in tests/test_gpu_fullsize.py (networks: rel-L2 <= 3e-2 vs the fp32 oracle)."""
import importlib

import pytest
import torch

from oracle import ref_ops as R
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NET_TOL = 3e-2
E = importlib.import_module("video-to-video-diffusion_amd.engine")
P = importlib.import_module("video-to-video-diffusion_amd.parallel")
S = importlib.import_module("video-to-video-diffusion_amd.sampler")
UNET_CFG = dict(model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=[1, 2, 4, 4], num_heads=4,
                scaling_factor=1.0)
FULL_CFG = {'model': {'in_channels': 1, 'latent_dim': 8, 'vae_base_channels': 128, 'vae_scaling_factor': 1.0},
            'pretrained': {'use_pretrained': True, 'vae': {'enabled': True, 'checkpoint_path': 'unused'}},
            'noise_schedule': 'cosine', 'diffusion_timesteps': 1000}


@pytest.fixture(autouse=True)
def _convt_as_forward_conv(monkeypatch):
    monkeypatch.setattr(R, "CONVT_AS_CONV", True)     # (see tests/test_gpu_fullsize.py: MIOpen's fp32 ConvT search)


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _free():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# config 4
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_unet_world8_at_full_latent(pkg):
    torch.manual_seed(0)
    un = pkg.UNet3D(latent_dim=8).eval().to(DEV)
    sd = {k: v.detach() for k, v in un.state_dict().items()}
    n, L, d, h, w = 1, 8, 48, 128, 128
    x, c = _randn((n, L, d, h, w), 1).to(DEV), _randn((n, L, d, h, w), 2).to(DEV)
    t = torch.tensor([500], device=DEV)
    ctx = E.Ctx.get(torch.device(DEV))
    world = 8
    with ctx.scope():
        ref = E.UNetProgram(ctx, un, n, d, h, w, 2)
        ref.load_latents(x, c)
        ref.set_schedule([500])
        ref.run()
        eps_unsharded = ref.eps_ncdhw()
        del ref
        comm = P.LocalComm(world)
        progs = []
        for r in range(world):
            pr = E.UNetProgram(ctx, un, n, d // world, h, w, 2, shard=P.ShardSpec(r, world, comm, d))
            pr.load_latents(x, c)
            pr.set_schedule([500])
            progs.append(pr)
        meta = progs[3].op_meta
        nsync = sum(1 for m in meta if m[2] == "comm")
        nsplit = sum(1 for m in meta if m[0] == "halo.exchange.async")
        # the halo messages of an inner rank: bytes of one boundary slice per exchanging sync point
        print(f"config 4, rank 3 of 8: {nsync} sync points, {nsplit} overlapped exchanges (3-launch convs), "
              f"{len(meta)} launches; 6 slices + 2 halo slices per tensor")
        assert nsync <= 70 and nsplit >= 4, "the interior/boundary split must be on at 6 slices per rank"
        P.run_lockstep(progs)
        eps = torch.cat([p.eps_ncdhw() for p in progs], dim=2)
        del progs
    torch.cuda.synchronize()
    _free()
    with torch.no_grad():
        ref32 = R.unet_forward(sd, UNET_CFG, x, t, c)
    e_un, e_or, e_un_or = rel_l2(eps, eps_unsharded), rel_l2(eps, ref32), rel_l2(eps_unsharded, ref32)
    print(f"world-8 sharded U-Net at (1,8,48,128,128): rel-L2 vs unsharded {e_un:.3g}, vs fp32 oracle {e_or:.3g} "
          f"(unsharded vs oracle {e_un_or:.3g})")
    assert torch.isfinite(eps).all() and e_un < NET_TOL and e_or < NET_TOL
    un.invalidate_engine_cache()
    _free()


def test_config4_vae_decode_world8_at_512(pkg):
    torch.manual_seed(0)
    vae = pkg.VideoVAE(in_channels=1, latent_dim=8, base_channels=128, scaling_factor=1.0).eval().to(DEV)
    sd = {k: v.detach() for k, v in vae.state_dict().items()}
    z = _randn((1, 8, 48, 128, 128), 4).to(DEV)
    ctx = E.Ctx.get(torch.device(DEV))
    world = 8
    ref = vae.decode(z)
    vae.invalidate_engine_cache()
    _free()
    with ctx.scope():
        comm = P.LocalComm(world)
        progs = []
        for r in range(world):
            pr = E.VAEDecodeProgram(ctx, vae, 1, 48 // world, 128, 128, shard=P.ShardSpec(r, world, comm, 48))
            pr.load(z)
            progs.append(pr)
        nsync = sum(1 for m in progs[0].op_meta if m[2] == "comm")
        P.run_lockstep(progs)
        out = torch.cat([p.out for p in progs], dim=2)
        del progs
    torch.cuda.synchronize()
    _free()
    e = rel_l2(out, ref)
    print(f"world-8 sharded VAE decode (1,8,48,128,128) -> {tuple(out.shape)}: {nsync} sync points, rel-L2 vs unsharded {e:.3g}")
    assert tuple(out.shape) == (1, 1, 48, 512, 512) and torch.isfinite(out).all() and e < 2e-2
    with torch.no_grad():
        ref32 = R.vae_decode(sd, z, 1.0)
    e_or = rel_l2(out, ref32)
    print(f"  vs fp32 oracle {e_or:.3g} (unsharded vs oracle {rel_l2(ref, ref32):.3g})")
    assert e_or < NET_TOL
    vae.invalidate_engine_cache()
    _free()


# ---------------------------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------------------------
def test_config5_batch4_captured_step_equals_single_volume_runs(pkg):
    """Per-GPU share of config 5: 4 volumes @512^2 through ONE captured step graph (two replays = two DDIM steps, per-sample
    timestep rows); every sample must reproduce its own B = 1 captured run.  The B = 4 plans pick other tiles than the B = 1
    plans for some layers (the grid-fill score sees 4x the blocks: no 384-voxel / split-K forms), i.e. other summation orders;
    the network amplifies such bf16 re-roundings to its noise floor, exactly as sharded vs unsharded (2.1e-2 at this size):
    the bound is the network tolerance, 3e-2, not bitwise equality."""
    torch.manual_seed(0)
    un = pkg.UNet3D(latent_dim=8).eval().to(DEV)
    g = pkg.GaussianDiffusion()
    B, L, d, h, w = 4, 8, 48, 128, 128
    x, c = _randn((B, L, d, h, w), 11).to(DEV), _randn((B, L, d, h, w), 12).to(DEV)
    t_desc = [500, 480]                      # mid-schedule: no z0 clamp chaos (SURVEY 0-5 concerns t = 999)
    coef = S.ddim_coef_rows(g.alphas_cumprod, t_desc, 0.0).to(DEV)
    ctx = E.Ctx.get(torch.device(DEV))

    def run(xs, cs):
        n = xs.shape[0]
        with ctx.scope():
            pr = E.UNetProgram(ctx, un, n, d, h, w, max_rows=len(t_desc) * n)
            pr.add_sampler_step("ddim", False)
            pr.load_latents(xs, cs)
            pr.set_schedule([t for t in t_desc for _ in range(n)], coef)
            pr.capture()
            pr.step_ptr.zero_()
            pr.launch()
            eps1 = pr.eps_ncdhw()
            pr.launch()
            z2 = pr.z_ncdhw()
            kinds = sorted({m[2] for m in pr.op_meta if m[2].startswith("conv_mfma")})
            del pr
        torch.cuda.synchronize()
        return eps1, z2, kinds

    eps4, z4, kinds4 = run(x, c)
    _free()
    assert torch.isfinite(eps4).all() and torch.isfinite(z4).all()
    for b in range(B):
        eps1, z1, kinds1 = run(x[b:b + 1], c[b:b + 1])
        e_eps, e_z = rel_l2(eps4[b:b + 1], eps1), rel_l2(z4[b:b + 1], z1)
        print(f"config 5 sample {b}: eps rel-L2 {e_eps:.3g}, z after 2 steps {e_z:.3g}, bit-equal {torch.equal(z4[b:b + 1], z1)}"
              + (f"; kernel variants differ: B=4 {set(kinds4) - set(kinds1)} / B=1 {set(kinds1) - set(kinds4)}"
                 if kinds4 != kinds1 else ""))
        assert e_eps < NET_TOL and e_z < NET_TOL
        _free()
    un.invalidate_engine_cache()
    _free()


def test_config5_generate_batch_of_two_at_512(pkg):
    torch.manual_seed(0)
    model = pkg.VideoToVideoDiffusion(FULL_CFG).eval().to(DEV)
    v_in = (torch.rand((2, 1, 8, 512, 512), generator=torch.Generator().manual_seed(5)) * 2 - 1).to(DEV)

    def nf_for(sl):
        def nf(i, shape):
            full = torch.randn((2,) + tuple(shape[1:]), generator=torch.Generator().manual_seed(2000 + i))
            return full[sl].to(DEV)
        return nf

    out2 = model.generate(v_in, 'ddim', num_inference_steps=4, target_depth=48, noise_fn=nf_for(slice(0, 2)))
    assert tuple(out2.shape) == (2, 1, 48, 512, 512) and torch.isfinite(out2).all()
    assert float(out2.abs().max()) <= 1.0
    model.invalidate_engine_cache()
    _free()
    for b in range(2):
        o1 = model.generate(v_in[b:b + 1], 'ddim', num_inference_steps=4, target_depth=48, noise_fn=nf_for(slice(b, b + 1)))
        p = R.psnr(out2[b:b + 1], o1, 2.0)
        print(f"generate() B=2 @512^2, sample {b}: PSNR vs its own B=1 run {p:.1f} dB, bit-equal {torch.equal(out2[b:b + 1], o1)}")
        # same arithmetic per sample; only tile choices may differ between the B=2 and B=1 plans, and the chaotic first
        # DDIM step (z0 clamp at t=999) amplifies such rounding differences: the bound is the sampling criterion's
        assert p > 25.0
        model.invalidate_engine_cache()
        _free()
