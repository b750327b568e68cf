"""GPU: whole networks and the sampling loop through the reference's Python API (models.*, inference.*)
against golden vectors from the real reference and against the CPU oracle.

Tolerances (bf16 storage / fp32 accumulate engine vs the reference's fp32):
  U-Net forward       rel-L2 <= 3e-2   (PyTorch's own bf16 autocast of the reference: 2.6e-2)
  VAE encode/decode   rel-L2 <= 3e-2
  end-to-end          PSNR(hip, ref_fp32) >= PSNR(oracle under CPU bf16 autocast, ref_fp32) - 0.1 dB
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import ref_ops as R
from tests.helpers import (MID_UNET, TINY_CFG, TINY_UNET, formula_input, formula_noise, formula_sd, load_formula,
                           rel_l2, tiny_model_sd, unet_cfg)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NET_TOL = 3e-2


def _noise_fn(i, shape):
    return formula_noise(i, shape)


def test_native_library_is_what_runs(pkg):
    lib = pkg.get_lib()
    assert lib.device_available() == 1
    import ctypes
    assert any("libctsi.so" in l for l in open("/proc/self/maps").read().splitlines())


def test_unet_forward_tiny_vs_golden_and_oracle(golden, pkg):
    un = pkg.UNet3D(**TINY_UNET)
    sd = load_formula(un, 8)
    un.to(DEV)
    x, c = formula_input((2, 8, 4, 8, 8), 10), formula_input((2, 8, 4, 8, 8), 11)
    t = torch.tensor([500, 37])
    out = un(x.to(DEV), t.to(DEV), c.to(DEV)).cpu()
    assert out.dtype == torch.float32 and tuple(out.shape) == (2, 8, 4, 8, 8)
    assert rel_l2(out, golden["unet.tiny.out"]) < NET_TOL
    assert rel_l2(out, R.unet_forward(sd, unet_cfg(TINY_UNET), x, t, c)) < NET_TOL
    # exact-mode attention (rowsum(softmax) evaluated, == 1 +- 1e-7) is an equally valid bf16 realisation:
    # it differs from the fast path only by re-rounded intermediates
    un.attention_mode = "exact"
    out_exact = un(x.to(DEV), t.to(DEV), c.to(DEV)).cpu()
    assert rel_l2(out_exact, golden["unet.tiny.out"]) < NET_TOL
    assert rel_l2(out_exact, out) < NET_TOL
    # repeated evaluation is bit-stable
    un.attention_mode = "fast"
    assert torch.equal(un(x.to(DEV), t.to(DEV), c.to(DEV)).cpu(), out)


def test_unet_forward_three_levels_two_attention_levels(golden, pkg):
    un = pkg.UNet3D(**MID_UNET)
    load_formula(un, 9)
    un.to(DEV)
    out = un(formula_input((1, 4, 6, 12, 8), 12).to(DEV), torch.tensor([999], device=DEV),
             formula_input((1, 4, 6, 12, 8), 13).to(DEV)).cpu()
    assert rel_l2(out, golden["unet.mid.out"]) < NET_TOL


def test_unet_weight_update_is_picked_up(pkg):
    un = pkg.UNet3D(**TINY_UNET)
    load_formula(un, 8)
    un.to(DEV)
    x, c = formula_input((1, 8, 2, 4, 4), 1).to(DEV), formula_input((1, 8, 2, 4, 4), 2).to(DEV)
    t = torch.tensor([10], device=DEV)
    a = un(x, t, c)
    with torch.no_grad():
        un.conv_out[2].bias.add_(1.0)
    b = un(x, t, c)
    assert torch.allclose(b - a, torch.ones_like(a), atol=1e-5)


def test_vae_tiny_vs_golden(golden, pkg):
    vae = pkg.VideoVAE(in_channels=1, latent_dim=8, base_channels=16, scaling_factor=0.5)
    load_formula(vae, 10)
    vae.to(DEV)
    z = vae.encode(formula_input((1, 1, 3, 16, 12), 14).to(DEV))
    assert tuple(z.shape) == (1, 8, 3, 4, 3) and z.dtype == torch.float32
    assert rel_l2(z.cpu(), golden["vae.tiny.latent"]) < NET_TOL
    rec = vae.decode(torch.tensor(golden["vae.tiny.latent"]).to(DEV))
    assert rel_l2(rec.cpu(), golden["vae.tiny.recon"]) < NET_TOL
    assert float(rec.abs().max()) <= 1.0
    assert vae.get_latent_shape((1, 1, 3, 16, 12)) == (1, 8, 3, 4, 3)
    recon, z2 = vae(formula_input((1, 1, 3, 16, 12), 14).to(DEV))
    assert torch.equal(z2, z) and tuple(recon.shape) == (1, 1, 3, 16, 12)


def _bf16_autocast_reference(fn):
    with torch.autocast("cpu", dtype=torch.bfloat16):
        return fn().float()


def test_vae_on_sizes_that_are_not_multiples_of_four(pkg):
    """Planes the stride-2 stages do not divide evenly (the reference's Conv3d / ConvTranspose3d floor and double them):
    encode (1,1,3,18,22) -> latent (.,.,3,4,5) -> decode (.,.,3,16,20), against the oracle; odd planes take the gather
    kernel instead of the parity-sub-grid Downsample form."""
    vae = pkg.VideoVAE(in_channels=1, latent_dim=8, base_channels=16, scaling_factor=0.5)
    sd = load_formula(vae, 10)
    vae.to(DEV)
    x = formula_input((1, 1, 3, 18, 22), 14)
    z = vae.encode(x.to(DEV)).cpu()
    z_ref = R.vae_encode(sd, x, 0.5)
    assert tuple(z.shape) == tuple(z_ref.shape) == (1, 8, 3, 4, 5) and rel_l2(z, z_ref) < NET_TOL
    y = vae.decode(z_ref.to(DEV)).cpu()
    y_ref = R.vae_decode(sd, z_ref, 0.5)
    assert tuple(y.shape) == tuple(y_ref.shape) == (1, 1, 3, 16, 20) and rel_l2(y, y_ref) < NET_TOL


def test_ddim_trajectory_and_psnr_criterion(golden, pkg):
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    shape = (1, 8, 4, 8, 8)
    cond = formula_input(shape, 15)
    ref_model = lambda z, t, c: R.unet_forward(sd, cfg, z, t, c, "unet.")
    bufs = R.diffusion_buffers("cosine", 1000)
    for eta in (0.0, 0.5):
        traj = []
        z = pkg.DDIMSampler(model.diffusion, model.unet).sample(shape, cond.to(DEV), 10, DEV, eta=eta, progress=False,
                                                                noise_fn=_noise_fn, trajectory=traj)
        ref = torch.tensor(golden[f"traj.ddim.eta{eta}"])
        assert len(traj) == 11 and tuple(z.shape) == shape
        assert torch.equal(traj[-1], z)
        errs = [rel_l2(traj[i].cpu(), ref[i]) for i in range(11)]
        # the reference pipeline under PyTorch's own bf16 autocast, same inputs/noise
        traj_b = []
        zb = _bf16_autocast_reference(lambda: R.ddim_sample(ref_model, bufs, shape, cond, 10, eta=eta,
                                                            noise_fn=_noise_fn, trajectory=traj_b))
        errs_b = [rel_l2(traj_b[i].float(), ref[i]) for i in range(11)]
        e_hip, e_bf16 = rel_l2(z.cpu(), ref[-1]), rel_l2(zb, ref[-1])
        print(f"eta={eta} per-step rel-L2 {['%.3g' % e for e in errs]}  final hip {e_hip:.3g} vs autocast {e_bf16:.3g}")
        print(f"         autocast per-step  {['%.3g' % e for e in errs_b]}")
        # EVERY step of the trajectory is held to the criterion, not only the end point (a mid-trajectory regression that
        # the z0 clamp washes out of the final latent would otherwise pass): the -0.1 dB of the PSNR criterion is a factor
        # 10^(0.1/20) = 1.0116 on the error against the same reference trajectory
        for i in range(11):
            assert errs[i] <= 1.0116 * errs_b[i], (eta, i, errs[i], errs_b[i])
        # step 0 divides by ~1e-4 and clamps to +-10: elements whose numerator is near zero flip sign under
        # any bf16 perturbation, so the yardstick is the reference's own bf16-autocast trajectory.  ONE criterion, the
        # stated one: PSNR(hip) >= PSNR(autocast) - 0.1 dB, i.e. rmse_hip <= 1.0116 * rmse_autocast on the final latent
        psnr_hip, psnr_bf = R.psnr(z.cpu(), ref[-1], 20.0), R.psnr(zb, ref[-1], 20.0)
        assert psnr_hip >= psnr_bf - 0.1, (psnr_hip, psnr_bf)


def test_nonfinite_values_are_sanitised_and_logged(pkg, caplog):
    """The reference's NaN/Inf checkpoints (inference/sampler.py:268-275, 288-292): values are sanitised on device and
    what the reference would have logged is logged once per sample(), from device-side counters."""
    import logging
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    shape = (1, 8, 4, 8, 8)
    cond = formula_input(shape, 15)
    cond[0, 1, 2, 3, 4] = float("inf")

    def nf(i, s_):
        z = formula_noise(i, s_)
        if i == -1:
            z[0, 0, 0, 0, :3] = float("nan")
        return z

    with caplog.at_level(logging.ERROR):
        z = pkg.DDIMSampler(model.diffusion, model.unet).sample(shape, cond.to(DEV), 3, DEV, progress=False, noise_fn=nf)
    assert torch.isfinite(z).all()
    text = caplog.text
    assert "NaN/Inf in initial noise z! NaN: 3, Inf: 0" in text
    assert "NaN/Inf in conditioning! NaN: 0, Inf: 1" in text
    caplog.clear()
    with caplog.at_level(logging.ERROR):
        pkg.DDIMSampler(model.diffusion, model.unet).sample(shape, formula_input(shape, 15).to(DEV), 3, DEV,
                                                            progress=False, noise_fn=_noise_fn)
    assert "NaN/Inf" not in caplog.text          # healthy run: silent


def test_sampler_over_a_generic_callable(golden, pkg):
    """The reference's samplers take any model(z, t, c) callable (inference/sampler.py:211-219).  A plain torch function
    goes through the engine's update kernels step by step; wrapping the engine's own U-Net that way must reproduce the
    captured-graph path, and a non-callable is refused."""
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    shape = (1, 8, 4, 8, 8)
    cond = formula_input(shape, 15).to(DEV)
    unet = model.unet
    wrapped = lambda z, t, c: unet(z, t, c)        # not a UNet3D instance: the generic path
    for cls, kw in ((pkg.DDIMSampler, dict(num_inference_steps=5, eta=0.3)), (pkg.DDPMSampler, dict(num_steps=5))):
        a_t, b_t = [], []
        if cls is pkg.DDIMSampler:
            a = cls(model.diffusion, unet).sample(shape, cond, 5, DEV, eta=0.3, progress=False, noise_fn=_noise_fn,
                                                  trajectory=a_t)
            b = cls(model.diffusion, wrapped).sample(shape, cond, 5, DEV, eta=0.3, progress=False, noise_fn=_noise_fn,
                                                     trajectory=b_t)
        else:
            a = cls(model.diffusion, unet).sample(shape, cond, DEV, progress=False, noise_fn=_noise_fn, num_steps=5,
                                                  trajectory=a_t)
            b = cls(model.diffusion, wrapped).sample(shape, cond, DEV, progress=False, noise_fn=_noise_fn, num_steps=5,
                                                     trajectory=b_t)
        assert len(a_t) == len(b_t) and tuple(b.shape) == shape
        assert rel_l2(b.cpu(), a.cpu()) < 1e-5, cls.__name__     # same kernels; only the bf16 input copy path differs
    # a pure-torch epsilon model (no engine inside) against the oracle's sampler on the same callable
    toy = lambda z, t, c: 0.1 * z + 0.05 * c
    got = pkg.DDIMSampler(model.diffusion, toy).sample(shape, cond, 10, DEV, progress=False, noise_fn=_noise_fn)
    bufs = R.diffusion_buffers("cosine", 1000)
    ref = R.ddim_sample(lambda z, t, c: 0.1 * z + 0.05 * c, bufs, shape, cond.cpu(), 10, noise_fn=_noise_fn)
    assert rel_l2(got.cpu(), ref) < 1e-5
    with pytest.raises(pkg.CtsiError, match="callable"):
        pkg.DDIMSampler(model.diffusion, object()).sample(shape, cond, 3, DEV, progress=False)


def test_ddpm_first_steps(golden, pkg):
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    shape = (1, 8, 4, 8, 8)
    traj = []
    pkg.DDPMSampler(model.diffusion, model.unet).sample(shape, formula_input(shape, 15).to(DEV), DEV, progress=False,
                                                        noise_fn=_noise_fn, num_steps=20, trajectory=traj)
    ref = golden["traj.ddpm.first20"]
    errs = [rel_l2(traj[i].cpu(), ref[i]) for i in range(20)]
    print("ddpm per-step rel-L2", ["%.3g" % e for e in errs])
    assert max(errs) < 3e-2


def test_single_reverse_steps_per_sample_t(golden, pkg):
    """GaussianDiffusion.p_mean_variance / p_sample / _predict_z_0_from_noise (models/diffusion.py:249-338) with PER-SAMPLE
    timesteps and both clip settings, against the reference's own outputs.  The U-Net evaluation is the bf16 engine
    (NET_TOL on the mean); variance / log_variance are buffer gathers (bit-exact); the elementwise arithmetic on a given
    epsilon is fp32 (1e-6)."""
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    g = model.diffusion
    shape = (2, 8, 4, 8, 8)
    z, c, eps = formula_input(shape, 23).to(DEV), formula_input(shape, 24).to(DEV), formula_input(shape, 25).to(DEV)
    for tag in ("a", "b"):
        t = torch.tensor(golden[f"pmv.{tag}.t"], device=DEV)
        for clip in (1, 0):
            mean, var, logvar = g.p_mean_variance(model.unet, z, t, c, clip_denoised=bool(clip))
            ref = golden[f"pmv.{tag}.clip{clip}.mean"]
            assert tuple(mean.shape) == shape and mean.dtype == torch.float32
            # t = 999 divides by sqrt(abar) = 4.9e-5: unclipped, the mean IS the z_0 prediction times 1, dominated by that
            # sample; the comparison is relative to the whole tensor, like the reference-under-autocast yardstick
            e = rel_l2(mean.cpu(), ref)
            print(f"p_mean_variance {tag} clip={clip}: rel-L2 {e:.3g}")
            assert e < NET_TOL
            assert np.array_equal(var.cpu().numpy(), golden[f"pmv.{tag}.clip{clip}.var"])
            assert np.array_equal(logvar.cpu().numpy(), golden[f"pmv.{tag}.clip{clip}.logvar"])
            assert tuple(var.shape) == (2, 1, 1, 1, 1)
            zs = g.p_sample(model.unet, z, t, c, clip_denoised=bool(clip), noise=formula_noise(0, shape).to(DEV))
            assert rel_l2(zs.cpu(), golden[f"pmv.{tag}.clip{clip}.p_sample"]) < NET_TOL
        z0 = g._predict_z_0_from_noise(z, t, eps)
        assert rel_l2(z0.cpu(), golden[f"pmv.{tag}.z0_from_noise"]) < 1e-6
    # p_sample: default noise is drawn by torch.randn_like after the network evaluation; batch-uniform clipped steps on
    # the engine's own U-Net take the captured-graph path and agree with the general path given the same noise
    t_u = torch.tensor([500, 500], device=DEV)
    torch.manual_seed(5)
    a = g.p_sample(model.unet, z, t_u, c)
    torch.manual_seed(5)
    nz = torch.randn_like(z)
    b = g.p_sample(model.unet, z, t_u, c, noise=nz)
    assert torch.isfinite(a).all() and rel_l2(a.cpu(), b.cpu()) < 1e-5
    # any model(z, t, c) callable works, as in the reference
    mean_c, _, _ = g.p_mean_variance(lambda zz, tt, cc: eps, z, t_u, c)
    z0c = g._predict_z_0_from_noise(z, t_u, eps).clamp(-1, 1)
    ex = lambda name: g._extract(getattr(g, name), t_u, z.shape)
    assert rel_l2(mean_c.cpu(), (ex("posterior_mean_coef1") * z0c + ex("posterior_mean_coef2") * z).cpu()) < 1e-6
    with pytest.raises(pkg.CtsiError):
        g._predict_z_0_from_noise(z.cpu(), t_u.cpu(), eps.cpu())


def test_legacy163_unet_full_width_vs_reference_golden(golden, pkg):
    """The flat-config 163,410,692-parameter U-Net (heads 8, time_embed_dim 1024, latent 4, three levels of 128 x (1,2,4)):
    full width at low resolution against the REFERENCE's own forward (golden), formula weights."""
    from tests.helpers import LEGACY163_UNET
    un = pkg.UNet3D(**LEGACY163_UNET)
    assert sum(p.numel() for p in un.parameters()) == 163410692
    load_formula(un, 21)
    un.to(DEV)
    x, c = formula_input((1, 4, 6, 16, 16), 31).to(DEV), formula_input((1, 4, 6, 16, 16), 32).to(DEV)
    out = un(x, torch.tensor([321], device=DEV), c)
    e = rel_l2(out.cpu(), golden["unet.legacy163.out"])
    print(f"legacy 163.4 M U-Net (1,4,6,16,16): rel-L2 vs reference golden {e:.3g}")
    assert e < NET_TOL
    un.attention_mode = "exact"
    e2 = rel_l2(un(x, torch.tensor([321], device=DEV), c).cpu(), golden["unet.legacy163.out"])
    print(f"  exact-mode attention: {e2:.3g}")
    assert e2 < NET_TOL
    un.invalidate_engine_cache()


def test_generate_end_to_end(golden, pkg):
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    v_in = formula_input((1, 1, 2, 32, 32), 16).clamp(-1, 1)
    out = model.generate(v_in.to(DEV), 'ddim', num_inference_steps=10, target_depth=12, noise_fn=_noise_fn)
    assert tuple(out.shape) == (1, 1, 12, 32, 32) and out.dtype == torch.float32
    ref = torch.tensor(golden["generate.tiny.out"])
    out_bf = _bf16_autocast_reference(lambda: R.generate(sd, cfg, v_in, "ddim", 10, 12, noise_fn=_noise_fn))
    p_hip, p_bf = R.psnr(out.cpu(), ref, 2.0), R.psnr(out_bf, ref, 2.0)
    print(f"generate PSNR vs reference fp32: hip {p_hip:.2f} dB, reference-under-bf16-autocast {p_bf:.2f} dB")
    assert p_hip >= p_bf - 0.1
    from inference.generate import generate_batch
    outb = generate_batch(model, v_in, 'ddim', 5, DEV, noise_fn=_noise_fn)
    assert tuple(outb.shape) == (1, 1, 2, 32, 32)
    # same relative criterion as generate(): generate_batch = encode -> sample at the input depth -> decode
    # (inference/generate.py:118-155), which is the oracle's generate() with target_depth=None
    refb = torch.tensor(golden["generate_batch.tiny.out"])
    outb_bf = _bf16_autocast_reference(lambda: R.generate(sd, cfg, v_in, "ddim", 5, None, noise_fn=_noise_fn))
    pb_hip, pb_bf = R.psnr(outb.cpu(), refb, 2.0), R.psnr(outb_bf, refb, 2.0)
    print(f"generate_batch PSNR vs reference fp32: hip {pb_hip:.2f} dB, reference-under-bf16-autocast {pb_bf:.2f} dB")
    assert pb_hip >= pb_bf - 0.1
    # default RNG path (no injected noise): runs, finite, deterministic under a fixed seed
    torch.manual_seed(3)
    a = model.generate(v_in.to(DEV), 'ddim', num_inference_steps=3, target_depth=4)
    torch.manual_seed(3)
    b = model.generate(v_in.to(DEV), 'ddim', num_inference_steps=3, target_depth=4)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_full_width_unet_low_resolution(pkg):
    """Effective production U-Net (264.66 M params: 128*(1,2,4,4), heads 4) at a small latent, against the
    oracle run with torch on the GPU in fp32 (same state dict)."""
    torch.manual_seed(0)
    un = pkg.UNet3D(latent_dim=8)
    un.eval().to(DEV)
    sd = {k: v.detach() for k, v in un.state_dict().items()}
    cfg = dict(model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=[1, 2, 4, 4], num_heads=4)
    x = formula_input((1, 8, 12, 16, 16), 50).to(DEV)
    c = formula_input((1, 8, 12, 16, 16), 51).to(DEV)
    t = torch.tensor([400], device=DEV)
    out = un(x, t, c)
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x, t, c)
    assert rel_l2(out.cpu(), ref.cpu()) < NET_TOL


def test_config1_patch_size_unet_matches_oracle_on_gpu(pkg):
    """BASELINE config-1 latent (1, 8, 48, 48, 48) through the full-width net: every tile shape / edge
    case of the real workload.  Oracle = oracle.ref_ops on the same device in fp32."""
    torch.manual_seed(0)
    un = pkg.UNet3D(latent_dim=8)
    un.eval().to(DEV)
    sd = {k: v.detach() for k, v in un.state_dict().items()}
    cfg = dict(model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=[1, 2, 4, 4], num_heads=4)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 8, 48, 48, 48, generator=g).to(DEV)
    c = torch.randn(1, 8, 48, 48, 48, generator=g).to(DEV)
    t = torch.tensor([700], device=DEV)
    out = un(x, t, c)
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x, t, c)
    assert torch.isfinite(out).all()
    assert rel_l2(out.cpu(), ref.cpu()) < NET_TOL


def test_sample_with_stitching(golden, pkg):
    """Sliding-window stitching at depth_ratio 1 (the reference's only working ratio) vs the golden from the
    reference; blend accumulate / normalise kernels vs a direct torch computation."""
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    v_full = formula_input((1, 1, 6, 24, 24), 17).clamp(-1, 1)
    sampler = pkg.DDIMSampler(model.diffusion, model.unet)
    import importlib
    S = importlib.import_module("video-to-video-diffusion_amd.sampler")
    out = S._stitched(sampler, v_full, model.vae, (4, 16, 16), (4, 16, 16), (2, 8, 8), DEV, False,
                      lambda shp, cond: sampler.sample(shp, cond, 3, DEV, progress=False,
                                                       noise_fn=lambda i, s_: formula_noise(-1, s_)))
    ref = torch.tensor(golden["stitch.tiny.out"])
    ref_bf = _bf16_autocast_reference(lambda: R.ddim_stitched(sd, cfg, v_full, 3, (4, 16, 16), (2, 8, 8),
                                                              noise_fn=lambda i, s_: formula_noise(-1, s_)))
    p_hip, p_bf = R.psnr(out.cpu(), ref, 2.0), R.psnr(ref_bf, ref, 2.0)
    print(f"stitching PSNR vs reference fp32: hip {p_hip:.2f} dB, reference under bf16 autocast {p_bf:.2f} dB")
    assert tuple(out.shape) == (1, 1, 6, 24, 24) and p_hip >= p_bf - 0.1
    with pytest.raises(pkg.CtsiError, match="must equal the thick patch"):
        sampler.sample_with_stitching(v_full.to(DEV), model.vae, 3, patch_size=(4, 16, 16),
                                      target_patch_size=(12, 8, 16), stride=(2, 8, 8), device=DEV, progress=False)
    # depth_ratio 3 (the reference raises a shape error here; intended behaviour = per-window depth upsample of the
    # conditioning latent, checked against the oracle's restatement of that intent).  Two bf16 realisations of a chaotic
    # 3-step random-weight pipeline scatter by a few tenths of a dB around each other on ONE noise draw, so the criterion
    # is stated over 8 independent noise seeds: mean(PSNR_hip - PSNR_autocast) >= -0.1 dB.
    deltas = []
    sd_dev = {k: v.to(DEV) for k, v in sd.items()}       # the oracle on this device: fp32, and under bf16 autocast (the
    v_dev = v_full.to(DEV)                               # reference's own AMP path) -- 16 CPU runs would take minutes
    for seed in range(8):
        nf = lambda i, s_, k=seed: formula_noise(100 + 7 * k, s_).to(DEV)
        out3 = S._stitched(sampler, v_full, model.vae, (4, 16, 16), (12, 16, 16), (2, 8, 8), DEV, False,
                           lambda shp, cond: sampler.sample(shp, cond, 3, DEV, progress=False, noise_fn=nf))
        ref3 = R.ddim_stitched(sd_dev, cfg, v_dev, 3, (4, 16, 16), (2, 8, 8), noise_fn=nf, target_d=12)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref3_bf = R.ddim_stitched(sd_dev, cfg, v_dev, 3, (4, 16, 16), (2, 8, 8), noise_fn=nf, target_d=12).float()
        assert tuple(out3.shape) == (1, 1, 18, 24, 24)
        deltas.append(R.psnr(out3, ref3, 2.0) - R.psnr(ref3_bf, ref3, 2.0))
    print("stitching depth_ratio 3, PSNR(hip) - PSNR(oracle under bf16 autocast) over 8 seeds:",
          ["%.2f" % v for v in deltas], "mean %.3f dB" % (sum(deltas) / len(deltas)))
    assert sum(deltas) / len(deltas) >= -0.1
    # the two blend kernels alone, bit-for-bit against torch
    E = importlib.import_module("video-to-video-diffusion_amd.engine")
    ctx = E.Ctx.get(torch.device(DEV))
    acc = torch.zeros(2, 1, 5, 9, 7, device=DEV)
    ws = torch.zeros_like(acc)
    patch = formula_input((2, 1, 3, 4, 5), 3).to(DEV)
    wd, wh, ww = (S._axis_window(n).to(DEV) for n in (3, 4, 5))
    with ctx.scope():
        for (d0, h0, w0) in ((0, 0, 0), (2, 5, 2), (1, 3, 1)):
            ctx.lib.blend_accumulate(E._ptr(acc), E._ptr(ws), E._ptr(patch), E._ptr(wd), E._ptr(wh), E._ptr(ww), 2, 3, 4, 5,
                                     5, 9, 7, d0, h0, w0, ctx.sptr)
        ctx.lib.blend_normalize(E._ptr(acc), E._ptr(ws), acc.numel(), ctx.sptr)
    torch.cuda.synchronize()
    win = S.gaussian_weight(3, 4, 5)
    a2, w2 = torch.zeros(2, 1, 5, 9, 7), torch.zeros(2, 1, 5, 9, 7)
    for (d0, h0, w0) in ((0, 0, 0), (2, 5, 2), (1, 3, 1)):
        a2[:, :, d0:d0 + 3, h0:h0 + 4, w0:w0 + 5] += patch.cpu() * win
        w2[:, :, d0:d0 + 3, h0:h0 + 4, w0:w0 + 5] += win
    assert torch.allclose(acc.cpu(), a2 / (w2 + 1e-8), rtol=1e-6, atol=1e-7)


def test_device_metrics_vs_reference(golden, pkg):
    """utils.metrics drop-in: PSNR / SSIM per frame on device against the values the reference computed."""
    from utils.metrics import calculate_psnr, calculate_ssim, calculate_video_metrics
    a = formula_input((2, 1, 5, 40, 36), 21).clamp(-1, 1)
    b = (a + 0.15 * formula_input((2, 1, 5, 40, 36), 22)).clamp(-1, 1)
    ad, bd = a.to(DEV), b.to(DEV)
    vm = calculate_video_metrics(ad, bd, max_val=2.0)
    assert np.allclose(vm["psnr_per_frame"], golden["metrics.psnr_per_frame"], atol=1e-4)
    assert np.allclose(vm["ssim_per_frame"], golden["metrics.ssim_per_frame"], atol=2e-6)
    assert abs(vm["psnr"] - golden["metrics.mean"][0]) < 1e-4 and abs(vm["ssim"] - golden["metrics.mean"][1]) < 2e-6
    assert abs(calculate_psnr(ad, bd, max_val=2.0) - golden["metrics.psnr_all"][0]) < 1e-4
    assert abs(calculate_ssim(ad, bd, max_val=2.0) - golden["metrics.ssim_5d"][0]) < 2e-6
    assert abs(calculate_ssim(ad[:, :, 0], bd[:, :, 0], window_size=7, max_val=1.0) - golden["metrics.ssim_4d_w7"][0]) < 2e-6
    # identical inputs: the MSE floor of 1e-8 caps the PSNR at 20 log10(max_val / 1e-4) (86.02 dB for max_val 2)
    assert abs(calculate_psnr(ad, ad, max_val=2.0) - golden["metrics.psnr_identical"][0]) < 1e-5
    bad = bd.clone()
    bad[0, 0, 2, 3, 4] = float("nan")
    assert calculate_video_metrics(ad, bad, max_val=2.0) == {'psnr': 0.0, 'ssim': 0.0, 'psnr_per_frame': [], 'ssim_per_frame': []}
    with pytest.raises(pkg.CtsiError):
        calculate_psnr(a, b)
    # (C,T,H,W) input and a volume-sized case against the oracle
    big_a = formula_input((1, 1, 6, 200, 168), 23).clamp(-1, 1)
    big_b = (big_a + 0.05 * formula_input((1, 1, 6, 200, 168), 24)).clamp(-1, 1)
    vm2 = calculate_video_metrics(big_a[0].to(DEV), big_b[0].to(DEV), max_val=2.0)
    assert abs(vm2["ssim"] - R.ssim_box(big_a, big_b, 11, 2.0)) < 2e-6
    assert abs(vm2["psnr"] - np.mean([R.psnr(big_a[:, :, i], big_b[:, :, i], 2.0) for i in range(6)])) < 1e-4


def test_stitching_window_batching_keeps_the_rng_stream(pkg):
    """sample_with_stitching(window_batch=4) == window_batch=1 under the same seed: every window's initial noise is its
    own torch.randn call in window order, and a batched evaluation is per-sample independent."""
    model, sd, cfg = tiny_model_sd(pkg)
    model.to(DEV)
    v_full = formula_input((1, 1, 6, 40, 24), 17).clamp(-1, 1).to(DEV)
    sampler = pkg.DDIMSampler(model.diffusion, model.unet)
    outs = []
    for wb in (1, 4, 3, None):       # None = the default: every window the device memory holds, here all of them at once
        torch.manual_seed(123)
        outs.append(sampler.sample_with_stitching(v_full, model.vae, 3, patch_size=(4, 16, 16),
                                                  target_patch_size=(4, 16, 16), stride=(2, 8, 8), device=DEV,
                                                  progress=False, window_batch=wb).cpu())
    assert tuple(outs[0].shape) == (1, 1, 6, 40, 24)
    for o in outs[1:]:
        assert R.psnr(o, outs[0], 2.0) > 45.0, R.psnr(o, outs[0], 2.0)
    # eta > 0 draws noise inside every step: falls back to one window at a time (same stream as the reference)
    torch.manual_seed(5)
    a = sampler.sample_with_stitching(v_full, model.vae, 3, patch_size=(4, 16, 16), target_patch_size=(4, 16, 16),
                                      stride=(2, 8, 8), device=DEV, eta=0.3, progress=False, window_batch=4)
    torch.manual_seed(5)
    b = sampler.sample_with_stitching(v_full, model.vae, 3, patch_size=(4, 16, 16), target_patch_size=(4, 16, 16),
                                      stride=(2, 8, 8), device=DEV, eta=0.3, progress=False, window_batch=1)
    assert torch.equal(a, b)
