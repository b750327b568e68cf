"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

Imports the real reference modules, loads formula-initialised weights (oracle.ref_ops.formula_state_dict,
reproducible anywhere), runs the reference's own code and stores inputs' recipe + expected outputs in
tests/golden/golden_v1.npz.  The reference's source never enters this repo: only numeric vectors do.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.dont_write_bytecode = True
# the reference's packages are called `models` / `inference` like this repo's drop-ins: make sure the
# reference wins and the repo root is NOT importable as a package root here
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REF)

spec = importlib.util.spec_from_file_location("ref_ops", os.path.join(REPO, "oracle", "ref_ops.py"))
R = importlib.util.module_from_spec(spec)
spec.loader.exec_module(R)

import models.unet3d as ref_unet            # noqa: E402
import models.vae as ref_vae                # noqa: E402
import models.diffusion as ref_diff         # noqa: E402
from models import VideoToVideoDiffusion    # noqa: E402
from inference.sampler import DDIMSampler, DDPMSampler  # noqa: E402
import utils.metrics as ref_metrics         # noqa: E402

assert ref_unet.__file__.startswith(REF), ref_unet.__file__
torch.set_num_threads(8)
out = {}


formula_input = R.formula_input
formula_noise = R.formula_noise


def load_formula(module, seed=0):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = R.formula_state_dict(shapes, seed)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


with torch.no_grad():
    # 1. schedules and timestep subsets ----------------------------------------------------------------
    for sched in ("cosine", "linear"):
        g = ref_diff.GaussianDiffusion(noise_schedule=sched, timesteps=1000)
        for name, buf in g.named_buffers():
            out[f"sched.{sched}.{name}"] = buf.numpy().copy()
    samp = DDIMSampler(ref_diff.GaussianDiffusion(), None)
    for n in (3, 7, 10, 20, 50, 100, 1000):
        out[f"timesteps.{n}"] = np.asarray(samp._get_timesteps(n)).astype(np.int64).copy()

    # 2. per-op / per-block -----------------------------------------------------------------------------
    te = ref_unet.TimeEmbedding(128, 512)
    load_formula(te, 1)
    out["op.time_embed.t"] = np.array([0, 1, 500, 999], dtype=np.int64)
    out["op.time_embed.out"] = te(torch.tensor([0, 1, 500, 999])).numpy()

    rb = ref_unet.ResBlock3D(16, 32, 64)
    load_formula(rb, 2)
    x = formula_input((2, 16, 3, 6, 5), 1)
    temb = formula_input((2, 64), 2)
    out["op.resblock.out"] = rb(x, temb).numpy()

    rb2 = ref_unet.ResBlock3D(32, 32, 64)
    load_formula(rb2, 3)
    out["op.resblock_same.out"] = rb2(formula_input((1, 32, 4, 5, 6), 3), formula_input((1, 64), 4)).numpy()

    at = ref_unet.TemporalAttention(64, 4)
    load_formula(at, 4)
    out["op.attn.out"] = at(formula_input((2, 64, 6, 5, 4), 5)).numpy()

    at2 = ref_unet.TemporalAttention(256, 4)
    load_formula(at2, 5)
    xa = formula_input((1, 256, 5, 3, 3), 6)
    out["op.attn256.out"] = at2(xa).numpy()

    dn = ref_unet.Downsample3D(16)
    load_formula(dn, 6)
    out["op.down.out"] = dn(formula_input((1, 16, 3, 8, 6), 7)).numpy()
    up = ref_unet.Upsample3D(16)
    load_formula(up, 7)
    out["op.up.out"] = up(formula_input((1, 16, 3, 4, 5), 8)).numpy()

    for (din, dout) in ((8, 48), (2, 12), (5, 7)):
        z = formula_input((1, 3, din, 4, 5), 9)
        out[f"op.trilinear.{din}_{dout}"] = torch.nn.functional.interpolate(
            z, size=(dout, 4, 5), mode="trilinear", align_corners=False).numpy()

    # 3. tiny U-Net forward -------------------------------------------------------------------------------
    tiny = dict(latent_dim=8, model_channels=32, num_res_blocks=1, attention_levels=[1], channel_mult=(1, 2),
                num_heads=4, time_embed_dim=64)
    un = ref_unet.UNet3D(**tiny)
    load_formula(un, 8)
    zx, zc = formula_input((2, 8, 4, 8, 8), 10), formula_input((2, 8, 4, 8, 8), 11)
    out["unet.tiny.out"] = un(zx, torch.tensor([500, 37]), zc).numpy()

    # three-level net with odd spatial sizes and attention at two levels
    mid = dict(latent_dim=4, model_channels=32, num_res_blocks=2, attention_levels=[1, 2], channel_mult=(1, 2, 4),
               num_heads=8, time_embed_dim=128)
    un3 = ref_unet.UNet3D(**mid)
    load_formula(un3, 9)
    zx3, zc3 = formula_input((1, 4, 6, 12, 8), 12), formula_input((1, 4, 6, 12, 8), 13)
    out["unet.mid.out"] = un3(zx3, torch.tensor([999]), zc3).numpy()

    # 3b. the flat-config "163 M" U-Net named by the README / north_star at FULL WIDTH, low resolution:
    #     128 x (1,2,4), heads 8, time_embed_dim 1024, latent 4 -- 163,410,692 parameters, formula-initialised
    leg_kw = dict(latent_dim=4, model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=(1, 2, 4),
                  num_heads=8, time_embed_dim=1024)
    unl = ref_unet.UNet3D(**leg_kw)
    assert sum(p_.numel() for p_ in unl.parameters()) == 163410692
    load_formula(unl, 21)
    zxl, zcl = formula_input((1, 4, 6, 16, 16), 31), formula_input((1, 4, 6, 16, 16), 32)
    out["unet.legacy163.out"] = unl(zxl, torch.tensor([321]), zcl).numpy()
    del unl

    # 4. tiny VAE -------------------------------------------------------------------------------------------
    vae = ref_vae.SliceInterpolationVAE(in_channels=1, latent_dim=8, base_channels=16, scaling_factor=0.5)
    load_formula(vae, 10)
    vx = formula_input((1, 1, 3, 16, 12), 14)
    zlat = vae.encode(vx)
    out["vae.tiny.latent"] = zlat.numpy()
    out["vae.tiny.recon"] = vae.decode(zlat).numpy()

    # 5. sampler trajectories with injected noise -----------------------------------------------------------
    cfg = {'in_channels': 1, 'latent_dim': 8, 'vae_base_channels': 16, 'vae_scaling_factor': 1.0,
           'unet_model_channels': 32, 'unet_num_res_blocks': 1, 'unet_attention_levels': [1],
           'unet_channel_mult': [1, 2], 'unet_num_heads': 4, 'unet_time_embed_dim': 64,
           'noise_schedule': 'cosine', 'diffusion_timesteps': 1000}
    model = VideoToVideoDiffusion(cfg)
    load_formula(model, 11)
    # formula_state_dict also overwrote the diffusion buffers: restore the real schedule
    model.diffusion = ref_diff.GaussianDiffusion('cosine', 1000)
    model.eval()
    shape = (1, 8, 4, 8, 8)
    cond = formula_input(shape, 15)

    class Injected:
        """torch.randn / randn_like replacement handing out formula noise in call order."""

        def __init__(self, first=-1):
            self.i = first

        def randn(self, *size, **kw):
            shp = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            v = formula_noise(self.i, shp)
            self.i += 1
            return v

        def randn_like(self, t, **kw):
            return self.randn(tuple(t.shape))

    def with_injected(fn, first=-1):
        inj = Injected(first)
        orig = torch.randn, torch.randn_like
        torch.randn, torch.randn_like = inj.randn, inj.randn_like
        try:
            return fn()
        finally:
            torch.randn, torch.randn_like = orig

    def record_traj(sampler_call):
        traj = []
        orig_forward = model.unet.forward

        def spy(z, t, c):
            traj.append(z.clone())
            return orig_forward(z, t, c)

        model.unet.forward = spy
        try:
            z_final = sampler_call()
        finally:
            model.unet.forward = orig_forward
        return traj[1:] + [z_final]   # z after every update

    for eta in (0.0, 0.5):
        traj = with_injected(lambda: record_traj(
            lambda: DDIMSampler(model.diffusion, model.unet).sample(shape, cond, 10, 'cpu', eta=eta, progress=False)))
        out[f"traj.ddim.eta{eta}"] = torch.stack(traj).numpy()

    # DDPM: first 20 steps of the 1000-step loop
    class Stop(Exception):
        pass

    traj = []
    orig_forward = model.unet.forward

    def spy20(z, t, c):
        traj.append(z.clone())
        if len(traj) == 21:
            raise Stop
        return orig_forward(z, t, c)

    model.unet.forward = spy20
    try:
        with_injected(lambda: DDPMSampler(model.diffusion, model.unet).sample(shape, cond, 'cpu', progress=False))
    except Stop:
        pass
    model.unet.forward = orig_forward
    out["traj.ddpm.first20"] = torch.stack(traj[1:]).numpy()

    # 5b. single reverse steps with PER-SAMPLE timesteps (diffusion.py:249-338): p_mean_variance with and without the
    #     clip, _predict_z_0_from_noise, p_sample (its randn_like injected as formula noise index 0)
    shape2 = (2, 8, 4, 8, 8)
    z_pm, c_pm = formula_input(shape2, 23), formula_input(shape2, 24)
    for tag, tv in (("a", [500, 37]), ("b", [0, 999])):
        t_pm = torch.tensor(tv)
        out[f"pmv.{tag}.t"] = np.array(tv, dtype=np.int64)
        for clip in (True, False):
            mean, var, logvar = model.diffusion.p_mean_variance(model.unet, z_pm, t_pm, c_pm, clip_denoised=clip)
            out[f"pmv.{tag}.clip{int(clip)}.mean"] = mean.numpy()
            out[f"pmv.{tag}.clip{int(clip)}.var"] = var.numpy()
            out[f"pmv.{tag}.clip{int(clip)}.logvar"] = logvar.numpy()
            zs = with_injected(lambda: model.diffusion.p_sample(model.unet, z_pm, t_pm, c_pm, clip_denoised=clip), first=0)
            out[f"pmv.{tag}.clip{int(clip)}.p_sample"] = zs.numpy()
        eps_pm = formula_input(shape2, 25)
        out[f"pmv.{tag}.z0_from_noise"] = model.diffusion._predict_z_0_from_noise(z_pm, t_pm, eps_pm).numpy()

    # 6. generate() end to end on the tiny config ---------------------------------------------------------------
    v_in = formula_input((1, 1, 2, 32, 32), 16).clamp(-1, 1)
    # generate draws one discarded randn (model.py:303) before the sampler's own: start the injected
    # sequence at -2 so the sampler's initial draw is index -1, as in the engine's noise_fn contract
    v_out = with_injected(lambda: model.generate(v_in, 'ddim', num_inference_steps=10, target_depth=12), first=-2)
    out["generate.tiny.out"] = v_out.numpy()
    m = ref_metrics.calculate_psnr(v_out, torch.zeros_like(v_out), max_val=2.0)
    out["generate.tiny.psnr_vs_zero"] = np.array([m], dtype=np.float64)

    # generate_batch equivalent (generate.py:118-155): encode -> DDIM at input depth -> decode
    def gb():
        z_in = model.vae.encode(v_in)
        z0 = DDIMSampler(model.diffusion, model.unet).sample(z_in.shape, z_in, 5, 'cpu', progress=False)
        return model.vae.decode(z0)
    out["generate_batch.tiny.out"] = with_injected(gb).numpy()

# 6b. sliding-window stitching (sampler.py:338-453) at depth_ratio == 1 (the only ratio that does not crash
#     in the reference): every patch's initial draw is the same injected noise (stateless injector)
with torch.no_grad():
    v_full = formula_input((1, 1, 6, 24, 24), 17).clamp(-1, 1)
    orig = torch.randn, torch.randn_like
    torch.randn = lambda *size, **kw: formula_noise(-1, tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size))
    torch.randn_like = lambda t, **kw: formula_noise(-1, tuple(t.shape))
    try:
        st = DDIMSampler(model.diffusion, model.unet).sample_with_stitching(
            v_full, model.vae, num_inference_steps=3, patch_size=(4, 16, 16), target_patch_size=(4, 16, 16),
            stride=(2, 8, 8), device='cpu', progress=False)
    finally:
        torch.randn, torch.randn_like = orig
    out["stitch.tiny.out"] = st.numpy()
    out["stitch.gauss_4_16_16"] = DDIMSampler(model.diffusion, model.unet)._create_gaussian_weight(4, 16, 16).numpy()

# 6c. training forward + backward (model.py:158-228 -> diffusion.py:108-203 + autograd): injected t / noise,
#     full-volume mode (thick depth 2 -> thin depth 6), without a mask and with a variable-depth mask.
#     Stored: the losses, the L2 norm of every U-Net parameter gradient, and the gradients themselves for the
#     parameters with <= 30k elements plus one large conv weight.
v_tr_in = formula_input((2, 1, 2, 32, 32), 18).clamp(-1, 1)
v_tr_gt = formula_input((2, 1, 6, 32, 32), 19).clamp(-1, 1)
t_fix = torch.tensor([37, 812])
mask_var = torch.tensor([[[1., 1., 1., 1., 1., 1.]], [[1., 1., 1., 1., 0., 0.]]])   # (B, 1, T_gt)
orig_ri, orig_rl = torch.randint, torch.randn_like
torch.randint = lambda *a, **kw: t_fix.clone()
torch.randn_like = lambda tt, **kw: formula_noise(-1, tuple(tt.shape))
try:
    for tag, mk in (("nomask", None), ("mask", mask_var)):
        for p_ in model.parameters():
            p_.grad = None
            p_.requires_grad_(True)
        loss, metrics = model(v_tr_in, v_tr_gt, mask=mk)
        loss.backward()
        out[f"train.{tag}.loss"] = np.array([loss.item()], dtype=np.float64)
        names_u, norms = [], []
        for name_, p_ in model.unet.named_parameters():
            names_u.append(name_)
            norms.append(float(p_.grad.double().norm()))
            if tag == "nomask" and (p_.numel() <= 30000 or name_ == "mid_block1.conv1.conv.weight"):
                out[f"train.nomask.grad.{name_}"] = p_.grad.numpy().copy()
        out[f"train.{tag}.grad_norms"] = np.array(norms, dtype=np.float64)
        out["train.param_names"] = np.array(names_u)
        assert all(p_.grad is None for p_ in model.vae.parameters()), "VAE must stay frozen (no_grad encode)"
finally:
    torch.randint, torch.randn_like = orig_ri, orig_rl
    for p_ in model.parameters():
        p_.grad = None

# 6d. validation metrics (utils/metrics.py): PSNR / box-window SSIM per frame and overall
m_a = formula_input((2, 1, 5, 40, 36), 21).clamp(-1, 1)
m_b = (m_a + 0.15 * formula_input((2, 1, 5, 40, 36), 22)).clamp(-1, 1)
vm = ref_metrics.calculate_video_metrics(m_a, m_b, max_val=2.0)
out["metrics.psnr_per_frame"] = np.array(vm["psnr_per_frame"], dtype=np.float64)
out["metrics.ssim_per_frame"] = np.array(vm["ssim_per_frame"], dtype=np.float64)
out["metrics.mean"] = np.array([vm["psnr"], vm["ssim"]], dtype=np.float64)
out["metrics.psnr_all"] = np.array([ref_metrics.calculate_psnr(m_a, m_b, max_val=2.0)], dtype=np.float64)
out["metrics.ssim_5d"] = np.array([ref_metrics.calculate_ssim(m_a, m_b, max_val=2.0)], dtype=np.float64)
out["metrics.ssim_4d_w7"] = np.array([ref_metrics.calculate_ssim(m_a[:, :, 0], m_b[:, :, 0], window_size=7, max_val=1.0)],
                                     dtype=np.float64)
out["metrics.psnr_identical"] = np.array([ref_metrics.calculate_psnr(m_a, m_a, max_val=2.0)], dtype=np.float64)

# 7. state-dict layout of the effective production model (names + shapes only, built on `meta`) ----------
import yaml  # noqa: E402
cfg_full = yaml.safe_load(open(os.path.join(REF, "config", "slice_interpolation_full_medium.yaml")))
with torch.device("meta"):
    full = VideoToVideoDiffusion(cfg_full)
names = list(full.state_dict().keys())
out["statedict.effective.names"] = np.array(names)
out["statedict.effective.shapes"] = np.array([",".join(str(d) for d in v.shape) for v in full.state_dict().values()])
cnt = full.count_parameters()
out["statedict.effective.counts"] = np.array([cnt["total"], cnt["vae"], cnt["unet"]], dtype=np.int64)
legacy = {'in_channels': 1, 'latent_dim': 4, 'vae_base_channels': 128, 'unet_model_channels': 128,
          'unet_num_res_blocks': 2, 'unet_attention_levels': [1, 2], 'unet_channel_mult': [1, 2, 4],
          'unet_num_heads': 8, 'unet_time_embed_dim': 1024}
with torch.device("meta"):
    leg = VideoToVideoDiffusion(legacy)
out["statedict.legacy163.unet_params"] = np.array([leg.count_parameters()["unet"]], dtype=np.int64)

path = os.path.join(REPO, "tests", "golden", "golden_v1.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
