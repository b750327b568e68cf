"""Headline benchmark: DDIM steps/sec (one step = one U-Net evaluation + the x_{t-1} update) on the
8->48-slice @512x512 volume of BASELINE.json config 2, one volume per GPU.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `value` = total steps/s over all ranks with the latent, the
conditioning and every weight resident in HBM when the timed region starts.  The model is the one
`VideoToVideoDiffusion(yaml.safe_load(config/slice_interpolation_full_medium.yaml))` builds in the
reference (264.66 M-param U-Net, latent 8), random-init (seed 0), synthetic input.

Extra objects on the same line:
  roofline     — the dominant kernel (the one with the most time per step; currently the 3x3x3 LDS
                 halo-tile conv): algorithmic FLOPs of its launches in one U-Net evaluation / their
                 summed duration, measured with HIP events on the engine stream around every launch
                 (eager pass, not the graph).  `conv_family` aggregates every MFMA conv launch.
  cpu_baseline — the CPU oracle (fp32 torch ops, every core the process may run on; `host_cores` = os.cpu_count()) timed on a
                 bounded sample of the same workload: ONE U-Net evaluation on the benchmarked latent (1,8,48,128,128) = one
                 step; rank 0, N == 1 only.  --cpu-config1 adds BASELINE config 1 phase by phase (minutes).
  volume_wall_s — wall-clock of the whole 8->48 @512^2 generate() (encode + 51 steps + decode), warm;
                 volume_wall_first_call_s = the cold first call (plans, 0.53 GB weight pack, graph capture).
  config.captured_vs_eager_rel_l2 — the timed hipGraph step replayed once more against the same step launched eagerly from
                 the same state (must be 0); eps_checksum_per_rank: one checksum of that noise prediction per rank.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EFFECTIVE_CFG = {  # what the reference builds from its production YAML (U-Net keys fall back to defaults)
    'model': {'in_channels': 1, 'latent_dim': 8, 'vae_base_channels': 128, 'vae_scaling_factor': 1.0},
    'pretrained': {'use_pretrained': True, 'vae': {'enabled': True, 'checkpoint_path': 'unused'}},
    'noise_schedule': 'cosine', 'diffusion_timesteps': 1000,
}
LEGACY163_CFG = {  # the flat config behind the "163 M-param U-Net" of the reference's README / north_star
    'in_channels': 1, 'latent_dim': 4, 'vae_base_channels': 128, 'unet_model_channels': 128, 'unet_num_res_blocks': 2,
    'unet_attention_levels': [1, 2], 'unet_channel_mult': [1, 2, 4], 'unet_num_heads': 8, 'unet_time_embed_dim': 1024,
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable with a float4 copy)
PMC_TRAFFIC_FILE = os.path.join("profiles", "r04_pmc_traffic.json")   # written by tools/pmc_traffic.py (records its commit)
PMC_TRAFFIC_TRAIN_FILE = os.path.join("profiles", "r04_pmc_traffic_train.json")   # the same over tools/profile_train.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=51)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hw", type=int, default=512, help="slice height/width (512 = config 2)")
    ap.add_argument("--depth-in", type=int, default=8)
    ap.add_argument("--depth-out", type=int, default=48)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1, help="volumes per GPU")
    ap.add_argument("--no-volume", action="store_true", help="skip the end-to-end generate() timing")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle baseline")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads for the CPU oracle leg (0 = every core this process may run on)")
    ap.add_argument("--cpu-config1", action="store_true",
                    help="also time BASELINE config 1 phase by phase on the host cores (encode, one U-Net evaluation, DDIM-10, "
                         "decode at 192x192: minutes of CPU work, so not part of the default run; BASELINE.md section 4)")
    ap.add_argument("--sampler", choices=["ddim", "ddpm"], default="ddim",
                    help="sampler of the end-to-end volume_wall_s leg (ddpm = all 1000 ancestral steps, models/diffusion.py:340-367)")
    ap.add_argument("--model", choices=["effective", "legacy163"], default="effective",
                    help="effective: what the production YAML builds (264.66 M U-Net, the headline); legacy163: the flat "
                         "163.4 M-param variant (latent 4, 3 levels) - secondary figure, skips the CPU leg")
    ap.add_argument("--train-hw", type=int, default=192, help="--mode train: slice height/width (192 = config 3)")
    ap.add_argument("--train-batch", type=int, default=4, help="--mode train: micro-batch per GPU (4 = config 3)")
    ap.add_argument("--mode", choices=["dp", "shard", "train"], default="dp",
                    help="dp: one volume per GPU, no exchange (weak scaling, BASELINE config 5 style); "
                         "shard: ONE volume depth-sharded over the GPUs with RCCL halo exchange (strong scaling, config 4); "
                         "train: forward+backward micro-steps of config 3 (frozen-VAE encode + U-Net fwd/bwd), "
                         "gradients all-reduced over RCCL when N > 1")
    return ap.parse_args()


def _cgroup_cpu_quota():
    """CPUs the container's cgroup may use (cpu.max of cgroup v2, cfs quota of v1), or None when unlimited / unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period) + 0.5))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, int(quota / period + 0.5))
    except Exception:
        pass
    return None


def _host_cores():
    """(cores this process may use, cores of the host).  The GPU boxes hand a container a CPU SHARE of a 256-core host: the
    affinity mask still lists every core, the cgroup quota is what limits the process (256 oracle threads on a 16-CPU share
    took 108 s for the evaluation 16 threads finish in 23 s).  So: affinity mask, cut to the cgroup quota; if neither limits
    anything on a large host, a short probe picks between all cores and 16."""
    total = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = total
    avail = max(1, min(avail, total))
    quota = _cgroup_cpu_quota()
    if quota is not None:
        avail = min(avail, quota)
    elif avail > 32:
        import torch.nn.functional as F
        x = torch.randn(1, 64, 16, 64, 64)
        w = torch.randn(64, 64, 3, 3, 3)
        best = None
        for n in (avail, 16):
            torch.set_num_threads(n)
            with torch.no_grad():
                F.conv3d(x, w, padding=1)
                t0 = time.time()
                for _ in range(3):
                    F.conv3d(x, w, padding=1)
                dt = time.time() - t0
            if best is None or dt < best[0]:
                best = (dt, n)
        avail = best[1]
    return avail, total


def cpu_baseline(threads, latent_hw, depth, config1=False):
    """Oracle U-Net evaluation on the host cores: ONE step's U-Net evaluation of the benchmarked workload itself
    (latent (1,8,depth,hw,hw); 31.4 TFLOP at 512x512 = roughly 15-25 s).  `cores` = threads used = every core this process
    may run on unless --cpu-threads says otherwise; `host_cores` = os.cpu_count().  config1: BASELINE config 1 phase by phase."""
    from oracle import ref_ops as R
    pkg = importlib.import_module("video-to-video-diffusion_amd")
    avail, total = _host_cores()
    cores = avail if threads <= 0 else max(1, min(threads, total))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    un = pkg.UNet3D(latent_dim=8).eval()
    sd = {k: v.detach() for k, v in un.state_dict().items()}
    cfg = dict(model_channels=128, num_res_blocks=2, attention_levels=[1, 2], channel_mult=[1, 2, 4, 4], num_heads=4)
    g = torch.Generator().manual_seed(1)
    shape = (1, 8, depth, latent_hw, latent_hw)
    x = torch.randn(shape, generator=g)
    c = torch.randn(shape, generator=g)
    t = torch.tensor([500])
    with torch.no_grad():
        t0 = time.time()
        R.unet_forward(sd, cfg, x, t, c)
        dt = time.time() - t0
    flops = 31408.6e9 * (depth * latent_hw * latent_hw) / (48 * 128 * 128)
    res = {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "host_cores": total, "cores_available": avail, "kind": "port",
           "sample": f"1 U-Net evaluation (fp32 torch CPU oracle, {cores} threads) on the benchmarked latent {list(shape)}: "
                     f"{dt:.2f} s, {flops / dt / 1e9:.0f} GFLOP/s (the DDIM update itself is negligible)"}
    if config1:
        # config 1 (192x192 patch, 8 -> 48 slices, DDIM-10) phase by phase, seeds as in the survey
        torch.manual_seed(0)
        model = pkg.VideoToVideoDiffusion(EFFECTIVE_CFG).eval()
        msd = {k: v.detach() for k, v in model.state_dict().items()}
        mcfg = dict(cfg, scaling_factor=1.0)
        torch.manual_seed(1)
        v_in = torch.rand(1, 1, 8, 192, 192) * 2 - 1
        ph = {}
        with torch.no_grad():
            t0 = time.time()
            z_in = R.vae_encode(msd, v_in, 1.0, "vae.")
            ph["vae_encode_s"] = time.time() - t0
            z_c = R.trilinear_depth(z_in, 48)
            torch.manual_seed(2)
            zt = torch.randn(z_c.shape)
            t0 = time.time()
            R.unet_forward(msd, mcfg, zt, torch.tensor([500]), z_c, "unet.")
            ph["unet_eval_s"] = time.time() - t0
            bufs = {k[len("diffusion."):]: v for k, v in msd.items() if k.startswith("diffusion.")}
            t0 = time.time()
            z0 = R.ddim_sample(lambda z, tt, cc: R.unet_forward(msd, mcfg, z, tt, cc, "unet."), bufs, tuple(z_c.shape), z_c, 10)
            ph["ddim10_11_evals_s"] = time.time() - t0
            t0 = time.time()
            R.vae_decode(msd, z0, 1.0, "vae.")
            ph["vae_decode_s"] = time.time() - t0
        ph["end_to_end_s"] = ph["vae_encode_s"] + ph["ddim10_11_evals_s"] + ph["vae_decode_s"]
        ph["ddim_steps_per_s"] = 11.0 / ph["ddim10_11_evals_s"]
        res["config1_192_phases"] = ph
    return res


def shard_parity_check(P, E, model, ctx, dev, rank, world, dist, tol=3e-2):
    """Multi-rank parity of the depth-sharded U-Net evaluation against the unsharded engine, per transport, on a small
    RAGGED volume (depth 3*world + 2: slabs of 4 and 3 slices, so the interior/boundary overlap split is on): every rank
    evaluates the unsharded program itself, the sharded epsilon is gathered along depth (padded ragged all-gather) and
    compared on every rank; the worst rank's rel-L2 is reported.  First-contact check for csrc/comm.hip on real xGMI."""
    L = model.vae.latent_dim
    d, h, w = 3 * world + 2, 16, 16
    gen = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((1, L, d, h, w), generator=gen).to(dev)
    c = torch.randn((1, L, d, h, w), generator=gen).to(dev)
    res = {}
    with ctx.scope():
        ref = E.UNetProgram(ctx, model.unet, 1, d, h, w, 2)
        ref.load_latents(x, c)
        ref.set_schedule([500])
        ref.run()
        eps_ref = ref.eps_ncdhw()
        del ref
    for name in ("dist", "rccl"):
        # The verdict is COLLECTIVE: a rank whose transport throws must not skip the all-reduce its peers are blocked in (and
        # must not move on to the next transport's collectives on its own), so the failure is folded into the error value
        # (1e9) and every rank makes the same all_reduce(MAX) on torch.distributed's own group, outside the try.
        err, info, failure = torch.full((1,), 1e9, dtype=torch.float64), {}, None
        try:
            comm = P.DistComm() if name == "dist" else P.RcclComm.from_process_group()
            spec = P.ShardSpec(rank, world, comm, d)
            with ctx.scope():
                pr = E.UNetProgram(ctx, model.unet, 1, spec.depth_local, h, w, 2, shard=spec)
                pr.load_latents(x, c)
                pr.set_schedule([500])
                pr.run()
                full = comm.gather_depth(rank, pr.eps_ncdhw(), counts=spec.depth_counts)
                info = {"depth": d, "slabs": spec.depth_counts,
                        "overlap_split": any(m[0] == "halo.exchange.async" for m in pr.op_meta)}
                del pr
            torch.cuda.synchronize()
            err = ((full.double() - eps_ref.double()).norm() / eps_ref.double().norm()).reshape(1).cpu()
            err = torch.nan_to_num(err, nan=1e9)
        except Exception as exc:   # reported, not fatal -- as long as torch.distributed itself still works (below)
            failure = f"{type(exc).__name__}: {exc}"[:300]
        err = err.to(dev)
        try:
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
        except Exception as exc:   # the process group itself is broken: nothing sensible can follow
            raise SystemExit(f"rank {rank}: torch.distributed all_reduce failed after the '{name}' transport check: {exc}")
        worst = float(err.item())
        res[name] = {"rel_l2_vs_unsharded_max_over_ranks": worst, "ok": bool(worst < tol), **info}
        if failure is not None:
            res[name]["error"] = failure
        elif worst >= 1e9:
            res[name]["error"] = "another rank failed (see its log)"
    return res


def bench_shard(args, pkg, S, E, model, ctx, dev, rank, world, dist):
    """Config 4: one 8->48 @512^2 volume, depth slab of 48/world slices per GPU.  64 sync points per step (RCCL through the
    C ABI, csrc/comm.hip): GroupNorm statistics travel with the boundary slices of the tensor they normalise.  Eager
    launches by default; CTSI_SHARD_CAPTURE=1 captures kernels AND collectives into one hipGraph per step."""
    P = importlib.import_module("video-to-video-diffusion_amd.parallel")
    L = model.vae.latent_dim
    d, h, w = args.depth_out, args.hw // 4, args.hw // 4
    shape = (1, L, d, h, w)
    gen = torch.Generator(device="cpu").manual_seed(1)          # identical volume on every rank
    cond = torch.randn(shape, generator=gen).to(dev)
    z_T = torch.randn(shape, generator=gen).to(dev)
    t_desc = [int(t) for t in pkg.DDIMSampler(model.diffusion, model.unet)._get_timesteps(args.ddim_steps)]
    total = args.warmup + args.steps
    reps = (total + len(t_desc) - 1) // len(t_desc) + 1
    # Transports: "rccl" = RCCL issued by libctsi on the engine stream (C ABI, capture-safe; CTSI_SHARD_CAPTURE=1 replays
    # the step as one hipGraph), "dist" = the same sync points through torch.distributed (nccl backend = RCCL).  On a
    # multi-GPU run BOTH are first checked against the unsharded engine on a small ragged volume (overlap split on); the
    # timed run uses the C-ABI transport only if its check passed (CTSI_SHARD_TRANSPORT overrides), and says which.
    parity, transport = None, "rccl"
    if world > 1:
        parity = shard_parity_check(P, E, model, ctx, dev, rank, world, dist)
        want = os.environ.get("CTSI_SHARD_TRANSPORT")
        transport = want if want in ("rccl", "dist") else ("rccl" if parity["rccl"]["ok"] else "dist")
        comm = P.RcclComm.from_process_group() if transport == "rccl" else P.DistComm()
    else:
        comm = P.RcclComm.single()
    capture = os.environ.get("CTSI_SHARD_CAPTURE") == "1" and transport == "rccl"
    spec = P.ShardSpec(rank, world, comm, d)
    with ctx.scope():
        prog = E.UNetProgram(ctx, model.unet, 1, spec.depth_local, h, w, max_rows=len(t_desc) * reps, shard=spec)
        prog.add_sampler_step("ddim", False)
        prog.load_latents(z_T, cond)
        coef = S.ddim_coef_rows(model.diffusion.alphas_cumprod, t_desc, 0.0).repeat(reps, 1)
        prog.set_schedule([t for _ in range(reps) for t in t_desc], coef.to(dev))
        if capture:
            prog.capture()
            prog.step_ptr.zero_()
        step = prog.launch if capture else prog.run
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ctx.scope():
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    with ctx.scope():
        finite = bool(torch.isfinite(prog.z_ncdhw()).all().item())
    ncomm = sum(1 for m in prog.op_meta if m[2] == "comm")
    if rank == 0:
        print(json.dumps({
            "metric": "ddim_steps_per_sec", "value": args.steps / dt, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"DDIM-{args.ddim_steps} step on ONE latent {list(shape)} depth-sharded "
                                   f"{spec.depth_local} slices/GPU, RCCL: GroupNorm statistics travel with the boundary slices (one ncclGroup per sync point)",
                       "parallelism": f"depth-shard{world}", "sync_points_per_step": ncomm, "captured": capture,
                       "transport": transport, "parity": parity, "finite_outputs": finite},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


def bench_train(args, pkg, E, model, ctx, dev, rank, world, dist):
    """Config 3: v_in (B,1,8,hw,hw), v_gt (B,1,48,hw,hw): frozen-VAE encode of both, U-Net forward + backward
    (models/model.py:158-228).  One timed step = one micro-step (loss.backward() included, no optimizer step, as in
    the reference's gradient-accumulation loop); data parallel over ranks with a bucketed gradient all-reduce."""
    P = importlib.import_module("video-to-video-diffusion_amd.parallel")
    B, hw = args.train_batch, args.train_hw
    gen = torch.Generator(device="cpu").manual_seed(1 + rank)
    v_in = (torch.rand(B, 1, args.depth_in, hw, hw, generator=gen) * 2 - 1).to(dev)
    v_gt = (torch.rand(B, 1, args.depth_out, hw, hw, generator=gen) * 2 - 1).to(dev)
    params = [p for p in model.unet.parameters()]
    for p in model.vae.parameters():
        p.requires_grad_(False)

    def micro_step():
        for p in params:
            p.grad = None
        loss, _ = model(v_in, v_gt)
        loss.backward()
        if world > 1:
            P.allreduce_gradients(params)
        return loss

    for _ in range(max(args.warmup, 1)):
        loss = micro_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = micro_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    finite = bool(torch.isfinite(loss).item()) and all(bool(torch.isfinite(p.grad).all().item()) for p in params[:8])
    prog = [pr for k, pr in model.unet.__dict__["_ctsi_programs"].items() if k[0] == "unet-train"][0]
    # The timed micro-steps do not step the optimizer (gradient accumulation, as in the reference's loop), so the weights
    # never change and are never re-packed.  What an optimizer step adds per accumulation window, measured on its own:
    #   fused  = optim.FusedAdamW.step(): ONE multi-tensor AdamW launch + the engine's fast re-pack of every bf16 kernel image
    #            and fp32 operand of the training program (tables of pointers: csrc/optim.hip, engine.Program.fast_repack)
    #   torch  = torch.optim.AdamW.step() + the generic re-pack through torch (what round 2 measured: 41 + 6 ms)
    def timed(fn):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) * 1e3

    def fresh():
        with ctx.scope():
            prog.ensure_fresh()

    fopt = pkg.FusedAdamW(params, lr=1e-6, engine_modules=[model.unet])
    fopt.step()                       # first calls: build the optimizer's and the program's tables, record the re-pack graph
    fopt.step()
    fused_ms = min(timed(fopt.step) for _ in range(3))
    fused_left_ms = timed(fresh)      # nothing is left to re-pack (this is the fingerprint check every forward makes)
    fopt.engine_modules = []
    fused_only_ms = min(timed(fopt.step) for _ in range(3))     # the optimizer alone (kernel + host bookkeeping)
    fresh()
    del fopt
    opt = torch.optim.AdamW(params, lr=1e-6)
    opt.step()
    adamw_ms = min(timed(opt.step) for _ in range(2))
    repack_ms = timed(fresh)
    groups = {}
    if rank == 0 and not args.no_roofline:
        with ctx.scope():
            prof = prog.profile_ops(repeats=1)
        for i, (name, kern, fl, ms) in enumerate(prof):
            if kern == "conv_wgrad":
                key = "wgrad"
            elif kern.startswith("conv_mfma"):
                key = "dgrad" if name.endswith(".dgrad") else ("fwd_conv" if i < prog.n_fwd else "bwd_conv_other")
            else:
                key = "other_fwd" if i < prog.n_fwd else "other_bwd"
            g = groups.setdefault(key, [0, 0.0, 0.0])
            g[0] += 1
            g[1] += fl
            g[2] += ms
    if rank == 0:
        wg = groups.get("wgrad")
        roof = None
        if wg:
            traffic = None
            traffic_source = None
            try:  # HBM bytes per launch from rocprofv3 --pmc passes over tools/profile_train.py (tools/pmc_traffic.py); null
                pj = json.load(open(os.path.join(ROOT, PMC_TRAFFIC_TRAIN_FILE)))   # when no file of this round exists
                fam = {k_: v for k_, v in pj["kernels"].items() if k_.startswith("conv_wgrad")}
                main = sum(v["launches_FETCH_SIZE"] for k_, v in fam.items() if "reduce" not in k_)
                # bytes of the whole family (the split-K reduce passes included) per weight-gradient kernel launch
                traffic = sum(v["hbm_bytes_per_launch"] * v["launches_FETCH_SIZE"] for v in fam.values()) / main
                traffic_source = {"file": PMC_TRAFFIC_TRAIN_FILE, "commit": pj.get("commit"), "method": pj.get("method"),
                                  "per": "weight-gradient kernel launch, its split-K reduce pass included (family total / "
                                         f"{main} launches in the counter run)"}
            except Exception:
                pass
            ach = wg[1] / (wg[2] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "weight gradient: conv_wgrad_halo_kernel (3x3x3 layers: X halo tile + dY tile in LDS, all 27 "
                                               "taps per block, transposing LDS reads) + conv_wgrad_kernel / conv_wgrad_s1_kernel (the other "
                                               "layers) + their split-K reduce pass",
                    "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": wg[0],
                    "avg_launch_ms": wg[2] / wg[0],
                    "groups": {k: {"launches": v[0], "tflops": (v[1] / (v[2] * 1e-3) / 1e12) if v[1] else None,
                                   "ms": v[2]} for k, v in groups.items()}}
        print(json.dumps({
            "metric": "train_microsteps_per_sec", "value": args.steps * world / dt, "unit": "micro-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"config 3 micro-step: v_in ({B},1,{args.depth_in},{hw},{hw}), v_gt ({B},1,"
                                   f"{args.depth_out},{hw},{hw}); frozen-VAE encode x2 + U-Net forward + backward "
                                   "(264.66M params), every activation kept in HBM (no recomputation)",
                       "micro_batch_per_gpu": B, "parallelism": f"dp{world}", "finite": finite,
                       "samples_per_sec": args.steps * world * B / dt,
                       "unet_fwd_bwd_tflop": prog.flops / 1e12,
                       "after_optimizer_step": {"fused_adamw_step_and_repack_ms": fused_ms,
                                                "fused_adamw_step_alone_ms": fused_only_ms,
                                                "left_to_repack_after_fused_ms": fused_left_ms,
                                                "torch_adamw_step_ms": adamw_ms, "generic_repack_ms": repack_ms,
                                                "note": "not inside the timed micro-steps (no optimizer step there); "
                                                        "paid once per optimizer step, i.e. per accumulation window; "
                                                        "fused = optim.FusedAdamW (one ctsi_adamw_multi launch + fast "
                                                        "re-pack), torch = torch.optim.AdamW + the generic re-pack"}},
            "roofline": roof, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, RCCL) before
        # this process has touched the GPU, and exit with the launcher's code.  (Never exec: a process that has
        # initialised HIP must not be replaced.)
        import socket
        import subprocess
        ngpu = torch.cuda.device_count()
        if ngpu < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} requested but only {ngpu} GPU(s) are visible")
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module("video-to-video-diffusion_amd")
    S = importlib.import_module("video-to-video-diffusion_amd.sampler")
    E = importlib.import_module("video-to-video-diffusion_amd.engine")
    if pkg.get_lib().ablation_build():
        # libctsi_ablate.so (`make ablate`) carries switches that let a kernel skip work: no number is taken from it
        raise SystemExit("bench.py: CTSI_LIB points at an ablation build (timing-only switches compiled in); refusing to time it")

    torch.manual_seed(0)
    model = pkg.VideoToVideoDiffusion(EFFECTIVE_CFG if args.model == "effective" else LEGACY163_CFG).eval().to(dev)
    if args.model != "effective":
        args.no_cpu = True
    n, L = args.batch, model.vae.latent_dim
    d, h, w = args.depth_out, args.hw // 4, args.hw // 4
    shape = (n, L, d, h, w)
    gen = torch.Generator(device="cpu").manual_seed(1 + rank)
    cond = torch.randn(shape, generator=gen).to(dev)
    z_T = torch.randn(shape, generator=gen).to(dev)

    ctx = E.Ctx.get(dev)
    if args.mode == "shard":
        return bench_shard(args, pkg, S, E, model, ctx, dev, rank, world, dist)
    if args.mode == "train":
        return bench_train(args, pkg, E, model, ctx, dev, rank, world, dist)
    sampler = pkg.DDIMSampler(model.diffusion, model.unet)
    t_desc = [int(t) for t in sampler._get_timesteps(args.ddim_steps)]          # 51 entries for 50 steps
    total = args.warmup + args.steps
    reps = (total + len(t_desc) - 1) // len(t_desc) + 1
    with ctx.scope():
        prog = E.UNetProgram(ctx, model.unet, n, d, h, w, max_rows=max(len(t_desc) * reps * n, n), attention_mode="fast")
        prog.add_sampler_step("ddim", False)
        prog.load_latents(z_T, cond)
        coef = S.ddim_coef_rows(model.diffusion.alphas_cumprod, t_desc, 0.0).repeat(reps, 1)
        t_rows = [t for _ in range(reps) for t in t_desc for _ in range(n)]
        prog.set_schedule(t_rows, coef.to(dev))

        roof = None
        if not args.no_roofline:
            prog.launch()  # touch everything once (eager)
            prof = prog.profile_ops(repeats=2)
            variants = {}
            for i_op, (_, k, fl, ms) in enumerate(prof):
                if k.startswith("conv_mfma"):
                    v = variants.setdefault(k, [0, 0.0, 0.0, 0.0])
                    v[0] += 1
                    v[1] += fl
                    v[2] += ms
                    v[3] += prog.op_alg_bytes[i_op] if i_op < len(prog.op_alg_bytes) else 0.0
            step_ms = sum(ms for *_, ms in prof)
            conv_ms = sum(v[2] for v in variants.values())
            conv_fl = sum(v[1] for v in variants.values())
            dom_key, dom = max(variants.items(), key=lambda kv: kv[1][2])   # dominant kernel = most time per step
            kernel_names = {"m9s": "conv3_halo_k32_kernel<SK> (2-way split-K form: two blocks per tile, fp32 partials handed off through a workspace)",
                            "m9t": "conv3_halo_k32_kernel<TR> (ConvTranspose3d (3,4,4)/(1,2,2), 12 entries per chunk, four parity classes)",
                            "m9": "conv3_halo_k32_kernel (3x3x3 LDS halo tile 4x4x32 x 128 couts, 16-ch chunks, tap pairs on mfma_16x16x32_bf16)",
                            "m4": "conv3_halo32_kernel (3x3x3 LDS halo tile 4x2x32, mfma_32x32x16_bf16)",
                            "m3": "conv3_halo_kernel (3x3x3 LDS halo tile 4x4x16, mfma_16x16x32_bf16)"}
            dom_name = kernel_names.get(dom_key.rsplit("_", 1)[-1], "conv_gather_mfma_kernel " + dom_key)
            # PMC counters cannot be collected inside this process: `traffic` is the per-launch HBM byte count of the
            # dominant kernel from the last committed rocprofv3 --pmc passes over the same workload; the file records the
            # commit it was measured at.  null when no such file exists.
            traffic, traffic_source, traffic_cal = None, None, None
            try:
                pj = json.load(open(os.path.join(ROOT, PMC_TRAFFIC_FILE)))
                # the profile lists template instantiations: the bench's variant key names the tile (512x128 = <4, 4, 32, ...>)
                inst = {"conv_mfma_512x128_m9": "conv3_halo_k32_kernel<4, 4, 32, 128, 2, false",
                        "conv_mfma_512x128_m9t": "conv3_halo_k32_kernel<4, 4, 32, 128, 2, true",
                        "conv_mfma_384x128_m9": "conv3_halo_k32_kernel<3, 4, 32, 128, 2, false"}
                key = inst.get(dom_key, dom_name.split(" ")[0])
                ent = next(v for k_, v in pj["kernels"].items() if k_.startswith(key))
                traffic = ent["hbm_bytes_per_launch"]
                traffic_cal = ent.get("hbm_bytes_calibrated")
                traffic_source = {"file": PMC_TRAFFIC_FILE, "commit": pj.get("commit"), "method": pj.get("method"),
                                  "fetch_factor_calibrated": ent.get("fetch_factor_calibrated"),
                                  "FETCH_SIZE_bytes_per_launch": ent["FETCH_SIZE_KB_avg_per_launch"] * 1024.0,
                                  "WRITE_SIZE_bytes_per_launch": ent["WRITE_SIZE_KB_avg_per_launch"] * 1024.0,
                                  "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes.  FETCH_SIZE counts "
                                          "the L2's fabric-side read requests at 64 B each (Infinity-Cache hits included): for this "
                                          "kernel's halo DMA (32-byte pieces of 256-byte rows, one 16-channel chunk per pass) our "
                                          "calibration (profiles/r03_fetch_calibration.json) reads between 2 x (every piece its own "
                                          "request) and 4 x (one tallied request per row), so the read part lies between the figure "
                                          "given and twice that; it is fabric traffic, not necessarily HBM traffic"}
            except Exception:
                pass
            # HBM-bound kernels: algorithmic bytes / launch time (same HIP-event pass)
            hbm = {}
            for i, (nm, _, _, ms) in enumerate(prof):
                nb = prog.op_bytes[i] if i < len(prog.op_bytes) else 0.0
                if nb > 0:
                    hk = hbm.setdefault(nm, [0, 0.0, 0.0])
                    hk[0] += 1
                    hk[1] += nb
                    hk[2] += ms
            hbm_kernels = {k_: {"launches": v[0], "ms": v[2], "gbytes": v[1] / 1e9, "GB_per_s": v[1] / (v[2] * 1e-3) / 1e9,
                                "frac_of_peak": v[1] / (v[2] * 1e-3) / 1e9 / PEAK_HBM_GBS}
                           for k_, v in sorted(hbm.items(), key=lambda kv: -kv[1][2])}
            ach = dom[1] / (dom[2] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom_name, "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                    "traffic_source": traffic_source,
                    "launches_per_step": dom[0], "avg_launch_ms": dom[2] / dom[0],
                    "flops_per_launch_avg": dom[1] / dom[0], "share_of_step_time": dom[2] / step_ms,
                    "algorithmic_bytes": dom[3] / dom[0],
                    "traffic_over_algorithmic": (traffic / (dom[3] / dom[0])) if (traffic and dom[3]) else None,
                    # the same with the FETCH_SIZE factor calibrated for this kernel's access shape (tools/pmc_traffic.py): an upper bound
                    "traffic_calibrated": traffic_cal,
                    "traffic_calibrated_over_algorithmic": (traffic_cal / (dom[3] / dom[0])) if (traffic_cal and dom[3]) else None,
                    "conv_family": {"tflops": conv_fl / (conv_ms * 1e-3) / 1e12,
                                    "frac": conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                    "launches_per_step": sum(v[0] for v in variants.values()),
                                    "flops_per_step": conv_fl, "share_of_step_time": conv_ms / step_ms},
                    "other_ops_ms": step_ms - conv_ms,
                    "hbm_bound_kernels": {"peak_GB_per_s": PEAK_HBM_GBS, "kernels": hbm_kernels},
                    "variants": {k: {"launches": v[0], "tflops": v[1] / (v[2] * 1e-3) / 1e12, "ms": v[2]}
                                 for k, v in sorted(variants.items(), key=lambda kv: -kv[1][2])}}
            prog.load_latents(z_T, cond)
            prog.set_schedule(t_rows, coef.to(dev))

        prog.capture()
        prog.step_ptr.zero_()
        for _ in range(args.warmup):
            prog.launch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ctx.scope():
        for _ in range(args.steps):
            prog.launch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    with ctx.scope():
        finite = bool(torch.isfinite(prog.z_ncdhw()).all().item())
        # What the timed graph computes, checked beyond "finite": one more replay of the captured step against the same
        # step launched eagerly from the same state (z, conditioning, step counter): identical kernels in identical order, so
        # the noise predictions must agree to the last bit.
        z_keep, xin_keep, sp_keep = prog.z.clone(), prog.xin.t.clone(), prog.step_ptr.clone()
        prog.launch()
        eps_graph = prog.eps.clone()
        prog.z.copy_(z_keep)
        prog.xin.t.copy_(xin_keep)
        prog.step_ptr.copy_(sp_keep)
        prog.run()
        eps_eager = prog.eps.clone()
        cap_vs_eager = float(((eps_graph.double() - eps_eager.double()).norm() / eps_eager.double().norm().clamp_min(1e-30)).item())
        finite = finite and bool(torch.isfinite(eps_graph).all().item()) and float(eps_graph.abs().sum().item()) > 0.0
        checksum = eps_graph.double().sum().reshape(1)
    rank_checksums = None
    if dist is not None:      # data parallel, N > 1: one checksum of the captured step's noise prediction per rank
        allc = [torch.zeros_like(checksum) for _ in range(world)]
        dist.all_gather(allc, checksum)
        rank_checksums = [float(c.item()) for c in allc]
        fin = torch.tensor([1.0 if (finite and cap_vs_eager == 0.0) else 0.0], device=dev)
        dist.all_reduce(fin, op=dist.ReduceOp.MIN)
        finite = bool(fin.item() > 0.5)
    else:
        rank_checksums = [float(checksum.item())]

    unet_flops = prog.flops
    del prog
    model.unet.__dict__.pop("_ctsi_programs", None)
    torch.cuda.empty_cache()

    volume_wall, volume_wall_first, vae_legs = None, None, None
    if not args.no_volume and world == 1:
        v_in = (torch.rand(n, 1, args.depth_in, args.hw, args.hw, generator=gen) * 2 - 1).to(dev)
        model.invalidate_engine_cache()      # the first call below is a COLD one: plans, weight packing (0.53 GB), graph capture
        for rep in range(2):  # first pass builds/captures programs (reported as volume_wall_first_call_s), second is the measurement
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            v_out = model.generate(v_in, args.sampler, num_inference_steps=args.ddim_steps, target_depth=args.depth_out)
            torch.cuda.synchronize()
            volume_wall = time.perf_counter() - t1
            if rep == 0:
                volume_wall_first = volume_wall
        finite = finite and bool(torch.isfinite(v_out).all().item()) and tuple(v_out.shape) == (n, 1, args.depth_out,
                                                                                                 args.hw, args.hw)
        # the two frozen-VAE legs of the volume on their own (programs are cached by the generate() calls above)
        vae_legs = {}
        z_lat = torch.randn(shape, generator=gen).to(dev)
        for leg, fn, arg in (("encode", model.vae.encode, v_in), ("decode", model.vae.decode, z_lat)):
            fn(arg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn(arg)
            torch.cuda.synchronize()
            leg_s = time.perf_counter() - t1
            progs = [pr for k_, pr in model.vae.__dict__.get("_ctsi_programs", {}).items() if k_[0] == leg[:3]]
            fl = progs[-1].flops if progs else 0.0
            vae_legs[leg] = {"ms": leg_s * 1e3, "tflop": fl / 1e12, "tflops": fl / leg_s / 1e12,
                             "frac": fl / leg_s / 1e12 / PEAK_BF16_TFLOPS}

    if rank == 0:
        steps_total = args.steps * world * n
        res = {
            "metric": "ddim_steps_per_sec", "value": steps_total / dt, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"DDIM-{args.ddim_steps} step (U-Net eval + update) on latent {list(shape)} = "
                                   f"{args.depth_in}->{args.depth_out} slices @{args.hw}x{args.hw}, "
                                   + ("effective 264.66M U-Net (128x(1,2,4,4), latent 8)" if args.model == "effective"
                                      else "legacy 163.4M U-Net (128x(1,2,4), latent 4)")
                                   + ", one volume per GPU, hipGraph-captured step",
                       "volumes_per_gpu": n, "parallelism": f"dp{world}", "finite_outputs": finite,
                       "captured_vs_eager_rel_l2": cap_vs_eager, "eps_checksum_per_rank": rank_checksums,
                       "unet_tflop_per_step": unet_flops / 1e12,
                       "unet_tflops_achieved_per_gpu": unet_flops * args.steps / dt / 1e12},
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu or world > 1) else cpu_baseline(args.cpu_threads, args.hw // 4,
                                                                                  args.depth_out, args.cpu_config1),
            "volume_wall_s": volume_wall, "volume_wall_first_call_s": volume_wall_first,
            "volume_sampler": (f"ddim-{args.ddim_steps}" if args.sampler == "ddim" else "ddpm-1000"),
            "vae": vae_legs if (not args.no_volume and world == 1) else None,
        }
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
