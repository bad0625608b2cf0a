#!/usr/bin/env python3
"""bench.py -- throughput of the CLD latent-diffusion sampling path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic agents: the full ancestral
sampling loop (100 U-Net evaluations + DDPM updates, reference models/dm/dm_model.py:103-142)
followed by LSTM decode + unicycle roll-out (guide_dm_trainer.py:97-98) and, for N > 1, the
RCCL all-gather of the decoded trajectories at the rollout-step boundary.

Workload at N = 1: BASELINE.json configs[1] -- 32 scenes x 32 agents (B = 1,024), 100 denoising
steps, d = 256, seq 52, CFG off.  N > 1: weak scaling, every rank samples its own 32 x 32 shard
(scenes are independent: no collective inside the loop).

Prints ONE JSON line (rank 0): metric = denoising-step.agent/s over the whole job, plus
`roofline` (dominant kernel vs. the fp32-MFMA peak, timed with HIP events inside the library)
and `cpu_baseline` (the oracle on the host cores, bounded sample; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_STEP_AGENT = 119_232_512          # U-Net forward, SURVEY 8(d) / BASELINE.md section 2
PEAK_F32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.6              # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_f16 dense peak (f16x2 mode issues 3 MFMAs per product)
FLOP_PER_CONTEXT_AGENT = 2 * 3_031_000_000   # ResNet-18 on [34,224,224] (stem 1.337 GMAC) + fc + MLPs, SURVEY 8(f-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scenes", type=int, default=32, help="scenes per GPU")
    ap.add_argument("--agents", type=int, default=32, help="agents per scene")
    ap.add_argument("--denoise-steps", type=int, default=100)
    ap.add_argument("--cfg-w", type=float, default=0.0,
                    help="classifier-free guidance weight (BASELINE configs[2]: --agents 64 --cfg-w 2.0); 0 = off")
    ap.add_argument("--closed-loop", type=int, default=0, metavar="SIM_STEPS",
                    help="BASELINE configs[4]-style step: SIM_STEPS x (sample -> decode -> kinematic update -> gather); "
                         "e.g. --closed-loop 20 --denoise-steps 50 --scenes 64 --agents 64")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default="f32",
                    help="conv arithmetic: exact fp32 MFMA, or fp16 hi/lo split operands with fp32 accumulation (include/cld.h)")
    ap.add_argument("--guide", action="store_true",
                    help="sampling-time guidance on every step t > 0 (target-speed loss through decoder + roll-out, Adam lr 0.3; "
                         "BASELINE configs[2]: --agents 64 --cfg-w 2.0 --guide)")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the ContextEncoder (producer of cond_feat, SURVEY 8(f-1)) measurement / closed-loop stage")
    ap.add_argument("--no-alt-precision", action="store_true",
                    help="skip the secondary measurement of the same workload in the optional f16x2 split-precision mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event timing of the dominant kernel")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare with --gpus N: launch the N ranks as a CHILD torch.distributed.run (nothing has touched the GPU yet,
        # and a child process rather than exec keeps this safe under a profiler's preloaded runtime) and leave with its code
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # "nccl" is RCCL on ROCm.  CLD_DIST_BACKEND=gloo only exists to rehearse the multi-process flow on a box
        # with fewer GPUs than ranks (ranks then share devices; the gather stages through the host).
        backend = os.environ.get("CLD_DIST_BACKEND", "nccl")
        local_dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_dev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend=backend)
    else:
        local_dev = local_rank % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)

    from cld_amd import synth
    from cld_amd.engine import Engine
    from cld_amd.parallel import gather_trajectories

    n = args.denoise_steps
    B = args.scenes * args.agents
    eng = Engine(n_timesteps=n, device=dev, precision=args.precision)
    eng.load_state_dict(synth.make_unet_weights(0))            # PyTorch-default-like random init (no checkpoint ships)
    eng.load_state_dict(synth.make_decoder_weights(0))
    has_encoder = bool(args.closed_loop)
    if has_encoder:
        eng.load_state_dict(synth.make_encoder_weights(0))
    if not args.no_context:
        eng.load_state_dict(synth.make_context_weights(0))
    eng.finalize()

    # synthetic inputs, resident in HBM before the timed region starts
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    cond = torch.randn(B, 256, device=dev, generator=g)
    x_T = torch.randn(B, 52, 4, device=dev, generator=g)
    noise = torch.randn(n, B, 52, 4, device=dev, generator=g)
    non_cond = torch.randn(B, 256, device=dev, generator=g) if args.cfg_w != 0.0 else None
    cs = torch.zeros(B, 4, device=dev)
    cs[:, 2] = torch.rand(B, device=dev, generator=g) * 15.0
    gathered = torch.empty(world * B, 52, 6, device=dev) if distributed else None

    world0 = torch.zeros(B, 3, device=dev)
    guidance = None
    if args.guide:        # upstream defaults: adam, lr 0.3, one gradient step per denoising step (scene_edit_config.py:74-90)
        guidance = {"curr_states": cs, "target_speed": torch.rand(B, 52, device=dev, generator=g) * 12.0,
                    "loss_scale": torch.full((B,), 1.0 / (args.agents * 52), device=dev), "lr": 0.3, "optimizer": "adam"}

    use_ctx = not args.no_context
    if use_ctx:
        # synthetic raster with the reference's structure (trajdata_utils.py:123-156,409-420): 31 history planes that are
        # zero except a +1 agent pixel and a few -1 neighbour pixels, 3 semantic planes of 0/1 blobs; resident in HBM
        raster = torch.zeros(B, 34, 224, 224, device=dev)
        px = torch.randint(8, 216, (B, 31, 7, 2), device=dev, generator=g)
        bi = torch.arange(B, device=dev)[:, None, None].expand(B, 31, 7)
        pi = torch.arange(31, device=dev)[None, :, None].expand(B, 31, 7)
        val = torch.full((B, 31, 7), -1.0, device=dev)
        val[:, :, 0] = 1.0
        raster[bi, pi, px[..., 1], px[..., 0]] = val
        sem = (torch.rand(B, 3, 14, 14, device=dev, generator=g) > 0.5).float()
        raster[:, 31:] = sem.repeat_interleave(16, dim=2).repeat_interleave(16, dim=3)
        del px, bi, pi, val, sem

    def one_step():
        if args.closed_loop:      # rollout loop of env_utils.py:255-304 kept on the device (policy.closed_loop_rollout)
            world, c = world0, cs
            for _ in range(args.closed_loop):
                cnd = eng.context_encode(raster, c) if use_ctx else cond      # obs -> cond_feat (context_utils.py:40-61)
                x0, _, _ = eng.sample(x_T, cnd, noise=noise, non_cond=non_cond, guidance_w=args.cfg_w,
                                      want_x1=False, want_logp=False, guidance=None if guidance is None else dict(guidance, curr_states=c))
                traj = eng.decode(x0, cnd, c, descaled_output=True)
                if has_encoder:   # "VAE encode" stage of configs[4]: the plan re-encoded to its latent posterior (context_utils.py:64-70
                    sa = eng.state_to_state_and_action(traj[..., :2].contiguous(), traj[..., 3:4].contiguous(), c[:, 2].contiguous(),
                                                       scaled_output=True)                      # -> lstm_vae.py:87-99); result unused
                    eng.traj2z(sa, cnd, noise=None)
                if distributed:
                    gather_trajectories(traj, gathered)
                world, c = eng.world_step(traj, world[:, :2].contiguous(), world[:, 2].contiguous(), 4)
            return traj
        x0, x1, logp = eng.sample(x_T, cond, noise=noise, non_cond=non_cond, guidance_w=args.cfg_w, guidance=guidance)
        traj = eng.decode(x0, cond, cs, descaled_output=True)
        if distributed:
            gather_trajectories(traj, gathered)
        return traj

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    fence()
    if not args.no_profile:
        eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        traj = one_step()
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(traj).all()), "non-finite trajectories"

    roof = None
    if not args.no_profile:
        ms, launches, flop = eng.profile_read()
        eng.profile_enable(False)
        if launches > 0 and ms > 0:
            ach = flop / (ms * 1e-3) / 1e12
            split = args.precision == "f16x2"      # 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi) per algorithmic product
            peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": pmc_traffic(B) if not split else None,
                    "peak_basis": ("v_mfma_f32_16x16x32_f16 dense peak / 3 (three MFMAs per algorithmic product)" if split
                                   else "v_mfma_f32_16x16x4_f32 dense peak"),
                    "kernel": ("conv_block_kernel<13,13,1,5,32,%s,1,32,1,0,0,0,0> (Conv1d 256 -> 256 ch, k5 + GroupNorm + Mish at L=13; "
                               "8 launches per U-Net evaluation -- 7 with 256 input channels, 1 with 128 -- every 10th evaluation timed; "
                               "tiling picked by batch size)"
                               % ("4,1" if (B + 15) // 16 * 4 * 4 >= 2048 else ("4,2" if B >= 1024 else "2,2"))),
                    "launches": int(launches), "avg_us": round(ms * 1e3 / launches, 2),
                    "flop_per_launch": flop / launches}

    total_units = world * B * n * args.steps * max(1, args.closed_loop)
    value = total_units / dt
    out = {
        "metric": "denoising-step·agent/s", "value": round(value, 1), "unit": "step·agent/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "f16x2 (fp16 hi+lo operand split, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]" if (args.scenes, args.agents, args.cfg_w, args.closed_loop, args.guide) == (32, 32, 0.0, 0, False) else "custom")
                               + (f" closed loop, {args.closed_loop} sim steps per bench step, each" if args.closed_loop else "")
                               + f": {args.scenes} scenes x {args.agents} agents per GPU, {n} denoising steps "
                               "(DDPM ancestral loop of the reference), latent seq 52 x 4, cond 256, "
                               + (f"CFG w={args.cfg_w} (2 U-Net passes per step)" if args.cfg_w else "CFG off")
                               + ("; target-speed guidance gradient every step" if args.guide else "") + "; "
                               "+ LSTM decode + unicycle roll-out" + ("; + VAE encode of the plan, kinematic world update" if args.closed_loop else "")
                               + ("; + ContextEncoder per sim step" if (args.closed_loop and use_ctx) else "")
                               + ("; RCCL all-gather of trajectories" if distributed else ""),
                   "scenes_per_gpu": args.scenes, "agents_per_scene": args.agents, "agents_per_gpu": B,
                   "denoise_steps": n, "cfg_guidance_w": args.cfg_w, "unet_passes_per_step": 2 if args.cfg_w else 1,
                   "weights": "random init (synth seed 0)"},
        "scenes_per_s": round(world * args.scenes * args.steps / dt, 2),
        "unet_tflops_effective": round(value * FLOP_PER_STEP_AGENT * (2 if args.cfg_w else 1) / 1e12, 2),
        "roofline_whole_path_frac": round(value * FLOP_PER_STEP_AGENT * (2 if args.cfg_w else 1) / 1e12
                                          / ((PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS / 3.0) * world), 4),
    }
    if roof:
        out["roofline"] = roof

    if use_ctx and not args.closed_loop:
        # ContextEncoder measured on its own (it runs once per planning call, not per denoising step): B agents, rasters
        # resident in HBM.  Twice: on the structured raster (the stem skips all-zero strips of the near-empty history planes,
        # so fewer FLOPs are executed than the dense count) and on a dense U(-1,1) raster (every MFMA issued: the honest
        # fraction of the fp32-MFMA peak).
        def time_ctx():
            eng.context_encode(raster, cs)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.context_encode(raster, cs)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / 3
        t_struct = time_ctx()
        raster.uniform_(-1.0, 1.0, generator=g)
        t_dense = time_ctx()
        tf_dense = B * FLOP_PER_CONTEXT_AGENT / t_dense / 1e12
        out["context_encoder"] = {
            "agents": B, "agents_per_s": round(B / t_struct, 1), "ms": round(t_struct * 1e3, 3),
            "raster_read_GBps": round(B * 34 * 224 * 224 * 4 / t_struct / 1e9, 1),
            "dense_raster": {"agents_per_s": round(B / t_dense, 1), "ms": round(t_dense * 1e3, 3), "tflops": round(tf_dense, 2),
                             "frac_of_f32_mfma_peak": round(tf_dense / PEAK_F32_MFMA_TFLOPS, 4)},
            "flop_per_agent_dense": FLOP_PER_CONTEXT_AGENT,
            "note": "resnet18 [34,224,224] -> 256 + state / combine MLPs (models/context_utils.py:8-61), exact fp32 MFMA; "
                    "headline = structured synthetic raster (31 near-empty history planes + 3 semantic planes, "
                    "trajdata_utils.py:409-420); bound: MFMA (the 6.8 MB raster per agent is read once, in place)"}
    if world == 1 and args.precision == "f32" and not args.no_alt_precision and not args.closed_loop:
        # secondary, clearly labelled: the SAME workload in the optional split-precision mode (include/cld.h CLD_PRECISION_F16X2:
        # fp16 hi+lo operand planes, 3 fp16 MFMAs per product, fp32 accumulate; same parity bars).  Never the headline value.
        eng2 = Engine(n_timesteps=n, device=dev, precision="f16x2")
        eng2.load_state_dict(synth.make_unet_weights(0)); eng2.load_state_dict(synth.make_decoder_weights(0)); eng2.finalize()

        def alt_step():
            x0, _, _ = eng2.sample(x_T, cond, noise=noise, non_cond=non_cond, guidance_w=args.cfg_w, guidance=guidance)
            return eng2.decode(x0, cond, cs, descaled_output=True)
        alt_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            alt_step()
        torch.cuda.synchronize(dev)
        adt = time.perf_counter() - t0
        out["f16x2_mode"] = {"value": round(B * n * args.steps / adt, 1), "unit": "step·agent/s", "ms_per_step": round(adt / args.steps * 1e3, 3),
                             "note": "optional split-precision mode (--precision f16x2), same workload and parity bars; not the headline"}
        del eng2
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B)
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
    gfx950 correction + WRITE_SIZE, profiles/run_r01.sh); PMC cannot be collected inside this process, so
    the number is the committed one for the same kernel and batch size, else null."""
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
            with open(f) as fh:
                t = json.load(fh)
            if int(t.get("batch_agents", -1)) == B:
                return int(t["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box
    hands each 1-GPU job a share of its cores; oversubscribing the share makes torch crawl)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("CLD_CPU_THREADS", "16"))))


def cpu_baseline(B):
    """The oracle (CPU restatement, validated against the reference) on the host cores: a bounded
    sample of the same workload -- the first CPU_STEPS denoising steps of the B-agent batch."""
    import torch
    from cld_amd import synth
    from oracle import cld_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    w = O.to_torch(synth.make_unet_weights(0))
    s = O.schedule(100)
    cond = torch.randn(B, 256)
    x = torch.randn(B, 52, 4)
    z = torch.randn(B, 52, 4)
    with torch.no_grad():
        O.ddpm_step(w, s, x[:64], cond[:64], 99, z[:64])     # warm-up
        t0 = time.perf_counter()
        O.ddpm_step(w, s, x, cond, 99, z)                    # probe: sizes the bounded sample (~15 s of CPU work)
        probe = time.perf_counter() - t0
        CPU_STEPS = max(1, min(99, int(15.0 / max(probe, 1e-3))))
        t0 = time.perf_counter()
        for k in range(CPU_STEPS):
            x, _, _ = O.ddpm_step(w, s, x, cond, 98 - k, z)
        dt = time.perf_counter() - t0
    return {"value": round(B * CPU_STEPS / dt, 1), "unit": "step·agent/s", "cores": cores, "kind": "port",
            "sample": f"{CPU_STEPS} denoising steps (t=98..{99 - CPU_STEPS}) of the same {B}-agent batch, torch "
                      f"{torch.__version__} CPU fp32, {cores} threads, {dt:.1f} s"}


if __name__ == "__main__":
    main()
