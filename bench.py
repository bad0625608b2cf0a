#!/usr/bin/env python3
"""bench.py -- throughput of the CLD latent-diffusion sampling path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload configs1|configs2|configs3|configs4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic agents: the full ancestral sampling loop
(U-Net evaluations + DDPM updates, reference models/dm/dm_model.py:103-142; with classifier-free guidance and the
sampling-time guidance gradient where the workload has them, upstream src/tbsim/models/diffuser.py:766-789,844-929)
followed by LSTM decode + unicycle roll-out (guide_dm_trainer.py:97-98) and, for N > 1, the RCCL all-gather of the
decoded trajectories at the rollout-step boundary.

Workloads (BASELINE.json `configs`):
  configs2 (default at N = 1; the configuration BASELINE.json's metric is quoted on): 32 scenes x 64 agents per GPU,
            100 denoising steps, CFG w = 2.0 (second U-Net pass on non_cond_feat) + guidance gradient on every step t > 0
  configs3 (default at N > 1): the FIXED job of 1,024 scenes x 64 agents, sharded by scene over the ranks
            (parallel.shard_scenes) -- strong scaling; one all-gather of [B_local,52,6] per bench step
  configs4: closed loop, the fixed job of 512 scenes x 64 agents sharded by scene; per bench step 20 sim steps x
            (ContextEncoder -> 50 denoising steps -> decode -> VAE encode of the plan -> PPO reward -> all-gather -> world update)
  configs1: 32 scenes x 32 agents per GPU, 100 steps, CFG off (round 1's headline; reported as a secondary object by default)

Prints ONE JSON line (rank 0): metric = denoising-step.agent/s over the whole job, plus `roofline` (dominant kernel vs.
the fp32-MFMA peak, timed with HIP events inside the library on the launch stream) and `cpu_baseline` (the oracle on the
host cores, bounded sample of the same workload; N = 1 only).
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

# scenes: per GPU for the per-GPU workloads, the whole job for the sharded ones
WORKLOADS = {
    "configs1": dict(idx=1, scenes=32, agents=32, denoise=100, cfg_w=0.0, guide=False, closed=0, sharded=False),
    "configs2": dict(idx=2, scenes=32, agents=64, denoise=100, cfg_w=2.0, guide=True, closed=0, sharded=False),
    "configs3": dict(idx=3, scenes=1024, agents=64, denoise=100, cfg_w=0.0, guide=False, closed=0, sharded=True),
    "configs4": dict(idx=4, scenes=512, agents=64, denoise=50, cfg_w=0.0, guide=False, closed=20, sharded=True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="BASELINE.json configs[i]; default configs2 at --gpus 1, configs3 (strong scaling) at --gpus N > 1")
    ap.add_argument("--scenes", type=int, default=None, help="override: scenes (per GPU; the whole job for configs3 / configs4)")
    ap.add_argument("--agents", type=int, default=None, help="override: agents per scene")
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--cfg-w", type=float, default=None, help="override: classifier-free guidance weight; 0 = off")
    ap.add_argument("--guide", dest="guide", action="store_true", default=None,
                    help="override: sampling-time guidance on every step t > 0 (target-speed loss through decoder + roll-out, Adam lr 0.3)")
    ap.add_argument("--no-guide", dest="guide", action="store_false")
    ap.add_argument("--closed-loop", type=int, default=None, metavar="SIM_STEPS", help="override: sim steps per bench step")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default="f32",
                    help="conv arithmetic: exact fp32 MFMA, or fp16 hi/lo split operands with fp32 accumulation (include/cld.h)")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the ContextEncoder (producer of cond_feat, SURVEY 8(f-1)): its measurement / its closed-loop stage")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline only: skip the secondary objects (configs1, f16x2 mode, ContextEncoder, configs3 / configs4 single-GPU anchors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event timing of the dominant kernel")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (backend nccl = RCCL) even at world size 1, so the collective path runs on a one-GPU box")
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
    distributed = world > 1 or args.force_dist
    backend = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
        backend = dist.get_backend()
    else:
        local_dev = local_rank % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)

    from cld_amd import synth
    from cld_amd.engine import Engine
    from cld_amd.parallel import gather_ragged, gather_trajectories, shard_scenes

    # ---- workload ---------------------------------------------------------------------------------------------------
    name = args.workload or ("configs3" if args.gpus > 1 else "configs2")
    wl = dict(WORKLOADS[name])
    custom = []
    for key, val in (("scenes", args.scenes), ("agents", args.agents), ("denoise", args.denoise_steps), ("cfg_w", args.cfg_w),
                     ("guide", args.guide), ("closed", args.closed_loop)):
        if val is not None and val != wl[key]:
            wl[key] = val
            custom.append(key)

    def make_engine(n, precision, context, encoder):
        e = Engine(n_timesteps=n, device=dev, precision=precision)
        e.load_state_dict(synth.make_unet_weights(0))            # PyTorch-default-like random init (no checkpoint ships)
        e.load_state_dict(synth.make_decoder_weights(0))
        if encoder:
            e.load_state_dict(synth.make_encoder_weights(0))
        if context:
            e.load_state_dict(synth.make_context_weights(0))
        return e.finalize()

    def structured_raster(B, g):
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
        return raster

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(wl, eng, steps, warmup, profile=False, use_ctx=True, seed=1234):
        """Time `steps` bench steps of workload `wl` on this rank's shard; -> dict(dt, B_local, B_total, units, roof)."""
        n = wl["denoise"]
        if wl["sharded"]:
            lo, hi = shard_scenes(wl["scenes"], world, rank)
            sizes = [(shard_scenes(wl["scenes"], world, r)[1] - shard_scenes(wl["scenes"], world, r)[0]) * wl["agents"] for r in range(world)]
            B, B_total = (hi - lo) * wl["agents"], wl["scenes"] * wl["agents"]
        else:
            B, B_total = wl["scenes"] * wl["agents"], wl["scenes"] * wl["agents"] * world
            sizes = [B] * world
        even = len(set(sizes)) == 1
        # synthetic inputs, resident in HBM before the timed region starts
        g = torch.Generator(device=dev)
        g.manual_seed(seed + rank)
        cond = torch.randn(B, 256, device=dev, generator=g)
        x_T = torch.randn(B, 52, 4, device=dev, generator=g)
        noise = torch.randn(n, B, 52, 4, device=dev, generator=g)
        non_cond = torch.randn(B, 256, device=dev, generator=g) if wl["cfg_w"] != 0.0 else None
        cs = torch.zeros(B, 4, device=dev)
        cs[:, 2] = torch.rand(B, device=dev, generator=g) * 15.0
        gathered = torch.empty(world * B, 52, 6, device=dev) if (distributed and even) else None
        gathered_cl = torch.empty(world * B, 52 * 6 + 3, device=dev) if (distributed and even and wl["closed"]) else None
        guidance = None
        if wl["guide"]:   # upstream defaults: adam, lr 0.3, one gradient step per denoising step (scene_edit_config.py:74-90)
            guidance = {"curr_states": cs, "target_speed": torch.rand(B, 52, device=dev, generator=g) * 12.0,
                        "loss_scale": torch.full((B,), 1.0 / (wl["agents"] * 52), device=dev), "lr": 0.3, "optimizer": "adam"}
        closed = wl["closed"]
        ctx = bool(closed) and use_ctx
        if closed:
            world0 = torch.zeros(B, 3, device=dev)
            raster = structured_raster(B, g) if ctx else None
            # PPO reward inputs (models/rl/criticmodel.py:7-64): drivable-area map, raster transform, the other agents of the scene
            dmap = (torch.rand(B, 28, 28, device=dev, generator=g) > 0.2).repeat_interleave(8, dim=1).repeat_interleave(8, dim=2).to(torch.uint8)
            rfa = torch.tensor([[2.0, 0.0, 56.0], [0.0, 2.0, 112.0], [0.0, 0.0, 1.0]], device=dev).expand(B, 3, 3).contiguous()
            S = 8
            opos0 = torch.randn(B, S, 52, 2, device=dev, generator=g) * 20.0
            oav = torch.ones(B, S, 52, dtype=torch.uint8, device=dev)
            # observation stand-in (the env is out of scope: env_trajdata.py:314-369 builds the neighbour tensors from ALL agents'
            # poses): agent b's S neighbours are the next S agents of its scene, looked up in the GATHERED plans of the previous
            # sim step at this rank's row offset -- so sim step s + 1 cannot start before the all-gather of step s has landed
            row0 = sum(sizes[:rank])
            ag = wl["agents"]
            li = torch.arange(B, device=dev)
            nb_idx = row0 + (li // ag)[:, None] * ag + (li % ag)[:, None].add(torch.arange(1, S + 1, device=dev)[None, :]).remainder(ag)   # [B,S]

            def neighbours(plans_all, pose_all, pose_own):
                """plans_all [B_all,52,6] agent-frame plans, pose_all [B_all,3] world poses they start from, pose_own [B,3] ->
                the neighbours' planned positions in each agent's own frame [B,S,52,2]."""
                pn, wn = plans_all[nb_idx][..., :2], pose_all[nb_idx]                    # [B,S,52,2], [B,S,3]
                cn, sn = torch.cos(wn[..., 2])[..., None], torch.sin(wn[..., 2])[..., None]
                wx = wn[..., 0:1] + cn * pn[..., 0] - sn * pn[..., 1]
                wy = wn[..., 1:2] + sn * pn[..., 0] + cn * pn[..., 1]
                dx, dy = wx - pose_own[:, None, 0:1], wy - pose_own[:, None, 1:2]
                co, so = torch.cos(pose_own[:, 2])[:, None, None], torch.sin(pose_own[:, 2])[:, None, None]
                return torch.stack([co * dx + so * dy, -so * dx + co * dy], dim=-1).contiguous()

        def gather(traj, out=None):
            if not distributed:
                return traj
            return gather_trajectories(traj, gathered if out is None else out) if even else gather_ragged(traj, sizes)

        def one_step():
            if closed:      # rollout loop of env_utils.py:255-304 kept on the device (policy.closed_loop_rollout)
                wpose, c, opos = world0, cs, opos0
                for _ in range(closed):
                    cnd = eng.context_encode(raster, c) if ctx else cond      # obs -> cond_feat (context_utils.py:40-61)
                    x0, _, _ = eng.sample(x_T, cnd, noise=noise, non_cond=non_cond, guidance_w=wl["cfg_w"], want_x1=False, want_logp=False,
                                          guidance=None if guidance is None else dict(guidance, curr_states=c))
                    traj = eng.decode(x0, cnd, c, descaled_output=True)
                    # "VAE encode" stage of configs[4]: the plan re-encoded to its latent posterior (context_utils.py:64-70 -> lstm_vae.py:87-99)
                    sa = eng.state_to_state_and_action(traj[..., :2].contiguous(), traj[..., 3:4].contiguous(), c[:, 2].contiguous(), scaled_output=True)
                    eng.traj2z(sa, cnd, noise=None)
                    eng.compute_reward(traj, sa, rfa, dmap, opos, oav)          # PPO reward of the plan (guide_dm_trainer.py:104)
                    # ONE all-gather per rollout step: plans and the poses they start from travel as one [B_local, 52*6 + 3] block
                    both = gather(torch.cat([traj.reshape(B, -1), wpose], dim=1), gathered_cl)
                    wpose_new, c = eng.world_step(traj, wpose[:, :2].contiguous(), wpose[:, 2].contiguous(), 4)
                    opos = neighbours(both[:, :312].reshape(-1, 52, 6), both[:, 312:], wpose_new)     # next step's reward / observation input
                    wpose = wpose_new
                return traj
            x0, x1, logp = eng.sample(x_T, cond, noise=noise, non_cond=non_cond, guidance_w=wl["cfg_w"], guidance=guidance)
            traj = eng.decode(x0, cond, cs, descaled_output=True)
            gather(traj)
            return traj

        for _ in range(warmup):
            if closed and wl.get("warm_closed"):      # a warm-up of `warm_closed` sim steps instead of a whole closed-loop pass
                full, closed = closed, wl["warm_closed"]
                one_step()
                closed = full
            else:
                one_step()
        fence()
        if profile:
            eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            traj = one_step()
        fence()
        dt = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert bool(torch.isfinite(traj).all()), "non-finite trajectories"
        roof = None
        if profile:
            ms, launches, flop = eng.profile_read()
            ex_flop = eng.profile_read_executed()[0]      # (before profile_enable resets the counters)
            eng.profile_enable(False)
            rows = (B + 15) // 16 * 16 * (2 if wl["cfg_w"] else 1)      # agents per conv launch (CFG batches both passes)
            if launches > 0 and ms > 0:
                ach = flop / (ms * 1e-3) / 1e12
                split = eng.precision == "f16x2"      # 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi) per algorithmic product
                peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
                traffic = pmc_traffic(rows) if not split else (None, None)
                wino = not split and rows >= 384          # csrc/cld_api.hip kWino1dMinRows: the Winograd F(4, 5) form of these launches

                def wino_item_form(r):                    # csrc/wino1d_kernels.hip item_form_of: 1 whole items, 2 whole items of eight waves, 0 half items
                    nfull = (r + 15) // 16 * 4
                    if 192 <= nfull <= 256:
                        return 2
                    return 0 if (nfull < 512 or 0 < nfull % 512 <= 256) else 1
                what = ("Conv1d 256 -> 256 ch, k5 + GroupNorm + Mish at L=13; 7 launches per U-Net evaluation, all with 256 input channels "
                        "-- every 10th evaluation timed")
                # `achieved` / `frac`: FLOP the MFMA pipe EXECUTED (counted by the library for the form each timed launch took:
                # cld_profile_read_executed) / time / peak -- a utilisation, <= 1 by construction.  The Winograd F(4, 5) form computes the
                # same sums with 8 x 4 MFMA k-steps per agent and channel pair instead of 5 x 13, so the rate counted on the direct
                # form's multiplies (SURVEY 8d's unit) can pass the pipe's peak: it is reported under its own name, never as `frac`.
                ex_ach = ex_flop / (ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "achieved": round(ex_ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(ex_ach / peak, 4), "traffic": traffic[0], "traffic_source": traffic[1],
                        "peak_basis": ("v_mfma_f32_16x16x32_f16 dense peak / 3 (three MFMAs per algorithmic product)" if split
                                       else "v_mfma_f32_16x16x4_f32 dense peak"),
                        "kernel": (("wino1d_edge_kernel<13,256,256,256,%d> (%%s; rows of agents, output 12 of every row in the direct form)" % wino_item_form(rows)
                                    if wino_item_form(rows) else "wino1d_conv_kernel<13,256,256,256,1> (%s; half items)") % what if wino else
                                   "conv_block_kernel<13,13,1,5,32,%s,1,32,1,0,0,0,0> (%s; tiling picked by the rows per launch)"
                                   % ("4,1" if rows >= 2048 else ("4,2" if rows >= 1024 else "2,2"), what)),
                        "agents_per_launch": rows, "launches": int(launches), "avg_us": round(ms * 1e3 / launches, 2),
                        "executed_flop_per_launch": ex_flop / launches,
                        "achieved_basis": "MFMA FLOP issued per launch as reported by the library for the form it chose "
                                          "(cld_profile_read_executed), / HIP-event time of the same launches",
                        "direct_form_flop_per_launch": flop / launches,
                        "direct_form_equivalent_tflops": round(ach, 2),
                        "multiply_reduction": round(flop / ex_flop, 4)}
        ex_t, ev_alg, ev_ex, ev_n = eng.profile_read_executed()
        ev_rows = (B + 15) // 16 * 16          # agents of this rank; a CFG evaluation carries both passes of each
        evalinfo = {"alg_flop_per_agent": ev_alg / ev_rows, "exec_flop_per_agent": ev_ex / ev_rows, "launches": ev_n}
        units = B_total * n * steps * max(1, closed)
        return {"dt": dt, "B": B, "B_total": B_total, "units": units, "roof": roof, "even": even, "eval": evalinfo}

    def describe(wl, name, custom):
        c = "BASELINE configs[%d]" % wl["idx"] + ((" with overrides (%s)" % ", ".join(custom)) if custom else "")
        c += f" closed loop, {wl['closed']} sim steps per bench step, each" if wl["closed"] else ""
        c += (f": the fixed job of {wl['scenes']} scenes x {wl['agents']} agents sharded by scene over {world} GPU(s)" if wl["sharded"]
              else f": {wl['scenes']} scenes x {wl['agents']} agents per GPU")
        c += f", {wl['denoise']} denoising steps (DDPM ancestral loop of the reference), latent seq 52 x 4, cond 256, "
        c += (f"CFG w={wl['cfg_w']} (2 U-Net passes per step, non_cond_feat)" if wl["cfg_w"] else "CFG off")
        c += "; target-speed guidance gradient through decoder + roll-out on every step t > 0" if wl["guide"] else ""
        c += "; + LSTM decode + unicycle roll-out"
        c += "; + ContextEncoder, VAE encode of the plan, PPO reward, kinematic world update per sim step" if wl["closed"] else ""
        if distributed:
            c += f"; {'RCCL' if backend == 'nccl' else backend} all-gather of the decoded trajectories per rollout step"
        return c

    # ---- headline ---------------------------------------------------------------------------------------------------
    n = wl["denoise"]
    use_ctx = not args.no_context
    eng = make_engine(n, args.precision, context=use_ctx and bool(wl["closed"]), encoder=bool(wl["closed"]))
    r = run(wl, eng, args.steps, args.warmup, profile=not args.no_profile, use_ctx=use_ctx)
    value = r["units"] / r["dt"]
    passes = 2 if wl["cfg_w"] else 1
    split = eng.precision == "f16x2"
    peak_tf = (PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS) * world
    out = {
        "metric": "denoising-step·agent/s", "value": round(value, 1), "unit": "step·agent/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(r["dt"] / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if wl["sharded"] else "weak",
        "vs_baseline": None,
        "dtype": "f32" if not split else "f16x2 (fp16 hi+lo operand split, fp32 accumulate)",      # read back from the library (cld_get_precision)
        "data": "synthetic",
        "config": {"workload": describe(wl, name, custom),
                   "scenes_total": wl["scenes"] if wl["sharded"] else wl["scenes"] * world, "agents_per_scene": wl["agents"],
                   "agents_this_rank": r["B"], "agents_total": r["B_total"], "denoise_steps": n, "cfg_guidance_w": wl["cfg_w"],
                   "unet_passes_per_step": passes, "guidance_gradient": bool(wl["guide"]), "sim_steps_per_bench_step": wl["closed"],
                   "weights": "random init (synth seed 0)"},
        "scenes_per_s": round((wl["scenes"] if wl["sharded"] else wl["scenes"] * world) * args.steps / r["dt"], 2),
        # whole path: FLOP the MFMAs of ALL launches of a denoising step's U-Net evaluation execute (per agent, both CFG passes; counted
        # by the library per launch and form) x units/s / peak: the share of the fp32-MFMA pipe the whole job keeps busy (guidance,
        # decode and the host loop are in the time, not in the FLOP).  <= 1 by construction.
        "roofline_whole_path_frac": round(value * r["eval"]["exec_flop_per_agent"] / 1e12 / peak_tf, 4),
        "unet_executed_tflops": round(value * r["eval"]["exec_flop_per_agent"] / 1e12, 2),
        "unet_launches_per_evaluation": r["eval"]["launches"],
        # the same rate counted on the direct form's multiplies (119,232,512 FLOP per U-Net pass and agent); may exceed the
        # pipe's peak where Winograd forms run -- a speed-up figure, not a utilisation
        "direct_form_equivalent_tflops": round(value * FLOP_PER_STEP_AGENT * passes / 1e12, 2),
    }
    if distributed:
        anchor = None
        try:        # the N = 1 point of the same fixed job, as last committed by a one-GPU run (context for the curve; not measured here)
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_n1.json")), reverse=True):
                with open(f) as fh:
                    o = json.load(fh).get(name + "_one_gpu")
                if o and "value" in o and not custom:
                    anchor = {"value": o["value"], "unit": o["unit"], "source": os.path.relpath(f, ROOT) + " -> " + name + "_one_gpu"}
                    break
        except Exception:
            pass
        out["distributed"] = {"backend": backend, "world_size_seen": dist.get_world_size(), "one_gpu_anchor_committed": anchor,
                              "collective": ("all_gather_into_tensor" if r["even"] else "padded all_gather_into_tensor (uneven scene split)")
                                            + " of [B_local,52,6] once per rollout step; none inside the denoising loop"}
    if r["roof"]:
        out["roofline"] = r["roof"]
    del eng
    torch.cuda.empty_cache()

    # ---- secondary objects (N = 1 only; never the headline) -----------------------------------------------------------
    extras = world == 1 and not args.no_extras and not custom and args.precision == "f32"

    def secondary(key, fn):
        try:
            out[key] = fn()
        except Exception as e:          # a failing extra must not take the headline line down
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    def sec_workload(wname, steps, warmup, precision="f32", use_ctx=True, note=None, **over):
        w2 = dict(WORKLOADS[wname], **over)
        e2 = make_engine(w2["denoise"], precision, context=bool(w2["closed"]) and use_ctx, encoder=bool(w2["closed"]))
        r2 = run(w2, e2, steps, warmup, use_ctx=use_ctx)
        v2 = r2["units"] / r2["dt"]
        p2 = 2 if w2["cfg_w"] else 1
        pk = PEAK_F16_MFMA_TFLOPS / 3.0 if precision == "f16x2" else PEAK_F32_MFMA_TFLOPS
        d = {"workload": describe(w2, wname, sorted(over)), "value": round(v2, 1), "unit": "step·agent/s",
             "ms_per_step": round(r2["dt"] / steps * 1e3, 3), "steps": steps, "agents": r2["B_total"],
             "roofline_whole_path_frac": round(v2 * r2["eval"]["exec_flop_per_agent"] / 1e12 / pk, 4),
             "direct_form_equivalent_tflops": round(v2 * FLOP_PER_STEP_AGENT * p2 / 1e12, 2)}
        if note:
            d["note"] = note
        return d

    if extras and name == "configs2":
        secondary("configs1", lambda: sec_workload("configs1", args.steps, 1))
        secondary("f16x2_mode", lambda: sec_workload(
            "configs2", args.steps, 1, precision="f16x2",
            note="optional split-precision mode (--precision f16x2), same workload and parity bars; not the headline"))
        secondary("one_scene", lambda: sec_workload(
            "configs1", 5, 2, scenes=1, agents=64,
            note="ONE 64-agent scene, 100 denoising steps, CFG off: the latency regime (a sample call is a chain of dependent launches; "
                 "ms_per_step is the latency of a sample + decode)"))
        secondary("configs3_one_gpu", lambda: sec_workload(
            "configs3", 1, 1, note="the N = 1 point of the strong-scaling job that `bench.py --gpus N` runs for N > 1 (65,536 agents on one GPU)"))
        secondary("configs4_one_gpu_shard", lambda: sec_workload(
            "configs4", 1, 1, use_ctx=use_ctx, scenes=64, closed=5, warm_closed=1,
            note="one GPU's shard of configs[4] (64 of 512 scenes = 4,096 agents), 5 of the 20 sim steps timed after one warm-up sim step"))
    if extras and use_ctx and name in ("configs1", "configs2"):
        def ctx_extra():
            # ContextEncoder measured on its own (it runs once per planning call, not per denoising step): 1,024 agents, rasters
            # resident in HBM.  Twice: on the structured raster (the stem skips all-zero strips of the near-empty history planes,
            # so fewer FLOPs are executed than the dense count) and on a dense U(-1,1) raster (every MFMA issued: the honest
            # fraction of the fp32-MFMA peak).
            Bc = 1024
            e3 = make_engine(10, "f32", context=True, encoder=False)
            g = torch.Generator(device=dev)
            g.manual_seed(99)
            raster = structured_raster(Bc, g)
            cs = torch.zeros(Bc, 4, device=dev)

            def time_ctx():
                e3.context_encode(raster, cs)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(3):
                    e3.context_encode(raster, cs)
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / 3
            t_struct = time_ctx()
            raster.uniform_(-1.0, 1.0, generator=g)
            t_dense = time_ctx()
            tf_dense = Bc * FLOP_PER_CONTEXT_AGENT / t_dense / 1e12
            return {"agents": Bc, "agents_per_s": round(Bc / t_struct, 1), "ms": round(t_struct * 1e3, 3),
                    "raster_read_GBps": round(Bc * 34 * 224 * 224 * 4 / t_struct / 1e9, 1),
                    "dense_raster": {"agents_per_s": round(Bc / t_dense, 1), "ms": round(t_dense * 1e3, 3), "tflops": round(tf_dense, 2),
                                     "frac_of_f32_mfma_peak": round(tf_dense / PEAK_F32_MFMA_TFLOPS, 4)},
                    "flop_per_agent_dense": FLOP_PER_CONTEXT_AGENT,
                    "note": "resnet18 [34,224,224] -> 256 + state / combine MLPs (models/context_utils.py:8-61), exact fp32 MFMA; "
                            "headline = structured synthetic raster (31 near-empty history planes + 3 semantic planes, "
                            "trajdata_utils.py:409-420); the 6.8 MB raster per agent is read once, in place"}
        secondary("context_encoder", ctx_extra)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(r["B"], wl)
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(rows):
    """-> (HBM bytes per launch of the dominant kernel, the committed file they come from): rocprofv3 PMC passes (FETCH_SIZE x2
    per the gfx950 correction + WRITE_SIZE, profiles/run_r*.sh).  PMC cannot be collected inside this process, so the number is
    the committed one for the same kernel and the same rows per launch -- NOT measured in this run (`traffic_source` says which
    file) -- else (None, None)."""
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic*.json")), reverse=True):
            with open(f) as fh:
                t = json.load(fh)
            if int(t.get("batch_agents", -1)) == rows:
                return int(t["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT) + " (committed rocprofv3 PMC pass; not collected in this run)"
    except Exception:
        pass
    return None, None


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


def cpu_baseline(B, wl):
    """The oracle (CPU restatement, validated against the reference) on the host cores: a bounded sample of the same
    workload -- the first CPU_STEPS denoising steps of the B-agent batch, with the workload's CFG second pass and guidance
    gradient (autograd through the oracle's decoder + roll-out) when it has them."""
    import torch
    from cld_amd import synth
    from oracle import cld_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    n = wl["denoise"]
    w = O.to_torch(synth.make_unet_weights(0))
    wd = O.to_torch(synth.make_decoder_weights(0))
    s = O.schedule(n)
    cond, non_cond = torch.randn(B, 256), torch.randn(B, 256)
    x = torch.randn(B, 52, 4)
    z = torch.randn(B, 52, 4)
    cs = torch.zeros(B, 4)
    cs[:, 2] = torch.rand(B) * 15.0
    tgt = torch.rand(B, 52) * 12.0
    ls = torch.full((B,), 1.0 / (wl["agents"] * 52))
    cfg_w = float(wl["cfg_w"])

    def step(x, i, nb=None):
        xs, cn, nc = (x, cond, non_cond) if nb is None else (x[:nb], cond[:nb], non_cond[:nb])
        t = torch.full((xs.shape[0],), i, dtype=torch.long)
        with torch.no_grad():
            eps = O.unet_forward(w, xs, cn, t)
            if cfg_w:
                eps = (1 + cfg_w) * eps - cfg_w * O.unet_forward(w, xs, nc, t)
            mean = s["x_t_cof"][i] * xs - s["noise_cof"][i] * eps
            sigma = float((0.5 * s["posterior_log_variance_clipped"][i]).exp())
        if wl["guide"] and i > 0:
            k = xs.shape[0]
            mean, _ = O.guidance_step(wd, mean, cn, cs[:k], tgt[:k], ls[:k], 0.3, None, "adam")
        return mean + sigma * z[:xs.shape[0]]

    step(x, n - 1, nb=64)                                    # warm-up (kernels, thread pool)
    step(x, n - 1, nb=min(B, 512))                           # ... and the allocator at a realistic size
    t0 = time.perf_counter()
    x = step(x, n - 1)                                       # probe: sizes the bounded sample (~20 s of CPU work)
    probe = time.perf_counter() - t0
    CPU_STEPS = max(1, min(n - 1, int(20.0 / max(probe, 1e-3))))
    t0 = time.perf_counter()
    for k in range(CPU_STEPS):
        x = step(x, n - 2 - k)
    dt = time.perf_counter() - t0
    return {"value": round(B * CPU_STEPS / dt, 1), "unit": "step·agent/s", "cores": cores, "kind": "port",
            "sample": f"{CPU_STEPS} denoising steps (t={n - 2}..{n - 1 - CPU_STEPS}) of the same {B}-agent batch"
                      + (f", CFG w={cfg_w} (2 U-Net passes)" if cfg_w else "") + (", guidance gradient by autograd" if wl["guide"] else "")
                      + f"; torch {torch.__version__} CPU fp32, {cores} threads, {dt:.1f} s"}


if __name__ == "__main__":
    main()
