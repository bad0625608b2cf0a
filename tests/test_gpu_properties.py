"""GPU tests at and beyond the oracle's reach: edge-case batch sizes against the oracle, and -- at
BASELINE.json's full sizes, where the CPU oracle would take minutes -- size-independent properties:
agents are independent (a row's result does not depend on the batch around it or on the tile it
lands in), both kernel tilings agree, the call is deterministic, the on-device RNG is sane, and the
C-ABI reports errors instead of crashing."""
import ctypes as C

import numpy as np
import pytest
import torch

from cld_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(precision):
    from cld_amd.engine import Engine
    e = Engine(n_timesteps=100, device="cuda:0", precision=precision)
    e.load_state_dict(synth.make_unet_weights(0, affine_jitter=True))
    e.load_state_dict(synth.make_decoder_weights(0))
    return e.finalize()


@pytest.fixture(scope="module")
def oracle_w():
    from oracle import cld_oracle as O
    return O, O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))


@pytest.mark.parametrize("B", [1, 15, 16, 17, 100])
def test_edge_batches_vs_oracle(eng, oracle_w, B):
    """ragged / minimal batches: one U-Net evaluation, one DDPM step and the decode chain."""
    O, w, wd = oracle_w
    x = torch.from_numpy(synth.normal(11, f"x{B}", (B, 52, 4))) * 1.5
    inp = synth.make_inputs(B, 11)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    z = torch.from_numpy(synth.normal(12, f"z{B}", (B, 52, 4)))
    t = 37
    ref = O.unet_forward(w, x, cond, torch.full((B,), t, dtype=torch.long)).numpy()
    assert np.abs(eng.unet_forward(x, cond, t).cpu().numpy() - ref).max() <= 2e-5
    xr, mr, _ = O.ddpm_step(w, O.schedule(100), x, cond, t, z)
    xn, mean, _ = eng.ddpm_step(x, cond, t, z)
    assert np.abs(xn.cpu().numpy() - xr.numpy()).max() <= 1e-4 and np.abs(mean.cpu().numpy() - mr.numpy()).max() <= 1e-4
    tr = O.decode(wd, x, cond, cs, descaled_output=True).numpy()
    assert np.abs(eng.decode(x, cond, cs, descaled_output=True).cpu().numpy() - tr).max() <= 1e-4


@pytest.mark.parametrize("B", [256, 520, 1030, 1536, 2100])
def test_every_tiling_regime_vs_oracle(eng, oracle_w, B):
    """One U-Net evaluation and one DDPM step at batch sizes that walk the launcher's tiling choices (quarter / half / full
    height 32-column tiles, the 8-wave 64-column K-split tiling around 1,024 agents, the 64-column tiling from 2,048; ragged
    last tile each time) against the oracle."""
    O, w, _ = oracle_w
    x = torch.from_numpy(synth.normal(21, f"x{B}", (B, 52, 4))) * 1.5
    cond = torch.from_numpy(synth.make_inputs(B, 21)["cond_feat"])
    z = torch.from_numpy(synth.normal(22, f"z{B}", (B, 52, 4)))
    t = 58
    with torch.no_grad():
        ref = O.unet_forward(w, x, cond, torch.full((B,), t, dtype=torch.long)).numpy()
        xr, mr, _ = O.ddpm_step(w, O.schedule(100), x, cond, t, z)
    assert np.abs(eng.unet_forward(x, cond, t).cpu().numpy() - ref).max() <= 2e-5
    xn, mean, _ = eng.ddpm_step(x, cond, t, z)
    assert np.abs(xn.cpu().numpy() - xr.numpy()).max() <= 1e-4 and np.abs(mean.cpu().numpy() - mr.numpy()).max() <= 1e-4


def test_agents_are_independent_and_position_invariant(eng):
    """BASELINE configs[2] size (32 x 64 = 2,048 agents): rows of a big batch equal, bit for bit, the same
    rows evaluated in a small batch at other tile positions (no cross-agent term anywhere on the path)."""
    B = 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 52, 4, device="cuda", generator=g) * 2
    cond = torch.randn(B, 256, device="cuda", generator=g)
    big = eng.unet_forward(x, cond, 63)
    idx = torch.tensor([0, 1, 17, 255, 1023, 1024, 2047, 5, 640, 1999], device="cuda")   # 10 rows -> one padded tile
    small = eng.unet_forward(x[idx], cond[idx], 63)
    # the big batch runs the 64-column tiling, the small one the 32-column/2-way-K-split tiling: same math, other
    # fp32 summation order inside the K split
    assert float((big[idx] - small).abs().max()) <= 2e-5
    again = eng.unet_forward(x, cond, 63)
    assert torch.equal(big, again)                     # deterministic: no atomics, fixed reduction order
    perm = torch.randperm(B, device="cuda", generator=g)
    shuffled = eng.unet_forward(x[perm], cond[perm], 63)
    assert torch.equal(shuffled, big[perm])            # same tiling, other tile positions: bit-identical


def test_max_size_batch(eng):
    """BASELINE configs[3]'s WHOLE job on one GPU (1,024 scenes x 64 = 65,536 agents: one activation is 3.5 GB, past 32-bit
    byte offsets): rows of the big batch equal the same rows evaluated as a 4,096-agent batch -- bit for bit through the
    U-Net and the DDPM update (same tiling), to an ulp of the O(100 m) positions through decode + roll-out."""
    B = 65536
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.randn(B, 52, 4, device="cuda", generator=g)
    c = torch.randn(B, 256, device="cuda", generator=g)
    z = torch.randn(B, 52, 4, device="cuda", generator=g)
    cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = 5.0
    eps = eng.unet_forward(x, c, 40)
    xn, _, _ = eng.ddpm_step(x, c, 40, z)
    traj = eng.decode(x, c, cs, descaled_output=True)
    idx = torch.cat([torch.tensor([0, 1, B // 2 - 1, B // 2, B - 4097, B - 2, B - 1], device="cuda"),
                     torch.randint(0, B, (4089,), device="cuda", generator=g)])
    xs, cc, zs, css = x[idx].contiguous(), c[idx].contiguous(), z[idx].contiguous(), cs[idx].contiguous()
    assert torch.isfinite(eps).all() and torch.isfinite(traj).all()
    assert torch.equal(eps[idx], eng.unet_forward(xs, cc, 40))
    assert torch.equal(xn[idx], eng.ddpm_step(xs, cc, 40, zs)[0])
    assert float((traj[idx] - eng.decode(xs, cc, css, descaled_output=True)).abs().max()) <= 5e-5
    del x, z, eps, xn, traj
    # the full 100-step chain with explicit noise: a 5.4 GB [100, B, 52, 4] tensor walked with 64-bit offsets
    xT = torch.randn(B, 52, 4, device="cuda", generator=g)
    nz = torch.randn(100, B, 52, 4, device="cuda", generator=g)
    x0, x1, lp = eng.sample(xT, c, noise=nz)
    x0s, x1s, lps = eng.sample(xT[idx].contiguous(), cc, noise=nz[:, idx].contiguous())
    assert torch.equal(x0[idx], x0s) and torch.equal(x1[idx], x1s) and torch.equal(lp[idx], lps)
    del xT, nz, x0, x1, lp, c
    torch.cuda.empty_cache()


def test_full_size_chain_properties(eng):
    """configs[1] size (1,024 agents, 100 steps): finite, deterministic, log_prob_final is the closed form,
    and a 24-agent prefix run alone gives the same trajectories to chain-amplified rounding."""
    B = 1024
    g = torch.Generator(device="cuda").manual_seed(9)
    xT = torch.randn(B, 52, 4, device="cuda", generator=g)
    cond = torch.randn(B, 256, device="cuda", generator=g)
    noise = torch.randn(100, B, 52, 4, device="cuda", generator=g)
    x0, x1, logp = eng.sample(xT, cond, noise=noise)
    assert bool(torch.isfinite(x0).all()) and bool(torch.isfinite(x1).all())
    assert float((logp - 22.106914).abs().max()) <= 1e-4          # -log(1e-10) - 0.5 log(2 pi), SURVEY section 7
    x0b, _, _ = eng.sample(xT, cond, noise=noise)
    assert torch.equal(x0, x0b)
    x0s, _, _ = eng.sample(xT[:24], cond[:24], noise=noise[:, :24].contiguous())
    scale = float(x0[:24].abs().max())
    assert float((x0[:24] - x0s).abs().max()) <= 1e-3 * scale


def test_on_device_rng_path(eng):
    B = 64
    g = torch.Generator(device="cuda").manual_seed(3)
    xT = torch.randn(B, 52, 4, device="cuda", generator=g)
    cond = torch.randn(B, 256, device="cuda", generator=g)
    a, _, _ = eng.sample(xT, cond, noise=None, seed=7)
    b, _, _ = eng.sample(xT, cond, noise=None, seed=7)
    c, _, _ = eng.sample(xT, cond, noise=None, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())


def test_abi_error_paths(eng):
    from cld_amd import _lib
    lib = eng.lib
    B = 4
    x = torch.zeros(B, 52, 4, device="cuda"); cond = torch.zeros(B, 256, device="cuda"); out = torch.empty_like(x)
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    # workspace too small
    assert lib.cld_unet_forward(eng._h, p(x), p(cond), 5, p(out), B, p(ws), ws.numel(), None) == -3
    assert b"workspace" in lib.cld_last_error(eng._h)
    big = torch.empty(int(lib.cld_workspace_bytes(eng._h, B)), dtype=torch.uint8, device="cuda")
    # timestep out of range, null pointer, wrong step count
    assert lib.cld_unet_forward(eng._h, p(x), p(cond), 100, p(out), B, p(big), big.numel(), None) == -1
    assert lib.cld_unet_forward(eng._h, None, p(cond), 5, p(out), B, p(big), big.numel(), None) == -1
    assert lib.cld_sample(eng._h, p(x), None, p(cond), 50, p(out), None, None, B, 0, p(big), big.numel(), None) == -1
    # weights after finalize / unknown key / wrong size on a fresh handle
    w = np.zeros(4, np.float32)
    assert lib.cld_load_weight(eng._h, b"model.final_conv.1.bias", w.ctypes.data, 4) == -2
    cfg = _lib.CldConfig(); lib.cld_default_config(C.byref(cfg))
    h = C.c_void_p(); assert lib.cld_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.cld_load_weight(h, b"model.no_such.weight", w.ctypes.data, 4) == -1
    assert lib.cld_load_weight(h, b"model.final_conv.1.bias", w.ctypes.data, 3) == -1
    assert lib.cld_load_weight(h, b"dm.model.final_conv.1.bias", w.ctypes.data, 4) == 0       # Lightning prefix
    assert lib.cld_load_weight(h, b"betas", w.ctypes.data, 4) == 0                            # schedule keys ignored
    assert lib.cld_finalize(h, None) == -2 and b"missing weight" in lib.cld_last_error(h)
    assert lib.cld_unet_forward(h, p(x), p(cond), 5, p(out), B, p(big), big.numel(), None) == -2
    assert lib.cld_destroy(h) == 0
    cfg.horizon = 40
    assert lib.cld_create(C.byref(cfg), C.byref(h)) == -1                                      # unsupported architecture


def test_reference_call_surface(eng):
    """DmModel / VaeModel mirrors: dict keys, shapes, num_samp repeat, 4-D inputs (dm_model.py:98-142,
    temporal.py:127-135, vae_model.py:108-111)."""
    from cld_amd.dm_model import DmModel
    from cld_amd.vae_model import VaeModel
    dm = DmModel({"horizon": 52}, None, n_timesteps=100, engine=eng)
    vae = VaeModel(engine=eng)
    B, N = 3, 2
    cond = torch.randn(B, 256, device="cuda")
    cs = torch.zeros(B, 4, device="cuda")
    out = dm({"history_positions": torch.zeros(B, 31, 2)}, {"cond_feat": cond, "curr_states": cs}, {"num_samp": N})
    assert set(out) == {"pred_traj", "x1", "log_prob_final", "aux_info"}
    assert out["pred_traj"].shape == (B * N, 52, 4) and out["log_prob_final"].shape == (B * N,)
    assert out["aux_info"]["cond_feat"].shape == (B * N, 256)
    assert torch.equal(out["aux_info"]["cond_feat"][0], out["aux_info"]["cond_feat"][1])      # repeat-interleave
    act = vae.lstmvae.lstm_dec(out["pred_traj"], out["aux_info"]["cond_feat"])
    assert act.shape == (B * N, 52, 2)
    traj = vae.convert_action_to_state_and_action(act.reshape(B, N, 52, 2), out["aux_info"]["curr_states"])
    assert traj.shape == (B, N, 52, 6)
    x4 = torch.randn(B, N, 52, 4, device="cuda")
    eps = dm.model(x4, {"cond_feat": cond[:, None].expand(B, N, 256)}, torch.full((B,), 7))
    assert eps.shape == (B, N, 52, 4)
    t_mixed = torch.tensor([3, 50, 3])
    e2 = dm.model(x4[:, 0], {"cond_feat": cond}, t_mixed)                                      # per-row timesteps
    assert torch.allclose(e2[1], dm.model(x4[1:2, 0], {"cond_feat": cond[1:2]}, torch.tensor([50]))[0], atol=2e-5)
    xn, mean, sigma = dm.x_Tminus1(x4[:, 0], torch.full((B,), 9), {"cond_feat": cond})
    assert xn.shape == (B, 52, 4) and sigma.shape == (B, 1, 1)
    lp = dm.log_prob(x4[:, 0], xn, {"cond_feat": cond}, torch.full((B,), 9))
    assert lp.shape == (B,) and bool(torch.isfinite(lp).all())


def test_get_action_and_closed_loop(eng, oracle_w):
    """policy surface (algos.py:2024-2099 contract) + world update (env_trajdata.py:452-468)."""
    from cld_amd.dm_model import DmModel
    from cld_amd.policy import Action, CldPolicy, closed_loop_rollout
    from cld_amd.vae_model import VaeModel
    O, w, wd = oracle_w
    dm, vae = DmModel(None, None, 100, engine=eng), VaeModel(engine=eng)
    pol = CldPolicy(dm, vae, disable_control_on_stationary=True)
    B = 6
    inp = synth.make_inputs(B, 21)
    inp["curr_states"][0, 2] = 0.1                                    # one stationary agent
    cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
    nz = synth.make_noise(B, 100, 77)
    noise = {"x_T": torch.from_numpy(nz["x_T"]), "noise": torch.from_numpy(nz["noise"])}
    act, info = pol.get_action({"cond_feat": cond, "curr_states": cs}, step_index=0, noise=noise)
    assert isinstance(act, Action) and act.positions.shape == (B, 52, 2) and act.yaws.shape == (B, 52, 1)
    assert info["action_samples"]["positions"].shape == (B, 1, 52, 2)
    assert float(act.positions[0].abs().max()) == 0.0 and float(act.yaws[0].abs().max()) == 0.0     # stationary -> zeroed
    # the composition equals sample -> decode of the engine (bit for bit) and the oracle's chain to chain tolerance
    x0, _, _ = eng.sample(noise["x_T"], cond, noise=noise["noise"])
    tr = eng.decode(x0, cond, cs, descaled_output=True)
    assert torch.equal(act.positions[1:], tr[1:, :, :2]) and torch.equal(act.yaws[1:], tr[1:, :, 3:4])
    # world update vs a NumPy restatement of env_trajdata.py:452-468
    centroid = torch.from_numpy(synth.normal(3, "ctr", (B, 2)) * 50).cuda()
    yaw = torch.from_numpy(synth.uniform(3, "yaw", (B,), -3.1, 3.1)).cuda()
    k = 4
    world, ncs = eng.world_step(tr, centroid, yaw, k)
    trn, cn, yn = tr.cpu().numpy().astype(np.float64), centroid.cpu().numpy().astype(np.float64), yaw.cpu().numpy().astype(np.float64)
    for b in range(B):
        wfa = np.array([[np.cos(yn[b]), np.sin(yn[b])], [-np.sin(yn[b]), np.cos(yn[b])]])
        exp_xy = trn[b, k, :2] @ wfa + cn[b]
        assert np.abs(world[b, :2].cpu().numpy() - exp_xy).max() <= 1e-3 * max(1.0, np.abs(exp_xy).max())
        assert abs(float(world[b, 2]) - (yn[b] + trn[b, k, 3])) <= 1e-5
    assert torch.equal(ncs[:, 2], tr[:, k, 2]) and float(ncs[:, [0, 1, 3]].abs().max()) == 0.0
    # 3 closed-loop sim steps stay finite and move the non-stationary agents
    from cld_amd.timer import Timers
    timers = Timers(device="cuda:0")
    poses = closed_loop_rollout(CldPolicy(dm, vae), lambda s, wld, c: cond, centroid, yaw, cs, n_sim_steps=3, timers=timers)
    assert poses.shape == (3, B, 3) and bool(torch.isfinite(poses).all())
    # per-phase timers under the reference's keys (env_utils.py:268-298); the network phase dominates on the device
    assert all(k in timers._timers for k in ("obs", "network", "env_step", "step"))
    assert timers.gpu_ms("network") > 10 * timers.gpu_ms("env_step") > 0.0 and timers.gpu_ms("step") >= timers.gpu_ms("network")


def test_sampling_call_is_graph_capturable(eng):
    """include/cld.h promises that the compute calls neither allocate nor synchronise, so a whole 100-step sampling call
    (~3,000 launches) can be captured into a HIP graph and replayed: the replay must reproduce the eager result bit for bit,
    also after the inputs are overwritten in place."""
    B = 48
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, 52, 4, device="cuda", generator=g)
    c = torch.randn(B, 256, device="cuda", generator=g)
    z = torch.randn(100, B, 52, 4, device="cuda", generator=g)
    eager = eng.sample(x, c, noise=z)[0].clone()
    torch.cuda.synchronize()                           # the engine's workspace is shared: no overlap with the capture stream
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        eng.sample(x, c, noise=z)                      # warm-up on the capture stream (workspace, kernel attributes)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            out = eng.sample(x, c, noise=z)
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], eager)
    x2 = torch.randn(B, 52, 4, device="cuda", generator=g)
    eager2 = eng.sample(x2, c, noise=z)[0].clone()
    x.copy_(x2)                                        # same buffers, new contents
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], eager2)


def test_abi_error_paths_next_rows(eng):
    """Error reporting of the context / guidance / reward entry points: status codes, never a crash."""
    from cld_amd import _lib
    lib = eng.lib
    p = lambda t: C.c_void_p(t.data_ptr())
    B = 2
    img = torch.zeros(B, 34, 224, 224, device="cuda"); cs = torch.zeros(B, 4, device="cuda"); out = torch.empty(B, 256, device="cuda")
    ws = torch.empty(int(lib.cld_context_workspace_bytes(eng._h, B)), dtype=torch.uint8, device="cuda")
    # this engine has no context_encoder weights
    assert lib.cld_context_encode(eng._h, p(img), p(cs), p(out), None, B, p(ws), ws.numel(), None) == -2
    assert b"context_encoder" in lib.cld_last_error(eng._h)
    mean = torch.zeros(B, 52, 4, device="cuda"); cond = torch.zeros(B, 256, device="cuda"); tgt = torch.zeros(B, 52, device="cuda")
    big = torch.empty(int(lib.cld_workspace_bytes(eng._h, B)), dtype=torch.uint8, device="cuda")
    g = _lib.CldGuidance(cs.data_ptr(), tgt.data_ptr(), None, 0.3, -1.0, 7)                    # unknown optimizer
    assert lib.cld_guidance_step(eng._h, p(mean), p(cond), C.byref(g), 0.5, None, p(mean), None, None, B, p(big), big.numel(), None) == -1
    g = _lib.CldGuidance(None, tgt.data_ptr(), None, 0.3, -1.0, 0)                             # missing curr_states
    assert lib.cld_guidance_step(eng._h, p(mean), p(cond), C.byref(g), 0.5, None, p(mean), None, None, B, p(big), big.numel(), None) == -1
    assert lib.cld_sample_guided(eng._h, p(mean), None, p(cond), None, 0.0, None, 100, p(mean), None, None, B, 0, p(big), big.numel(), None) == -1
    R = torch.eye(3, device="cuda").repeat(B, 1, 1); dm = torch.ones(B, 8, 8, dtype=torch.uint8, device="cuda")
    traj = torch.zeros(B, 52, 6, device="cuda")
    assert lib.cld_compute_reward(eng._h, p(traj), None, p(R), p(dm), 8, 8, None, None, 3, 52, 0.8, p(out), None, None, B, None) == -1   # S > 0 without arrays
    assert lib.cld_compute_reward(eng._h, p(traj), None, p(R), p(dm), 8, 8, None, None, 0, 0, 0.8, None, None, None, B, None) == -1        # no output
    r = torch.empty(B, device="cuda")
    assert lib.cld_compute_reward(eng._h, p(traj), None, p(R), p(dm), 8, 8, None, None, 0, 0, 0.8, p(r), None, None, B, None) == 0
    torch.cuda.synchronize()
    assert float(r.abs().max()) == 0.0                       # on the map, drivable everywhere, no neighbours, no jerk term


def test_full_size_chain_vs_oracle(eng, oracle_w):
    """BASELINE configs[1] at full size -- 32 x 32 = 1,024 agents, 100 steps, same noise -- directly against the oracle
    (the GPU box's host cores finish it in ~10-20 s).  Bar: 1e-3 relative to max|x0| (the random-init chain
    amplifies to |x0| ~ 1e4, where the reference disagrees with itself by 5e-7 relative across thread counts);
    measured 1.8e-6.  Decoded trajectories: 1e-3 abs on O(100 m) positions; log_prob_final exact to 1e-4."""
    O, w, wd = oracle_w
    B, n = 1024, 100
    inp, nz = synth.make_inputs(B, 1), synth.make_noise(B, n, 123)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    xT, z = torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"])
    x0, x1, lp = eng.sample(xT, cond, noise=z)
    traj = eng.decode(x0, cond, cs, descaled_output=True)
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(16, __import__("os").cpu_count() or 1))
    try:
        with torch.no_grad():
            ref = O.sample(w, O.schedule(n), xT, z, cond)
            reft = O.decode(wd, ref["pred_traj"], cond, cs)
    finally:
        torch.set_num_threads(nthr)
    scale = float(ref["pred_traj"].abs().max())
    assert float((x0.cpu() - ref["pred_traj"]).abs().max()) <= 1e-3 * scale
    assert float((x1.cpu() - ref["x1"]).abs().max()) <= 1e-3 * scale
    assert float((traj.cpu() - reft).abs().max()) <= 1e-3
    assert float((lp.cpu() - ref["log_prob_final"]).abs().max()) <= 1e-4


def _configs2(precision, n, B=2048):
    """BASELINE configs[2] at its launch sizes: 32 x 64 = 2,048 agents, CFG w = 2 (both U-Net passes one 4,096-row launch set),
    guidance on every step t > 0 (2,048 agents: the 8-agents-per-workgroup kernel on all 256 CUs, the form the bench runs)."""
    from oracle import cld_oracle as O
    from cld_amd.engine import Engine
    e = Engine(n, "cuda:0", precision=precision)
    e.load_state_dict(synth.make_unet_weights(0, affine_jitter=True)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
    w, wd = O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))
    inp, nz = synth.make_inputs(B, 7), synth.make_noise(B, n, 9)
    d = {"cond": torch.from_numpy(inp["cond_feat"]).cuda(), "cs": torch.from_numpy(inp["curr_states"]).cuda(),
         "tgt": torch.from_numpy(synth.uniform(7, "tgt", (B, 52), 0.0, 12.0)).cuda(),
         "non_cond": torch.from_numpy(synth.normal(7, "non_cond_feat", (B, 256))).cuda(),
         "x_T": torch.from_numpy(nz["x_T"]).cuda(), "z": torch.from_numpy(nz["noise"]).cuda()}
    return O, e, w, wd, d


class _threads:
    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(min(16, __import__("os").cpu_count() or 1))
    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def test_configs2_guided_cfg_chain_every_step_vs_oracle(precision):
    """configs[2] at full launch size on a 6-step schedule, Adam (upstream's default optimiser, lr 0.3): the GPU chain is
    driven step by step (cld_sample_step) and EVERY step is checked against the oracle on the chain's own x_t
    (tests/guided_checks.py: posterior mean 1e-4, gradient 2e-5 relative, guided mean within rounding + what that gradient
    tolerance lets Adam's sign-like step do at that element -- every element bounded, none exempt).  Then the one-call chain
    (cld_sample_guided) must equal the stepwise one bit for bit, and zero loss weight the unguided CFG chain."""
    from guided_checks import check_guided_step
    n = 6
    O, e, w, wd, d = _configs2(precision, n)
    gd = {"curr_states": d["cs"], "target_speed": d["tgt"], "lr": 0.3, "optimizer": "adam"}
    gd_ref = {"curr_states": d["cs"].cpu(), "target_speed": d["tgt"].cpu(), "lr": 0.3, "optimizer": "adam"}
    x, shares = d["x_T"], []
    with _threads():
        for it in range(n):
            i = n - 1 - it
            got, share = check_guided_step(e, O, w, wd, x, d["cond"], d["non_cond"], 2.0, gd, gd_ref, i, d["z"][it], tag="configs2 n=6")
            shares.append(share)
            x = got["x_next"]
    print("share of elements whose Adam budget exceeds 1e-3 per step:", [f"{s_:.4f}" for s_ in shares])
    assert max(shares) <= 0.05                        # the budget is an exception for kink-adjacent elements, not the rule
    x0, _, _ = e.sample(d["x_T"], d["cond"], noise=d["z"], non_cond=d["non_cond"], guidance_w=2.0, guidance=gd)
    assert torch.equal(x0, x)
    plain, _, _ = e.sample(d["x_T"], d["cond"], noise=d["z"], non_cond=d["non_cond"], guidance_w=2.0)
    g0, _, _ = e.sample(d["x_T"], d["cond"], noise=d["z"], non_cond=d["non_cond"], guidance_w=2.0, guidance=dict(gd, loss_scale=torch.zeros(2048)))
    assert torch.equal(plain, g0)


def test_configs2_guided_cfg_chain_sgd_all_elements(precision):
    """The same chain with SGD (delta = -lr g: no sign function, nothing to flip): end to end against the oracle's autograd
    restatement with the strict bar on ALL 425,984 latent elements."""
    n, lr = 6, 2000.0          # |g| <~ 4e-3: steps of up to ~8 per element, far above the bar the result is held to
    O, e, w, wd, d = _configs2(precision, n)
    gd = {"curr_states": d["cs"], "target_speed": d["tgt"], "lr": lr, "optimizer": "sgd"}
    x0, x1, _ = e.sample(d["x_T"], d["cond"], noise=d["z"], non_cond=d["non_cond"], guidance_w=2.0, guidance=gd)
    with _threads():
        ref = O.sample_guided(w, wd, O.schedule(n), d["x_T"].cpu(), d["z"].cpu(), d["cond"].cpu(), d["cs"].cpu(), d["tgt"].cpu(), None, lr, "sgd",
                              d["non_cond"].cpu(), 2.0)
        unguided = O.sample_cfg(w, O.schedule(n), d["x_T"].cpu(), d["z"].cpu(), d["cond"].cpu(), d["non_cond"].cpu(), 2.0)["pred_traj"]
    scale = max(1.0, float(ref["pred_traj"].abs().max()))
    err = float((x0.cpu() - ref["pred_traj"]).abs().max())
    moved = float((ref["pred_traj"] - unguided).abs().max())
    print(f"SGD chain: max|d| {err:.3e} of max|x0| {scale:.3e}; guidance moved x0 by up to {moved:.3e}")
    assert moved > 20 * 1e-3 * scale                  # the guidance term is far above the bar it is checked to
    assert err <= 1e-3 * scale
    assert float((x1.cpu() - ref["x1"]).abs().max()) <= 1e-3 * scale


def test_configs2_teacher_forced_steps_of_the_100_step_schedule(precision):
    """configs[2] as the bench runs it -- 2,048 agents, CFG w = 2, guidance every step, the 100-step schedule: the GPU chain is
    driven step by step and at t = 99, 50 and 1 the step is checked against the oracle on the chain's own x_t (the latents
    have grown to their real magnitudes by then), all elements (tests/guided_checks.py).  The one-call chain must equal the
    stepwise one bit for bit, so the 97 steps in between run the same kernels on the same data."""
    from guided_checks import check_guided_step
    n = 100
    O, e, w, wd, d = _configs2(precision, n)
    gd = {"curr_states": d["cs"], "target_speed": d["tgt"], "lr": 0.3, "optimizer": "adam"}
    gd_ref = {"curr_states": d["cs"].cpu(), "target_speed": d["tgt"].cpu(), "lr": 0.3, "optimizer": "adam"}
    x, x1 = d["x_T"], None
    with _threads():
        for it in range(n):
            i = n - 1 - it
            if i in (99, 50, 1):
                got, share = check_guided_step(e, O, w, wd, x, d["cond"], d["non_cond"], 2.0, gd, gd_ref, i, d["z"][it], tag="configs2 n=100")
                print(f"t = {i}: max|x_t| = {float(x.abs().max()):.3e}, Adam-budget share {share:.4f}")
                assert share <= 0.05
            else:
                got = e.sample_step(x, d["cond"], i, z=d["z"][it], non_cond=d["non_cond"], guidance_w=2.0, guidance=gd)
            x = got["x_next"]
            if i == 1:
                x1 = x
    x0c, x1c, _ = e.sample(d["x_T"], d["cond"], noise=d["z"], non_cond=d["non_cond"], guidance_w=2.0, guidance=gd)
    assert torch.equal(x0c, x) and torch.equal(x1c, x1)


def test_configs4_closed_loop_at_per_gpu_size(precision):
    """BASELINE configs[4] on one GPU's shard: 64 scenes x 64 = 4,096 agents, 50 denoising steps per planning call, the loop
    ContextEncoder-free (cond_feat supplied) -> sample -> decode -> VAE encode of the plan -> world update, 2 sim steps.
    Size-independent property: agents are independent, so 40 rows picked across the batch and run as their own small batch
    through the same loop land on the same poses (to the chain's rounding through decode: the small batch takes other tilings),
    and the encoded plan's posterior matches row for row."""
    from cld_amd.dm_model import DmModel
    from cld_amd.engine import Engine
    from cld_amd.policy import CldPolicy, closed_loop_rollout
    from cld_amd.vae_model import VaeModel
    n, B, S = 50, 4096, 2
    e = Engine(n_timesteps=n, device="cuda:0", precision=precision)
    for sd in (synth.make_unet_weights(0, affine_jitter=True), synth.make_decoder_weights(0), synth.make_encoder_weights(0)):
        e.load_state_dict(sd)
    e.finalize()
    pol = CldPolicy(DmModel(None, None, n_timesteps=n, engine=e), VaeModel(engine=e))
    g = torch.Generator(device="cuda").manual_seed(44)
    conds = [torch.randn(B, 256, device="cuda", generator=g) for _ in range(S)]
    xT = torch.randn(B, 52, 4, device="cuda", generator=g)
    nz = torch.randn(n, B, 52, 4, device="cuda", generator=g)
    cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = torch.rand(B, device="cuda", generator=g) * 15.0
    ctr = torch.randn(B, 2, device="cuda", generator=g) * 100.0
    yaw = (torch.rand(B, device="cuda", generator=g) - 0.5) * 6.0
    encoded = []

    def run(idx):
        sel = (lambda t: t) if idx is None else (lambda t: t[idx].contiguous())
        plans = []

        def gather(traj):          # stands where the all-gather sits: keeps every executed plan, and re-encodes it (the "VAE encode" stage)
            plans.append(traj)
            sa = e.state_to_state_and_action(traj[..., :2].contiguous(), traj[..., 3:4].contiguous(), traj[:, 0, 2].contiguous(), scaled_output=True)
            encoded.append(e.traj2z(sa, sel(conds[len(plans) - 1]), noise=None)[1])
            return traj
        poses = closed_loop_rollout(pol, lambda s, wld, c: sel(conds[s]), sel(ctr), sel(yaw), sel(cs), n_sim_steps=S, gather=gather,
                                    noise={"x_T": sel(xT), "noise": nz if idx is None else nz[:, idx].contiguous()})
        return poses, plans
    poses, plans = run(None)
    mu_big = [m.clone() for m in encoded]
    encoded.clear()
    assert poses.shape == (S, B, 3) and bool(torch.isfinite(poses).all()) and all(bool(torch.isfinite(m).all()) for m in mu_big)
    idx = torch.cat([torch.tensor([0, 1, 63, 64, 2047, 2048, 4095], device="cuda"), torch.randint(0, B, (33,), device="cuda", generator=g)])
    poses_s, plans_s = run(idx)
    scale = max(1.0, float(plans[0][idx].abs().max()))
    for s_ in range(S):
        assert float((plans[s_][idx] - plans_s[s_]).abs().max()) <= 2e-3 * scale, s_
        assert float((mu_big[s_][idx] - encoded[s_]).abs().max()) <= 2e-3 * max(1.0, float(mu_big[s_].abs().max())), s_
    assert float((poses[:, idx] - poses_s).abs().max()) <= 2e-3 * max(1.0, float(poses.abs().max()))
    again, _ = run(None)
    assert torch.equal(again, poses)                   # deterministic


# ---------------------------------------------------------------------------------------------------------
# (e) multi-GPU: the RCCL branch, executed (world size 1 is all a one-GPU box can offer)
# ---------------------------------------------------------------------------------------------------------
def _child(args, env_extra, timeout=600):
    """Run a fresh python child (the GPU is initialised by this process already; nothing is exec'ed over it)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run([sys.executable] + args, env=env, capture_output=True, text=True, timeout=timeout,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_rccl_all_gather_on_device_tensors(precision):
    """`torch.distributed` backend "nccl" (= RCCL on ROCm) initialised with a device id, `all_gather_into_tensor` and the padded
    ragged gather of cld_amd.parallel on DEVICE tensors: executed once, at world size 1 (the driver's 8-GPU run is the first
    time more ranks exist).  The reference has no counterpart (single device, utils/trainer_utils.py:122-141)."""
    if precision != "f32":
        pytest.skip("precision-independent")
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, os.getcwd())\n"
        "from cld_amd.parallel import gather_trajectories, gather_ragged\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', device_id=torch.device('cuda', 0))\n"
        "assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "t = torch.randn(4096, 52, 6, device='cuda')\n"
        "out = torch.empty(4096, 52, 6, device='cuda')\n"
        "g = gather_trajectories(t, out)\n"
        "assert g.data_ptr() == out.data_ptr() and torch.equal(g, t)\n"
        "r = gather_ragged(t[:1000], [1000])\n"
        "assert torch.equal(r, t[:1000])\n"
        "dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group(); print('RCCL_OK')\n")
    p = _child(["-c", code], {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_bench_distributed_branch_runs_on_rccl(precision):
    """bench.py's distributed branch (init_process_group("nccl", device_id=...), the all-gather inside the timed region, the
    max-over-ranks reduction) on a real device: `--force-dist` at world size 1 must report backend nccl and land within 5 % of
    the same run without torch.distributed (configs[2], 2 timed steps; boxes repeat to ~1.5 %).  The closed loop (configs[4]
    shard, 2 sim steps) runs with the neighbours of sim step s + 1 read from the tensor the all-gather of step s filled."""
    import json
    if precision != "f32":
        pytest.skip("precision-independent")
    common = ["bench.py", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-profile"]
    plain = _child(common, {})
    assert plain.returncode == 0, plain.stderr[-2000:]
    forced = _child(common + ["--force-dist"], {"MASTER_PORT": "29542"})
    assert forced.returncode == 0, forced.stderr[-2000:]
    a, b = json.loads(plain.stdout.strip().splitlines()[-1]), json.loads(forced.stdout.strip().splitlines()[-1])
    assert "distributed" not in a
    assert b["distributed"]["backend"] == "nccl" and b["distributed"]["world_size_seen"] == 1
    print("configs2 plain", a["value"], "forced-dist", b["value"])
    assert abs(b["value"] / a["value"] - 1.0) <= 0.05
    closed = _child(["bench.py", "--workload", "configs4", "--scenes", "16", "--closed-loop", "2", "--steps", "1", "--warmup", "1", "--no-extras",
                     "--no-cpu-baseline", "--no-profile", "--force-dist"], {"MASTER_PORT": "29543"})
    assert closed.returncode == 0, closed.stderr[-2000:]
    c = json.loads(closed.stdout.strip().splitlines()[-1])
    assert c["distributed"]["backend"] == "nccl" and c["value"] > 0
