"""GPU parity: the HIP path (through the C-ABI) against the golden vectors recorded from the
reference and against the oracle on the same seeded inputs.  Bars:
  teacher-forced single U-Net / DDPM step : <= 1e-4 abs (scaled by max|value| when that exceeds 1)
  end-to-end chains                        : <= 1e-3 * max|x0|  (and <= 1e-3 abs when max|x0| <= 10)
  decode                                   : <= 1e-4 abs on [B,52,6]
"""
import numpy as np
import pytest
import torch

from cld_amd import synth

pytestmark = pytest.mark.gpu

NBUF_OFF = 3 * 208 + 1792          # floats per agent before the activation buffers (csrc/cld_api.hip carve())
ACT = 3328


PRECISION = "f32"     # set per module run by the `precision` fixture (tests/conftest.py): the whole file runs per precision mode


@pytest.fixture(scope="module", autouse=True)
def _precision_mode(precision):
    global PRECISION
    PRECISION = precision
    yield
    PRECISION = "f32"


def _engine(n=100, jitter=True, decoder=True):
    from cld_amd.engine import Engine
    e = Engine(n_timesteps=n, device="cuda:0", precision=PRECISION)
    e.load_state_dict(synth.make_unet_weights(0, affine_jitter=jitter))
    if decoder:
        e.load_state_dict(synth.make_decoder_weights(0))
    return e.finalize()


@pytest.fixture(scope="module")
def eng_jitter(_precision_mode):
    return _engine(100, True)


@pytest.fixture(scope="module")
def eng_default(_precision_mode):
    return _engine(100, False)


def _buf(e, B, idx, C, L):
    """Activation buffer `idx` of the last U-Net evaluation as [B, C, L] (reference layout)."""
    bp = (B + 15) // 16 * 16
    ws = e._ws.view(torch.float32)
    off = bp * NBUF_OFF + idx * bp * ACT
    raw = ws[off: off + bp * ACT]
    if e.precision == "f16x2" and idx != 7:      # S22 activations: per 8-channel block [8 x fp16 hi][8 x fp16 lo]
        h = raw.view(torch.float16).reshape(bp, L, C // 8, 2, 8).float()
        val = (h[:, :, :, 0] + h[:, :, :, 1]).reshape(bp, L, C)
    else:
        val = raw.reshape(bp, L, C)
    return val[:B].permute(0, 2, 1).cpu().numpy()


@pytest.mark.parametrize("tag", ["default", "jitter"])
def test_unet_forward_golden(golden, tag, eng_default, eng_jitter):
    meta, g = golden(f"unet_forward_{tag}")
    e = eng_jitter if meta["affine_jitter"] else eng_default
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "unet_x", (B, 52, 4))) * 3.0
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    for r, t in enumerate(meta["t"]):
        # one launch per layer: every layer's output exists in the workspace (the layer chains a batch of this size takes by
        # default keep the 64-channel levels in LDS and leave the noise prediction where final_conv.0's activations were)
        e.force_kernel("unet", "layers")
        eps = e.unet_forward(x, cond, t).cpu().numpy()
        # intermediate activations of row r (taps recorded with per-row timesteps)
        for name, idx, C, L in (("downs_1_1", 4, 128, 26), ("downs_2_1", 5, 256, 13), ("final_conv_0", 7, 64, 52)):
            got = _buf(e, B, idx, C, L)[r]
            assert np.abs(got - g["tap_" + name][r]).max() <= 2e-5, (name, t)
        assert np.abs(eps[r] - g["eps"][r]).max() <= 2e-5, t
        e.force_kernel("unet", "auto")
        eps = e.unet_forward(x, cond, t).cpu().numpy()
        for name, idx, C, L in (("downs_1_1", 4, 128, 26), ("downs_2_1", 5, 256, 13)):       # the taps that exist either way
            assert np.abs(_buf(e, B, idx, C, L)[r] - g["tap_" + name][r]).max() <= 2e-5, (name, t)
        assert np.abs(eps[r] - g["eps"][r]).max() <= 2e-5, t


def test_unet_forward_vs_oracle_ragged_batch(eng_jitter):
    from oracle import cld_oracle as O
    B = 37                                           # not a multiple of the 16-agent tile
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=True))
    x = torch.from_numpy(synth.normal(7, "rag_x", (B, 52, 4))) * 2.0
    cond = torch.from_numpy(synth.make_inputs(B, 7)["cond_feat"])
    for t in (73, 3):
        ref = O.unet_forward(w, x, cond, torch.full((B,), t, dtype=torch.long)).numpy()
        got = eng_jitter.unet_forward(x, cond, t).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5


def test_ddpm_step_golden(golden, eng_jitter):
    meta, g = golden("ddpm_step")
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "step_x", (B, 52, 4)))
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    z = torch.from_numpy(synth.normal(meta["noise_seed"], "step_z", (B, 52, 4)))
    for i in meta["t"]:
        xn, mean, sigma = eng_jitter.ddpm_step(x, cond, i, z)
        scale = max(1.0, float(np.abs(g[f"mean_t{i}"]).max()))
        assert np.abs(mean.cpu().numpy() - g[f"mean_t{i}"]).max() <= 1e-4 * scale
        assert np.abs(xn.cpu().numpy() - g[f"x_next_t{i}"]).max() <= 1e-4 * scale
        assert sigma == pytest.approx(float(g[f"sigma_t{i}"][0]), rel=2e-6)


@pytest.mark.parametrize("n,jitter", [(10, True), (50, True), (100, False), (100, True)])
def test_full_chain_golden(golden, n, jitter):
    meta, g = golden(f"sample_n{n}_{'jitter' if jitter else 'default'}")
    e = _engine(n, jitter, decoder=False)
    B = meta["B"]
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, logp = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]))
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"chain n={n} jitter={jitter} {k}: max|d|={err:.3e} max|ref|={scale:.3e} rel={err/scale:.2e}")
        assert err <= 1e-3 * scale
        if scale <= 10:
            assert err <= 1e-3
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


def test_cfg_chain_golden(golden):
    """classifier-free guidance (w = 2): two U-Net passes per step batched as one 2B launch set."""
    meta, g = golden("sample_cfg_n10")
    B, n = meta["B"], meta["n_timesteps"]
    e = _engine(n, True, decoder=False)
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    non_cond = torch.from_numpy(synth.normal(meta["in_seed"], "non_cond_feat", (B, 256)))
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]),
                         non_cond=non_cond, guidance_w=meta["guidance_w"])
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"cfg chain {k}: max|d|={err:.3e} max|ref|={scale:.3e}")
        assert err <= 1e-3 * scale
    # w = 0 with a non_cond given must reproduce the plain chain bit for bit (same kernels, same order)
    a, _, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]))
    b, _, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]), non_cond=non_cond, guidance_w=0.0)
    assert torch.equal(a, b)


def test_cfg_ragged_batch_vs_oracle():
    """CFG with a batch that is not a multiple of the 16-agent tile: the two halves of the 2B launch set stay
    tile-aligned (rows [0,32) cond, [32,64) uncond for B = 20)."""
    from oracle import cld_oracle as O
    B, n, wgt = 20, 10, 1.5
    e = _engine(n, True, decoder=False)
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=True))
    cond = torch.from_numpy(synth.make_inputs(B, 31)["cond_feat"])
    non_cond = torch.from_numpy(synth.normal(31, "nc", (B, 256)))
    nz = synth.make_noise(B, n, 32)
    ref = O.sample_cfg(w, O.schedule(n), torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"]), cond, non_cond, wgt)
    x0, x1, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]), non_cond=non_cond, guidance_w=wgt)
    scale = float(ref["pred_traj"].abs().max())
    assert float((x0.cpu() - ref["pred_traj"]).abs().max()) <= 1e-3 * scale
    assert float((x1.cpu() - ref["x1"]).abs().max()) <= 1e-3 * scale


def test_log_prob_golden(golden, eng_jitter):
    meta, g = golden("log_prob")
    B = meta["B"]
    x_t = torch.from_numpy(synth.normal(meta["in_seed"], "lp_xt", (B, 52, 4)))
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    lp = eng_jitter.log_prob(x_t, torch.from_numpy(g["x_tm1_t50"]), cond, 50).cpu().numpy()
    assert np.allclose(lp, g["log_prob_t50"], rtol=1e-4, atol=1e-4)
    lp0 = eng_jitter.log_prob(x_t, torch.from_numpy(g["x_tm1_t0"]), cond, 0).cpu().numpy()
    assert np.isfinite(lp0).all()        # sigma_0 = 1e-10: value is rounding-noise / 1e-20


def test_log_prob_at_t0_golden(golden, eng_jitter):
    """a-8 where the reference calls it: t == 0 (src/trainers/guide_dm_trainer.py:160-164), sigma_0 = 1e-10, so the value is
    -(x0 - mean)^2 / 2e-20 + 22.1 ~ -1e13 ... -1e15.  Golden `log_prob_t0` = the reference's own `dm.log_prob`:
      t0  : x_tm1 = mean_ref + 1e-3 z on the inputs of fixture `log_prob`;
      ppo : M = 128, (x1, x0) of a chain the reference sampled with the old weights, evaluated under perturbed weights
            (the PPO ratio's numerator as the trainer forms it).
    Bar: 1e-3 RELATIVE per agent (one ulp of an O(1) mean is ~1e-4 of a 1e-3 offset).  A flushed sigma_0 (inf / nan), a wrong
    sigma_0 (value off by its square) or a wrong reduction all fail it."""
    from cld_amd.engine import Engine
    meta, g = golden("log_prob_t0")
    assert float(np.exp(0.5 * eng_jitter.posterior_log_variance_clipped[0])) == pytest.approx(1e-10, rel=1e-5)
    B = meta["t0"]["B"]
    x_t = torch.from_numpy(synth.normal(meta["in_seed"], "lp_xt", (B, 52, 4)))
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    _, _, sigma0 = eng_jitter.ddpm_step(x_t, cond, 0, None)
    assert sigma0 == pytest.approx(float(g["t0_sigma"][0]), rel=2e-6) and sigma0 > 0.0          # 1e-10, not flushed
    lp = eng_jitter.log_prob(x_t, torch.from_numpy(g["t0_x_tm1"]), cond, 0).cpu().numpy()
    rel = np.abs(lp / g["t0_log_prob"] - 1.0).max()
    print(f"log_prob t=0: ref {g['t0_log_prob'][:2]} got {lp[:2]} max rel {rel:.2e}")
    assert np.isfinite(lp).all() and rel <= 1e-3
    # PPO-shaped
    M = meta["ppo"]["M"]
    w_old = synth.make_unet_weights(0, affine_jitter=True)
    for k in ("model.final_conv.1.weight", "model.final_conv.1.bias"):
        w_old[k] = (w_old[k] * np.float32(meta["ppo"]["final_scale"])).astype(np.float32)
    e_new = Engine(100, "cuda:0", precision=PRECISION)
    e_new.load_state_dict(synth.perturb_unet_weights(w_old, **meta["ppo"]["perturb"])); e_new.finalize()
    cond = torch.from_numpy(synth.make_inputs(M, meta["in_seed"])["cond_feat"])
    lp = e_new.log_prob(torch.from_numpy(g["ppo_x1"]), torch.from_numpy(g["ppo_x0"]), cond, 0).cpu().numpy()
    rel = np.abs(lp / g["ppo_log_prob_new"] - 1.0).max()
    print(f"log_prob PPO-shaped: ref {g['ppo_log_prob_new'][:2]} got {lp[:2]} max rel {rel:.2e}")
    assert np.isfinite(lp).all() and rel <= 1e-3
    # the trainer's first PPO iteration evaluates log_prob(x1, x0) with the SAMPLING weights and must get log_prob_final back
    # (ratio == 1): x0 is the t = 0 mean itself.  That holds only if cld_log_prob repeats cld_sample's last step bit for bit.
    e_old = Engine(100, "cuda:0", precision=PRECISION)
    e_old.load_state_dict(w_old); e_old.finalize()
    nz = synth.make_noise(M, 100, meta["noise_seed"])
    x0, x1, logp = e_old.sample(torch.from_numpy(nz["x_T"]) * meta["ppo"]["x_scale"], cond,
                                noise=torch.from_numpy(nz["noise"]) * meta["ppo"]["noise_scale"])
    again = e_old.log_prob(x1, x0, cond, 0)
    assert torch.equal(again, logp) and np.allclose(logp.cpu().numpy(), g["ppo_log_prob_old"], atol=1e-4)
    # and the chain itself against the reference's (O(1) latents: the absolute bar)
    assert float((x0.cpu() - torch.from_numpy(g["ppo_x0"])).abs().max()) <= 1e-3
    assert float((x1.cpu() - torch.from_numpy(g["ppo_x1"])).abs().max()) <= 1e-3


def test_decode_golden(golden, eng_jitter):
    meta, g = golden("decode")
    B = meta["B"]
    inp = synth.make_inputs(B, meta["in_seed"])
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    _, gs = golden("sample_n10_jitter")
    for tag, z in (("small", torch.from_numpy(synth.normal(meta["in_seed"], "dec_z", (B, 52, 4)))),
                   ("x0n10", torch.from_numpy(gs["pred_traj"]))):
        act = eng_jitter.lstm_decode(z, cond)
        assert np.abs(act.cpu().numpy() - g[f"act_{tag}"]).max() <= 5e-6
        for desc, key in ((True, "traj_descaled_"), (False, "traj_scaled_")):
            tr = eng_jitter.action_to_state(act, cs, True, desc).cpu().numpy()
            assert np.abs(tr - g[key + tag]).max() <= 1e-4
        tr2, act2 = eng_jitter.decode(z, cond, cs, descaled_output=True, want_act=True)
        assert np.abs(tr2.cpu().numpy() - g["traj_descaled_" + tag]).max() <= 1e-4
        assert np.abs(act2.cpu().numpy() - g[f"act_{tag}"]).max() <= 5e-6
    st = eng_jitter.action_to_state(torch.from_numpy(g["dyn_actions"]), cs, False, False).cpu().numpy()
    assert np.abs(st[..., :4] - g["dyn_states"]).max() <= 1e-4
    assert np.array_equal(st[..., 4:], g["dyn_actions"])


def test_encoder_row_golden(golden):
    """VAE encoder: state -> state+action, LSTM encoder, mu/logvar heads, reparametrisation."""
    from cld_amd.engine import Engine
    from cld_amd.vae_model import VaeModel
    meta, g = golden("encode")
    B = meta["B"]
    e = Engine(100, "cuda:0")
    e.load_state_dict(synth.make_unet_weights(0))
    e.load_state_dict(synth.make_encoder_weights(meta["w_seed"]))
    e.finalize()
    fut = {k: torch.from_numpy(v) for k, v in synth.make_future(B, meta["in_seed"]).items()}
    vae = VaeModel(engine=e)
    x6 = vae.get_state_and_action_from_data_batch(fut).cpu().numpy()
    assert np.abs(x6 - g["state_action"]).max() <= 2e-4
    x6s = vae.get_state_and_action_from_data_batch(fut, scaled=True)
    assert torch.allclose(x6s.cpu(), vae.scale_traj(torch.from_numpy(g["state_action"])), atol=2e-4)
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    nz = torch.from_numpy(synth.normal(meta["noise_seed"], "enc_noise", (B, 52, 4)))
    z, mu, lv = vae.lstmvae.traj2z(x6s, cond, noise=nz)
    for got, k in ((z, "z"), (mu, "mu"), (lv, "logvar")):
        assert np.abs(got.cpu().numpy() - g[k]).max() <= 2e-5, k
    # decoder absent on this handle: the call reports it instead of crashing
    with pytest.raises(Exception):
        e.lstm_decode(z, cond)


# ---------------------------------------------------------------------------------------------------------
# f-1  ContextEncoder (models/context_utils.py:8-61)
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eng_ctx(_precision_mode):
    from cld_amd.engine import Engine
    e = Engine(n_timesteps=10, device="cuda:0")
    e.load_state_dict(synth.make_unet_weights(0))
    e.load_state_dict(synth.make_context_weights(0))
    return e.finalize()


def test_context_encoder_golden(golden, eng_ctx):
    """cond_feat against the reference's ContextEncoder.forward (MLPs / concat from the reference, ResNet-18 from the
    oracle's restatement: torchvision is absent, so the map branch itself is parity-unpinned).  Bars: 1e-4 relative on
    the fc output (|.| ~ 60), 1e-4 abs on cond_feat (|.| ~ 1)."""
    from cld_amd.context_utils import ContextEncoder
    meta, g = golden("context")
    B = meta["B"]
    batch = {"history_positions": torch.from_numpy(synth.normal(meta["in_seed"], "hist_pos", (B, 31, 2))),
             "history_yaws": torch.from_numpy(synth.normal(meta["in_seed"], "hist_yaw", (B, 31, 1)) * 0.3),
             "curr_speed": torch.from_numpy(synth.uniform(meta["in_seed"], "curr_speed", (B,), 0.0, 15.0)),
             "image": torch.from_numpy(synth.make_raster(B, meta["in_seed"], dense=True)).cuda()}
    aux = ContextEncoder(eng_ctx)(batch)
    assert set(aux) == {"cond_feat", "curr_states", "image"}
    assert np.array_equal(aux["curr_states"].cpu().numpy(), g["curr_states"])
    _, mf = eng_ctx.context_encode(batch["image"], aux["curr_states"], want_map_feat=True)
    scale = float(np.abs(g["map_feat"]).max())
    assert np.abs(mf.cpu().numpy() - g["map_feat"]).max() <= 1e-4 * scale
    assert np.abs(aux["cond_feat"].cpu().numpy() - g["cond_feat"]).max() <= 1e-4


@pytest.mark.parametrize("B,dense", [(5, True), (3, False)])
def test_context_encoder_vs_oracle(eng_ctx, B, dense):
    """Ragged agent counts (tiles of 2 / 4 agents at 14x14 / 7x7) and the sparse raster (zero-strip shortcut of the stem)."""
    from oracle import cld_oracle as O
    img = torch.from_numpy(synth.make_raster(B, 7, dense=dense))
    cs = torch.from_numpy(synth.make_inputs(B, 7)["curr_states"])
    cond, mf = eng_ctx.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
    taps = {}
    ref = O.context_encode(O.to_torch(synth.make_context_weights(0)), img, cs, taps)
    scale = float(taps["map_feat"].abs().max())
    assert (mf.cpu() - taps["map_feat"]).abs().max().item() <= 1e-4 * scale
    assert (cond.cpu() - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("B", [1, 5, 67])
def test_context_winograd_and_direct_convolutions_agree(eng_ctx, B):
    """The 3x3 / stride-1 convolutions of the ResNet-18 run as Winograd F(2x2, 3x3) (wino_kernels.hip) by default and as the
    implicit GEMM of the other convolutions on request: two kernels, one function.  Ragged tile lists (B = 1: 784 / 196 / 49 / 16
    tiles of 2x2 outputs in workgroups of 32; the 7x7 map's tiles hang over its edge), against each other and against the oracle."""
    from oracle import cld_oracle as O
    img = torch.from_numpy(synth.make_raster(B, 13, dense=True))
    cs = torch.from_numpy(synth.make_inputs(B, 13)["curr_states"])
    try:
        eng_ctx.force_kernel("context", "direct")
        cd, md = eng_ctx.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
        cd, md = cd.clone(), md.clone()
        eng_ctx.force_kernel("context", "winograd")
        cw, mw = eng_ctx.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
    finally:
        eng_ctx.force_kernel("context", "auto")
    ca, ma = eng_ctx.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
    assert torch.equal(ca, cw) and torch.equal(ma, mw)            # the default IS the Winograd form
    assert not torch.equal(md, mw)                                # two different kernels ran
    scale = float(md.abs().max())
    assert float((md - mw).abs().max()) <= 2e-5 * scale
    assert float((cd - cw).abs().max()) <= 2e-5
    # round 4: the 56x56 and 28x28 layers of the Winograd form run as F(4x4, 3x3) (wino44_kernels.hip); F(2x2, 3x3) for every layer -- last
    # round's form -- on request: a third kernel set, the same function to the same bars
    try:
        eng_ctx.force_kernel("context", "winograd_f2")
        c2, m2 = eng_ctx.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
        c2, m2 = c2.clone(), m2.clone()
    finally:
        eng_ctx.force_kernel("context", "auto")
    assert not torch.equal(m2, mw)
    print(f"context B={B}: max|F(4x4) - direct| = {float((md - mw).abs().max()) / scale:.2e}, max|F(2x2) - direct| = {float((md - m2).abs().max()) / scale:.2e} of max|map_feat|")
    assert float((m2 - mw).abs().max()) <= 2e-5 * scale and float((md - m2).abs().max()) <= 2e-5 * scale
    assert float((c2 - cw).abs().max()) <= 2e-5
    if B <= 5:
        taps = {}
        ref = O.context_encode(O.to_torch(synth.make_context_weights(0)), img, cs, taps)
        assert (mw.cpu() - taps["map_feat"]).abs().max().item() <= 1e-4 * scale
        assert (cw.cpu() - ref).abs().max().item() <= 1e-4


def test_context_encoder_agents_are_independent(eng_ctx):
    """Size-independent property: every agent's cond_feat depends on its own raster / state only, so a 261-agent batch
    (two passes of <= 256 agents, ragged tiles) built by repeating 3 agents must reproduce their rows bit for bit."""
    img3 = torch.from_numpy(synth.make_raster(3, 11, dense=False)).cuda()
    cs3 = torch.from_numpy(synth.make_inputs(3, 11)["curr_states"]).cuda()
    c3 = eng_ctx.context_encode(img3, cs3)
    idx = torch.arange(261, device="cuda") % 3
    big = eng_ctx.context_encode(img3[idx], cs3[idx])
    assert torch.equal(big, c3[idx])


# ---------------------------------------------------------------------------------------------------------
# f-3  sampling-time guidance (upstream diffuser.py:844-929, guidance_loss.py:219-254,2221-2282)
# ---------------------------------------------------------------------------------------------------------
def _guidance_inputs(meta):
    B = meta["B"]
    inp = synth.make_inputs(B, meta["in_seed"])
    scale = np.concatenate([np.full(n, w / (n * 52), np.float32) for n, w in zip(meta["scenes"], meta["weights"])])
    return (torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"]),
            torch.from_numpy(synth.normal(meta["in_seed"], "guide_mean", (B, 52, 4))),
            torch.from_numpy(synth.uniform(meta["in_seed"], "guide_target_speed", (B, 52), 0.0, 12.0)), torch.from_numpy(scale))


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_guidance_step_golden(golden, eng_jitter, opt):
    """cld_guidance_step against the reference's own perturb() (golden 'guidance').  SGD compares the raw gradient path
    (step = lr * g, |g| ~ 1e-3): 1e-5 relative.  Adam normalises the step to +-lr wherever |g| >> 1e-8: 1e-3 abs."""
    from oracle import cld_oracle as O
    meta, g = golden("guidance")
    cond, cs, mean, tgt, scale = _guidance_inputs(meta)
    gd = {"curr_states": cs, "target_speed": tgt, "loss_scale": scale, "lr": meta[opt]["lr"], "perturb_th": None, "optimizer": opt}
    mg, grad = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
    _, gref = O.guidance_step(O.to_torch(synth.make_decoder_weights(0)), mean, cond, cs, tgt, scale, meta[opt]["lr"], None, opt)
    assert (grad.cpu() - gref).abs().max().item() <= 2e-5 * gref.abs().max().item()
    step = np.abs(g[f"guided_{opt}"] - mean.numpy()).max()
    assert np.abs(mg.cpu().numpy() - g[f"guided_{opt}"]).max() <= (1e-3 if opt == "adam" else max(2e-5 * step, 2.5e-7))   # 2.5e-7: one ulp of the O(1) means
    # the clip variants the reference intends but never applies: sigma_t and an explicit threshold
    for th, bound in (("sigma", 0.5), (0.1, 0.1)):
        gd2 = dict(gd, perturb_th=th)
        mg2 = eng_jitter.guidance_step(mean, cond, gd2, sigma=0.5)
        ref2, _ = O.guidance_step(O.to_torch(synth.make_decoder_weights(0)), mean, cond, cs, tgt, scale, meta[opt]["lr"], bound, opt)
        assert (mg2.cpu() - ref2).abs().max().item() <= 1e-3
        assert (mg2.cpu() - mean).abs().max().item() <= bound + 1e-6


@pytest.fixture(scope="module")
def eng10(_precision_mode):
    return _engine(10, True)


def test_guided_chain_vs_oracle(eng10):
    """10-step guided chain (Adam, lr 0.3, the upstream defaults), plain and with CFG: the GPU chain is driven step by step and
    every step is checked against the oracle's autograd restatement on the chain's own x_t, every element bounded
    (tests/guided_checks.py); the one-call chain must equal the stepwise one bit for bit.  The same chain with SGD goes end
    to end against the oracle with the strict bar on all elements; zero loss weight must reproduce the unguided chain."""
    from guided_checks import check_guided_step
    from oracle import cld_oracle as O
    B, n = 6, 10
    w, wd = O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))
    inp = synth.make_inputs(B, 3)
    nz = synth.make_noise(B, n, 5)
    cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
    tgt = torch.from_numpy(synth.uniform(3, "tgt", (B, 52), 0.0, 12.0)).cuda()
    non_cond = torch.from_numpy(synth.normal(3, "non_cond_feat", (B, 256))).cuda()
    x_T, z = torch.from_numpy(nz["x_T"]).cuda(), torch.from_numpy(nz["noise"]).cuda()
    gd = {"curr_states": cs, "target_speed": tgt, "lr": 0.3, "optimizer": "adam"}
    gd_ref = {"curr_states": cs.cpu(), "target_speed": tgt.cpu(), "lr": 0.3, "optimizer": "adam"}
    for cfg_w, nc in ((0.0, None), (2.0, non_cond)):
        x = x_T
        for it in range(n):
            got, _ = check_guided_step(eng10, O, w, wd, x, cond, nc, cfg_w, gd, gd_ref, n - 1 - it, z[it], tag=f"n=10 cfg_w={cfg_w}")
            x = got["x_next"]
        x0, _, _ = eng10.sample(x_T, cond, noise=z, non_cond=nc, guidance_w=cfg_w, guidance=gd)
        assert torch.equal(x0, x)
        sgd = dict(gd, lr=2000.0, optimizer="sgd")
        x0, _, _ = eng10.sample(x_T, cond, noise=z, non_cond=nc, guidance_w=cfg_w, guidance=sgd)
        ref = O.sample_guided(w, wd, O.schedule(n), x_T.cpu(), z.cpu(), cond.cpu(), cs.cpu(), tgt.cpu(), None, 2000.0, "sgd",
                              None if nc is None else nc.cpu(), cfg_w)
        scale = max(1.0, float(ref["pred_traj"].abs().max()))
        assert float((x0.cpu() - ref["pred_traj"]).abs().max()) <= 1e-3 * scale
    plain, _, _ = eng10.sample(x_T, cond, noise=z)
    zero = dict(gd, loss_scale=torch.zeros(B))
    g0, _, _ = eng10.sample(x_T, cond, noise=z, guidance=zero)
    assert torch.equal(plain, g0)


def test_reward_golden_and_oracle(golden, eng_jitter):
    """cld_compute_reward: collision counts and offroad flags bit-exact against the reference's helpers (golden 'reward'),
    counts / jerk term against the oracle."""
    from oracle import cld_oracle as O
    meta, g = golden("reward")
    ri = {k: torch.from_numpy(v) for k, v in synth.make_reward_inputs(meta["B"], meta["in_seed"]).items()}
    r, off, col = eng_jitter.compute_reward(ri["traj"], ri["traj_scaled"], ri["raster_from_agent"], ri["drivable_map"],
                                            ri["other_pos"], ri["other_avail"])
    assert np.array_equal(col.cpu().numpy(), g["collision_reward"])
    assert np.array_equal((off.cpu() < 0).float().numpy(), g["any_offroad"])
    rr, ro, rc = O.compute_reward(ri["traj"], ri["traj_scaled"], ri["raster_from_agent"], ri["drivable_map"], ri["other_pos"],
                                  ri["other_avail"])
    assert torch.equal(off.cpu(), ro) and torch.equal(col.cpu(), rc)
    assert (r.cpu() - rr).abs().max().item() <= 1e-5 * max(1.0, rr.abs().max().item())
    rates = eng_jitter.failure_rate_compute(ri["traj"], {"raster_from_agent": ri["raster_from_agent"], "drivable_map": ri["drivable_map"],
                                                         "all_other_agents_future_positions": ri["other_pos"],
                                                         "all_other_agents_future_availability": ri["other_avail"]})
    for k, v in meta["rates"].items():                      # the reference's failure_rate_compute on the same batch
        assert abs(rates[k] - v) <= 1e-6, k
    # no other agents: collision term vanishes
    r2, _, c2 = eng_jitter.compute_reward(ri["traj"], ri["traj_scaled"], ri["raster_from_agent"], ri["drivable_map"])
    assert float(c2.abs().max()) == 0.0


def test_get_action_from_raw_observation(eng_ctx):
    """The policy surface on the reference's observation batch (image + history): get_action must equal
    context_encode -> sample -> decode composed by hand (bit for bit: same kernels, same inputs)."""
    from cld_amd.dm_model import DmModel
    from cld_amd.policy import CldPolicy
    from cld_amd.vae_model import VaeModel
    eng = eng_ctx.__class__(n_timesteps=10, device="cuda:0")
    for sd in (synth.make_unet_weights(0), synth.make_decoder_weights(0), synth.make_context_weights(0)):
        eng.load_state_dict(sd)
    eng.finalize()
    dm, vae = DmModel(None, None, n_timesteps=10, engine=eng), VaeModel(engine=eng)
    B = 3
    obs = {"history_positions": torch.zeros(B, 31, 2), "history_yaws": torch.zeros(B, 31, 1),
           "curr_speed": torch.from_numpy(synth.uniform(2, "curr_speed", (B,), 0.0, 15.0)),
           "image": torch.from_numpy(synth.make_raster(B, 2)).cuda()}
    nz = synth.make_noise(B, 10, 9)
    noise = {"x_T": torch.from_numpy(nz["x_T"]), "noise": torch.from_numpy(nz["noise"])}
    act, info = CldPolicy(dm, vae).get_action(obs, noise=noise)
    aux = vae.context_encoder(obs)
    x0, _, _ = eng.sample(noise["x_T"], aux["cond_feat"], noise=noise["noise"])
    traj = eng.decode(x0, aux["cond_feat"], aux["curr_states"], descaled_output=True)
    assert torch.equal(act.positions, traj[..., :2]) and torch.equal(act.yaws, traj[..., 3:4])


def test_guidance_mfma_kernel_vs_oracle_and_valu(eng_jitter):
    """The MFMA formulations of the guidance kernel (16 agents per workgroup on 16x16x4 tiles; 8 agents per workgroup on the
    4x4x1 blocks): gradient against the oracle's autograd at B = 523 (ragged last tile in both), and against the 2-agent VALU
    kernel on the same inputs."""
    import os
    from oracle import cld_oracle as O
    B = 523
    inp = synth.make_inputs(B, 21)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    mean = torch.from_numpy(synth.normal(21, "guide_mean", (B, 52, 4)))
    tgt = torch.from_numpy(synth.uniform(21, "guide_target_speed", (B, 52), 0.0, 12.0))
    gd = {"curr_states": cs, "target_speed": tgt, "lr": 2.0, "perturb_th": None, "optimizer": "sgd"}
    outs = {}
    for k in ("mfma", "quad", "valu"):
        eng_jitter.force_kernel("guide", k)
        try:
            outs[k] = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
            torch.cuda.synchronize()
        finally:
            eng_jitter.force_kernel("guide", "auto")
    torch.set_num_threads(8)
    _, gref = O.guidance_step(O.to_torch(synth.make_decoder_weights(0)), mean, cond, cs, tgt, None, 2.0, None, "sgd")
    gmax = gref.abs().max().item()
    for k in outs:
        assert (outs[k][1].cpu() - gref).abs().max().item() <= 2e-5 * gmax, k
    for k in ("mfma", "quad"):
        assert (outs[k][0] - outs["valu"][0]).abs().max().item() <= 2.0 * 4e-5 * gmax + 2.5e-7, k


def test_vae_forward_reconstruction_path():
    """VaeModel.forward (vae_model.py:64-82) minus the loss: pre_vae -> lstmvae (encode, reparametrise, decode) ->
    convert_action_to_state_and_action, against the oracle's composition of the same pinned pieces."""
    from cld_amd.engine import Engine
    from cld_amd.vae_model import VaeModel
    from oracle import cld_oracle as O
    e = Engine(n_timesteps=10, device="cuda:0")
    sds = [synth.make_unet_weights(0), synth.make_decoder_weights(0), synth.make_encoder_weights(0), synth.make_context_weights(0)]
    for sd in sds:
        e.load_state_dict(sd)
    e.finalize()
    vae = VaeModel(engine=e)
    B = 3
    fut = synth.make_future(B, 4)
    img = torch.from_numpy(synth.make_raster(B, 4))
    batch = {"history_positions": torch.zeros(B, 31, 2), "history_yaws": torch.zeros(B, 31, 1),
             "curr_speed": torch.from_numpy(fut["curr_speed"]), "image": img.cuda(),
             "target_positions": torch.from_numpy(fut["target_positions"]), "target_yaws": torch.from_numpy(fut["target_yaws"])}
    nz = torch.from_numpy(synth.normal(4, "enc_noise", (B, 52, 4)))
    out = vae.forward(batch, noise=nz)
    w = {}
    for sd in sds[1:]:
        w.update(O.to_torch(sd))
    cs = torch.zeros(B, 4); cs[:, 2] = batch["curr_speed"]
    cond = O.context_encode(w, img, cs)
    x6 = O.state_to_state_and_action(batch["target_positions"], batch["target_yaws"], batch["curr_speed"], scaled=True)
    z, mu, lv = O.traj2z(w, x6, cond, nz)
    ref = O.decode(w, z, cond, cs, descaled_output=True)
    assert (out["mu"].cpu() - mu).abs().max().item() <= 5e-5 and (out["logvar"].cpu() - lv).abs().max().item() <= 5e-5
    assert (out["output"].cpu() - ref[..., :2]).abs().max().item() <= 2e-4 * max(1.0, ref[..., :2].abs().max().item())


@pytest.mark.parametrize("B", [63, 64, 2048, 2049])
def test_guidance_kernel_selection_boundaries(eng_jitter, B):
    """The batch sizes either side of the rule that picks the guidance kernel (launch_guide: 2-agent VALU below 64 agents, 8
    agents per workgroup up to 2,048, 16 agents per workgroup from 2,049): whatever the library picks must agree with the
    16-agent kernel forced on the same inputs (the three formulations differ in summation order only)."""
    inp = synth.make_inputs(B, 31)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    mean = torch.from_numpy(synth.normal(31, "guide_mean", (B, 52, 4)))
    tgt = torch.from_numpy(synth.uniform(31, "guide_target_speed", (B, 52), 0.0, 12.0))
    gd = {"curr_states": cs, "target_speed": tgt, "lr": 2.0, "perturb_th": None, "optimizer": "sgd"}
    auto = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
    eng_jitter.force_kernel("guide", "mfma")
    try:
        ref = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
        torch.cuda.synchronize()
    finally:
        eng_jitter.force_kernel("guide", "auto")
    gmax = float(ref[1].abs().max())
    assert float((auto[1] - ref[1]).abs().max()) <= 2e-5 * gmax
    assert float((auto[0] - ref[0]).abs().max()) <= 2.0 * 4e-5 * gmax + 2.5e-7


@pytest.mark.parametrize("kernel", ["valu", "mfma", "quad"])
def test_guidance_combined_losses_golden(golden, eng_jitter, kernel):
    """Target-speed + speed-limit + acceleration-limit guidance in one step against the reference's own perturb() (golden
    'guidance', combo_sgd), through every formulation of the guidance kernel."""
    import os
    meta, g = golden("guidance")
    cond, cs, mean, tgt, _ = _guidance_inputs(meta)
    c = meta["combo_sgd"]
    n0, n1 = meta["scenes"]
    ts = torch.tensor([c["scene0"]["target_speed"] / (n0 * 52)] * n0 + [0.0] * n1)
    sl = torch.tensor([c["scene0"]["speed_limit"][1] / (n0 * 52)] * n0 + [c["scene1"]["speed_limit"][1] / (n1 * 52)] * n1)
    al = torch.tensor([c["scene0"]["acc_limit"][1] / (n0 * 52)] * n0 + [0.0] * n1)
    gd = {"curr_states": cs, "target_speed": tgt, "loss_scale": ts, "speed_limit": (c["scene0"]["speed_limit"][0], sl),
          "acc_limit": (c["scene0"]["acc_limit"][0], al), "lr": c["lr"], "perturb_th": None, "optimizer": "sgd"}
    eng_jitter.force_kernel("guide", kernel)
    try:
        mg = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5)
        torch.cuda.synchronize()
    finally:
        eng_jitter.force_kernel("guide", "auto")
    step = np.abs(g["guided_combo_sgd"] - mean.numpy()).max()
    assert np.abs(mg.cpu().numpy() - g["guided_combo_sgd"]).max() <= max(3e-5 * step, 2.5e-7)
    # a speed limit alone (no target-speed term) is accepted too
    only = eng_jitter.guidance_step(mean, cond, {"curr_states": cs, "speed_limit": (6.0, 1.0), "lr": 1.0, "optimizer": "sgd"}, sigma=0.5)
    assert bool(torch.isfinite(only).all())


@pytest.mark.parametrize("kernel", ["valu", "mfma", "quad"])
def test_guidance_waypoint_golden(golden, eng_jitter, kernel):
    """Waypoint guidance (TargetPosAtTimeLoss): the gradient runs through positions, yaw and the speed-dependent yaw-rate
    bound of the unicycle roll-out, then through both decoder output channels.  Against the reference's perturb() (golden
    'guidance', waypoint_sgd) and, for the raw gradient, against the oracle's autograd."""
    import os
    from oracle import cld_oracle as O
    meta, g = golden("guidance")
    cond, cs, mean, tgt, _ = _guidance_inputs(meta)
    c = meta["waypoint_sgd"]
    n0, n1 = meta["scenes"]
    wp = torch.zeros(meta["B"], 2); wp[:n0] = torch.from_numpy(synth.uniform(meta["in_seed"], "guide_waypoint", (n0, 2), -5.0, 25.0))
    wt = torch.tensor(c["target_time"] + [0] * n1)
    tps = torch.tensor([c["weight"] / n0] * n0 + [0.0] * n1)
    ts = torch.tensor([0.0] * n0 + [c["scene1_target_speed_weight"] / (n1 * 52)] * n1)
    gd = {"curr_states": cs, "target_speed": tgt, "loss_scale": ts, "target_pos": (wp, wt, tps), "lr": c["lr"], "perturb_th": None,
          "optimizer": "sgd"}
    eng_jitter.force_kernel("guide", kernel)
    try:
        mg, grad = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
        torch.cuda.synchronize()
    finally:
        eng_jitter.force_kernel("guide", "auto")
    _, gref = O.guidance_step(O.to_torch(synth.make_decoder_weights(0)), mean, cond, cs, tgt, ts, c["lr"], None, "sgd", target_pos=(wp, wt, tps))
    assert (grad.cpu() - gref).abs().max().item() <= 5e-5 * gref.abs().max().item()
    step = np.abs(g["guided_waypoint_sgd"] - mean.numpy()).max()
    assert np.abs(mg.cpu().numpy() - g["guided_waypoint_sgd"]).max() <= max(5e-5 * step, 2.5e-7)


@pytest.mark.parametrize("kernel", ["valu", "mfma", "quad"])
def test_guidance_waypoint_yaw_bound_path(kernel):
    """The yaw-rate clip of the roll-out (bound = max(min(0.5|v|, 2pi/|v|), 0.1)) routes the waypoint gradient into the speed
    when it is active.  The reference statistics never reach it with random weights, so this case widens the yaw-rate scale
    (std[5] = 4) and starts agents slowly; gradient against the oracle's autograd with the same statistics."""
    import os
    from cld_amd.engine import Engine
    from oracle import cld_oracle as O
    mean6, std6 = list(O.NORM_MEAN), list(O.NORM_STD)
    std6[5] = 4.0
    e = Engine(n_timesteps=10, device="cuda:0", norm_info=(mean6, std6))
    e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
    B = 24
    inp = synth.make_inputs(B, 31)
    cond = torch.from_numpy(inp["cond_feat"])
    cs = torch.from_numpy(inp["curr_states"]).clone()
    cs[:, 2] = torch.from_numpy(synth.uniform(31, "slow", (B,), 0.05, 4.0))
    mean = torch.from_numpy(synth.normal(31, "guide_mean", (B, 52, 4)))
    wp = torch.from_numpy(synth.uniform(31, "wp", (B, 2), -3.0, 10.0))
    wt = torch.from_numpy(synth.uniform(31, "wt", (B,), 5.0, 51.9)).long()
    tps = torch.full((B,), 1.0 / B)
    gd = {"curr_states": cs, "target_pos": (wp, wt, tps), "lr": 1.0, "perturb_th": None, "optimizer": "sgd"}
    e.force_kernel("guide", kernel)
    old = O.NORM_STD
    try:
        _, grad = e.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
        torch.cuda.synchronize()
        O.NORM_STD = tuple(std6)
        wdec = O.to_torch(synth.make_decoder_weights(0))
        _, gref = O.guidance_step(wdec, mean, cond, cs, None, None, 1.0, None, "sgd", target_pos=(wp, wt, tps))
        with torch.no_grad():      # the case must actually clip: yaw rates beyond the bound on a fair share of steps
            traj = O.decode(wdec, mean, cond, cs, True)
            v_prev = torch.cat([cs[:, 2:3].clamp(-10, 30), traj[:, :-1, 2]], dim=1).abs()
            yb = torch.minimum(0.5 * v_prev, 2 * np.pi / v_prev.clamp(min=0.1)).clamp(min=0.1)
            clipped = (traj[..., 5].abs() > yb).float().mean().item()
    finally:
        O.NORM_STD = old
        e.force_kernel("guide", "auto")
    assert clipped > 0.05, clipped
    assert (grad.cpu() - gref).abs().max().item() <= 1e-4 * gref.abs().max().item()


@pytest.mark.parametrize("kernel", ["valu", "mfma", "quad"])
def test_guidance_targetpos_softmin_golden(golden, eng_jitter, kernel):
    """TargetPosLoss (softmin over the steps >= m, encoded as target_time = -(m + 1)) against the reference's perturb()."""
    import os
    from oracle import cld_oracle as O
    meta, g = golden("guidance")
    cond, cs, mean, _, _ = _guidance_inputs(meta)
    c = meta["targetpos_sgd"]
    n0, n1 = meta["scenes"]
    wp = torch.zeros(meta["B"], 2); wp[:n0] = torch.from_numpy(synth.uniform(meta["in_seed"], "guide_waypoint", (n0, 2), -5.0, 25.0))
    m = int(c["min_target_time"] * 52)
    wt = torch.tensor([-(m + 1)] * n0 + [0] * n1)
    tps = torch.tensor([c["weight"] / n0] * n0 + [0.0] * n1)
    gd = {"curr_states": cs, "target_pos": (wp, wt, tps), "lr": c["lr"], "perturb_th": None, "optimizer": "sgd"}
    eng_jitter.force_kernel("guide", kernel)
    try:
        mg, grad = eng_jitter.guidance_step(mean, cond, gd, sigma=0.5, want_grad=True)
        torch.cuda.synchronize()
    finally:
        eng_jitter.force_kernel("guide", "auto")
    _, gref = O.guidance_step(O.to_torch(synth.make_decoder_weights(0)), mean, cond, cs, None, None, c["lr"], None, "sgd", target_pos=(wp, wt, tps))
    assert (grad.cpu() - gref).abs().max().item() <= 1e-4 * gref.abs().max().item()
    step = np.abs(g["guided_targetpos_sgd"] - mean.numpy()).max()
    assert np.abs(mg.cpu().numpy() - g["guided_targetpos_sgd"]).max() <= max(1e-4 * step, 2.5e-7)


def test_policy_guidance_from_upstream_config(eng10):
    """CldPolicy.set_guidance with upstream-style per-scene guidance lists, two samples per agent: the guided action equals the
    engine's guided chain on the repeated batch (bit for bit) and differs from the unguided one."""
    from cld_amd.dm_model import DmModel
    from cld_amd.policy import CldPolicy
    from cld_amd.vae_model import VaeModel
    dm, vae = DmModel(None, None, n_timesteps=10, engine=eng10), VaeModel(engine=eng10)
    pol = CldPolicy(dm, vae)
    B, N = 4, 2
    inp = synth.make_inputs(B, 9)
    cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
    tgt = synth.uniform(9, "tgt", (B, 52), 0.0, 12.0)
    cfg = [[{"name": "target_speed", "weight": 1.0, "params": {"target_speed": tgt}, "agents": None}],
           [{"name": "target_pos_at_time", "weight": 1.0, "params": {"target_pos": [[20.0, 1.0], [15.0, -2.0]], "target_time": [40, 51]}, "agents": None}]]
    pol.set_guidance(cfg, torch.tensor([0, 0, 1, 1]), lr=0.3, optimizer="adam")
    nz = synth.make_noise(B * N, 10, 3)
    noise = {"x_T": torch.from_numpy(nz["x_T"]), "noise": torch.from_numpy(nz["noise"])}
    act, info = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise)
    assert info["trajectories"].shape == (B, N, 52, 6)
    rep = lambda t: t.repeat_interleave(N, dim=0)
    g = pol._guidance
    gd = {"curr_states": rep(cs), "target_speed": rep(g["target_speed"]), "loss_scale": rep(g["loss_scale"]),
          "target_pos": tuple(rep(v) for v in g["target_pos"]), "lr": 0.3, "optimizer": "adam"}
    x0, _, _ = eng10.sample(noise["x_T"], rep(cond), noise=noise["noise"], guidance=gd)
    traj = eng10.decode(x0, rep(cond), rep(cs), descaled_output=True).reshape(B, N, 52, 6)
    assert torch.equal(info["trajectories"], traj)
    pol.clear_guidance()
    act0, _ = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise)
    assert not torch.equal(act0.positions, act.positions)


def test_non_cond_feat_for_classifier_free_guidance(eng_ctx):
    """aux_info['non_cond_feat'] as upstream builds it (diffuser.py:390-411,459-471): the combine MLP on the agent's own state
    features and the map features of a raster filled with -1 -- against the oracle run on that raster, and equal to a full
    context_encode of the filled raster."""
    from cld_amd.context_utils import ContextEncoder
    from oracle import cld_oracle as O
    B = 5
    batch = {"history_positions": torch.zeros(B, 31, 2), "history_yaws": torch.zeros(B, 31, 1),
             "curr_speed": torch.from_numpy(synth.uniform(8, "curr_speed", (B,), 0.0, 15.0)),
             "image": torch.from_numpy(synth.make_raster(B, 8)).cuda()}
    aux = ContextEncoder(eng_ctx)(batch, include_class_free_cond=True)
    assert aux["non_cond_feat"].shape == (B, 256)
    filled = torch.full((B, 34, 224, 224), -1.0)
    full = eng_ctx.context_encode(filled.cuda(), aux["curr_states"])
    assert (aux["non_cond_feat"] - full).abs().max().item() <= 1e-5
    ref = O.context_encode(O.to_torch(synth.make_context_weights(0)), filled[:1].expand(B, -1, -1, -1), aux["curr_states"].cpu())
    assert (aux["non_cond_feat"].cpu() - ref).abs().max().item() <= 1e-4
    assert (aux["non_cond_feat"] - aux["cond_feat"]).abs().max().item() > 1e-2


@pytest.mark.parametrize("kernel", ["valu", "mfma", "quad"])
def test_decode_vjp_vs_autograd(kernel):
    """Engine.decode_vjp = J^T g for J = d decode / d z (decoder + descale + unicycle roll-out, all six trajectory channels,
    yaw-rate clip active on part of the steps): against torch autograd through the oracle's decode for a random g."""
    import os
    from cld_amd.engine import Engine
    from oracle import cld_oracle as O
    mean6, std6 = list(O.NORM_MEAN), list(O.NORM_STD)
    std6[5] = 2.0
    e = Engine(n_timesteps=10, device="cuda:0", norm_info=(mean6, std6))
    e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
    B = 20
    inp = synth.make_inputs(B, 41)
    cond = torch.from_numpy(inp["cond_feat"])
    cs = torch.from_numpy(inp["curr_states"]).clone()
    cs[:, 2] = torch.from_numpy(synth.uniform(41, "slow", (B,), 0.05, 8.0))
    z = torch.from_numpy(synth.normal(41, "z", (B, 52, 4)))
    G = torch.from_numpy(synth.normal(41, "G", (B, 52, 6)))
    e.force_kernel("guide", kernel)
    old = O.NORM_STD
    try:
        got = e.decode_vjp(z, cond, cs, G).cpu()
        O.NORM_STD = tuple(std6)
        zz = z.clone().requires_grad_(True)
        traj = O.decode(O.to_torch(synth.make_decoder_weights(0)), zz, cond, cs, True)
        (ref,) = torch.autograd.grad((traj * G).sum(), zz)
    finally:
        O.NORM_STD = old
        e.force_kernel("guide", "auto")
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_decode_fn_autograd_with_a_cross_agent_loss(eng_jitter):
    """vae_model.DecodeFn: a loss that couples agents (pairwise proximity penalty, the shape of upstream's AgentCollisionLoss)
    written in torch on the decoded trajectories, differentiated by torch autograd down to dL/dtraj and pulled back to the
    latent by the HIP vector-Jacobian product -- against autograd through the oracle's decode."""
    from cld_amd.vae_model import DecodeFn
    from oracle import cld_oracle as O
    B = 12
    inp = synth.make_inputs(B, 43)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    z = torch.from_numpy(synth.normal(43, "z", (B, 52, 4)))
    off = torch.from_numpy(synth.normal(43, "off", (B, 1, 2))) * 3.0          # agents start close to each other

    def loss_fn(traj, offset):
        p = traj[..., :2] + offset
        d = (p[:, None] - p[None]).norm(dim=-1) + torch.eye(p.shape[0], device=p.device)[..., None] * 1e3
        return torch.relu(1.0 - d / 4.0).sum() + 0.01 * traj[..., 3].pow(2).sum()

    zg = z.cuda().requires_grad_(True)
    loss_fn(DecodeFn.apply(zg, cond.cuda(), cs.cuda(), eng_jitter), off.cuda()).backward()
    zr = z.clone().requires_grad_(True)
    loss_fn(O.decode(O.to_torch(synth.make_decoder_weights(0)), zr, cond, cs, True), off).backward()
    assert float(zr.grad.abs().max()) > 0
    assert (zg.grad.cpu() - zr.grad).abs().max().item() <= 2e-4 * zr.grad.abs().max().item()


def test_sample_with_caller_defined_loss(eng10):
    """Engine.sample_with_loss: the guidance loss is torch code on the decoded trajectory (here TargetSpeedLoss + a term
    that couples agents), its gradient is pulled back by the HIP vector-Jacobian product.  With the target-speed loss alone the
    chain must reproduce the built-in guided sampler (same loss, same Adam step) up to the sign flips of tiny gradients."""
    B, n = 6, 10
    inp = synth.make_inputs(B, 3)
    nz = synth.make_noise(B, n, 5)
    cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
    tgt = torch.from_numpy(synth.uniform(3, "tgt", (B, 52), 0.0, 12.0)).cuda()
    x_T, z = torch.from_numpy(nz["x_T"]).cuda(), torch.from_numpy(nz["noise"]).cuda()
    loss = lambda tr: (tr[..., 2] - tgt).abs().mean(dim=1).sum()
    # SGD (no sign function in the step): the two routes to the same gradient must agree end to end on all elements
    builtin, _, _ = eng10.sample(x_T, cond, noise=z, guidance={"curr_states": cs, "target_speed": tgt, "lr": 2000.0, "optimizer": "sgd"})
    custom, x1 = eng10.sample_with_loss(x_T, cond, cs, z, loss, lr=2000.0, optimizer="sgd")
    scale = max(1.0, float(builtin.abs().max()))
    assert float((custom - builtin).abs().max()) <= 1e-3 * scale
    assert x1 is not None
    # Adam: step by step on the built-in chain's x_t -- the torch-loss route (decode -> autograd -> vector-Jacobian product) must give
    # the built-in gradient to 2e-5 and a guided mean within what that tolerance lets Adam's sign-like step do, every element
    from oracle import cld_oracle as O
    gd = {"curr_states": cs, "target_speed": tgt, "lr": 0.3, "optimizer": "adam"}
    x = x_T
    for it in range(n):
        i = n - 1 - it
        got = eng10.sample_step(x, cond, i, z=z[it], guidance=gd, want_grad=True)
        if i > 0:
            traj = eng10.decode(got["mean"], cond, cs, descaled_output=True).requires_grad_(True)
            with torch.enable_grad():
                (gtraj,) = torch.autograd.grad(loss(traj), traj)
            mg, gr = eng10.guidance_step(got["mean"], cond, {"curr_states": cs, "ext_grad": gtraj, "lr": 0.3, "optimizer": "adam"},
                                         sigma=got["sigma"], want_grad=True)
            gtol = 2e-5 * float(got["grad"].abs().max())
            assert float((gr - got["grad"]).abs().max()) <= gtol, i
            over = (mg - got["mean_guided"]).abs().cpu() - O.adam_step_budget(got["grad"].cpu(), 0.3, gtol) - 1e-6 * max(1.0, float(got["mean"].abs().max()))
            assert float(over.max()) <= 0.0, (i, float(over.max()))
        x = got["x_next"]
    custom, x1 = eng10.sample_with_loss(x_T, cond, cs, z, loss, lr=0.3, optimizer="adam")
    assert x1 is not None and bool(torch.isfinite(custom).all())
    # a loss the kernels do not know: keep agents apart (cross-agent term) -- runs, stays finite, changes the sample
    def apart(tr):
        p = tr[..., :2]
        dist = (p[:, None] - p[None]).norm(dim=-1) + torch.eye(B, device=p.device)[..., None] * 1e3
        return torch.relu(1.0 - dist / 5.0).sum()
    other, _ = eng10.sample_with_loss(x_T, cond, cs, z, apart, lr=0.3)
    plain, _, _ = eng10.sample(x_T, cond, noise=z)
    assert bool(torch.isfinite(other).all()) and not torch.equal(other, plain)


def test_sample_with_stride_golden(golden):
    """DmModel.stride (the reference's attribute, dm_model.py:25,119): stride 4 on the 100-step schedule against the reference's
    own sampler; x1 is None as in the reference; a wrong slab count is refused."""
    from cld_amd._lib import CldError
    from cld_amd.dm_model import DmModel
    meta, g = golden("sample_n100_stride4")
    B, n, st = meta["B"], meta["n_timesteps"], meta["stride"]
    dm = DmModel(None, None, n_timesteps=n, engine=_engine(n, meta["affine_jitter"], decoder=False))
    dm.stride = st
    assert dm.stride == st and dm.engine.loop_steps == 25
    nz = synth.make_noise(B, 25, meta["noise_seed"])
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"]).cuda()
    out = dm({"history_positions": torch.zeros(B, 31, 2)}, {"cond_feat": cond}, {"num_samp": 1},
             noise={"x_T": torch.from_numpy(nz["x_T"]), "noise": torch.from_numpy(nz["noise"])})
    scale = float(np.abs(g["pred_traj"]).max())
    assert out["x1"] is None
    assert np.abs(out["pred_traj"].cpu().numpy() - g["pred_traj"]).max() <= 1e-3 * max(1.0, scale)
    assert np.abs(out["log_prob_final"].cpu().numpy() - g["log_prob_final"]).max() <= 1e-4
    with pytest.raises(CldError):
        dm.engine.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.zeros(100, B, 52, 4))


def test_compute_losses_golden(golden, eng_jitter):
    """DmModel.compute_losses / q_sample / per-row timesteps of `dm.model` against the reference (golden 'compute_losses')."""
    from cld_amd.dm_model import DmModel
    from oracle import cld_oracle as O
    meta, g = golden("compute_losses")
    B = meta["B"]
    dm = DmModel(None, None, n_timesteps=100, engine=eng_jitter)
    z0 = torch.from_numpy(synth.normal(meta["in_seed"], "loss_z0", (B, 52, 4))).cuda()
    noise = torch.from_numpy(synth.normal(meta["noise_seed"], "loss_noise", (B, 52, 4))).cuda()
    t = torch.from_numpy(g["t"]).cuda()
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"]).cuda()
    assert np.abs(dm.q_sample(z0, t, noise).cpu().numpy() - g["z_noisy"]).max() <= 1e-6
    loss = float(dm.compute_losses({"cond_feat": cond}, z0, t=t, noise=noise))
    assert abs(loss - float(g["loss"][0])) <= 1e-4 * max(1.0, float(g["loss"][0]))
    # per-row timesteps through dm.model equal the single-timestep path row by row
    eps = dm.model(torch.from_numpy(g["z_noisy"]).cuda(), {"cond_feat": cond}, t)
    for b in (0, 5, 15):
        one = eng_jitter.unet_forward(torch.from_numpy(g["z_noisy"][b:b + 1]).cuda(), cond[b:b + 1], int(t[b]))
        assert (eps[b] - one[0]).abs().max().item() <= 2e-5 * max(1.0, one.abs().max().item())
    assert float(dm.compute_losses({"cond_feat": cond}, z0)) > 0.0          # own draws


def test_vae_loss_golden(golden):
    """VaeModel.compute_vae_loss forward (recon MSE + beta * KLD) against the reference (golden 'vae_loss')."""
    from cld_amd.engine import Engine
    from cld_amd.vae_model import VaeModel
    meta, g = golden("vae_loss")
    _, ge = golden("encode")
    B = meta["B"]
    e = Engine(n_timesteps=10, device="cuda:0")
    for sd in (synth.make_unet_weights(0), synth.make_decoder_weights(0), synth.make_encoder_weights(0)):
        e.load_state_dict(sd)
    e.finalize()
    vae = VaeModel(engine=e)
    cond = torch.from_numpy(synth.make_inputs(B, 1)["cond_feat"]).cuda()
    fut = synth.make_future(B, 1)
    x6s = e.state_to_state_and_action(fut["target_positions"], fut["target_yaws"], fut["curr_speed"], scaled_output=True)
    act = vae.lstmvae.lstm_dec(torch.from_numpy(ge["z"]), cond)
    loss, recon, kld = vae.compute_vae_loss(x6s, act, torch.from_numpy(ge["mu"]), torch.from_numpy(ge["logvar"]), meta["beta"])
    got = np.array([float(loss), float(recon), float(kld)])
    assert np.abs(got - g["loss"]).max() <= 1e-4 * max(1.0, np.abs(g["loss"]).max())


def test_decode_mfma_kernel_vs_oracle_and_valu(eng_jitter):
    """The 16-agents-per-workgroup MFMA formulation of the decoder (taken from 256 agents up): against the oracle at B = 300
    (ragged last tile) with the decode bars (actions 2e-5, trajectories 1e-4), and against the one-agent-per-workgroup kernel."""
    import os
    from oracle import cld_oracle as O
    B = 300
    inp = synth.make_inputs(B, 51)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    z = torch.from_numpy(synth.normal(51, "z", (B, 52, 4)))
    outs = {}
    for k in ("mfma", "valu"):
        eng_jitter.force_kernel("decode", k)
        try:
            outs[k] = eng_jitter.decode(z, cond, cs, descaled_output=True, want_act=True)
            torch.cuda.synchronize()
        finally:
            eng_jitter.force_kernel("decode", "auto")
    wd = O.to_torch(synth.make_decoder_weights(0))
    torch.set_num_threads(8)
    act_ref = O.lstm_decode(wd, z, cond)
    traj_ref = O.decode(wd, z, cond, cs, descaled_output=True)
    for k in ("mfma", "valu"):
        traj, act = outs[k]
        assert (act.cpu() - act_ref).abs().max().item() <= 2e-5, k
        assert (traj.cpu() - traj_ref).abs().max().item() <= 1e-4, k
    assert (outs["mfma"][0] - outs["valu"][0]).abs().max().item() <= 1e-4


def test_encode_mfma_kernel_vs_oracle_and_valu():
    """The MFMA formulation of the VAE encoder (from 256 agents up) at B = 300 (ragged last tile): z / mu / logvar against the
    oracle (2e-5, the encoder bar) and against the one-agent-per-workgroup kernel."""
    import os
    from cld_amd.engine import Engine
    from oracle import cld_oracle as O
    e = Engine(n_timesteps=10, device="cuda:0")
    for sd in (synth.make_unet_weights(0), synth.make_encoder_weights(0)):
        e.load_state_dict(sd)
    e.finalize()
    B = 300
    cond = torch.from_numpy(synth.make_inputs(B, 61)["cond_feat"])
    x6 = torch.from_numpy(synth.normal(61, "x6", (B, 52, 6)))
    nz = torch.from_numpy(synth.normal(61, "nz", (B, 52, 4)))
    outs = {}
    for k in ("mfma", "valu"):
        e.force_kernel("encode", k)
        try:
            outs[k] = e.traj2z(x6, cond, nz)
            torch.cuda.synchronize()
        finally:
            e.force_kernel("encode", "auto")
    torch.set_num_threads(8)
    ref = O.traj2z(O.to_torch(synth.make_encoder_weights(0)), x6, cond, nz)
    for k in ("mfma", "valu"):
        for got, want in zip(outs[k], ref):
            assert (got.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), k



# ---------------------------------------------------------------------------------------------------------
# round 2: schedule buffers, the absolute bar, configs[4] pieces, the get_action contract
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [10, 50, 100])
def test_schedule_buffers_golden(golden, n):
    """a-1 directly: the three buffers the sampler reads (dm_model.py:29-56), rebuilt in C++ by cld_create and returned by
    cld_get_schedule, against the reference's own buffers: <= 1 ulp (in fact bit-identical except one x_t_cof entry at n = 50,
    where torch's vectorised sqrt and the correctly rounded one differ in the last bit).  What it takes: the cumulative product
    accumulated in double like torch.cumprod does on the CPU, and the logarithm taken in double."""
    from cld_amd.engine import Engine
    _, g = golden(f"schedule_n{n}")
    e = Engine(n_timesteps=n, device="cuda:0")
    for name, ulps in (("x_t_cof", 1), ("noise_cof", 1), ("posterior_log_variance_clipped", 1)):
        got, ref = getattr(e, name), g[name]
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.all(np.abs(got - ref) <= ulps * np.spacing(np.abs(ref))), (name, np.abs(got - ref).max())
    assert e.posterior_log_variance_clipped[0] == np.float32(np.log(np.float32(1e-20)))       # the 1e-20 clamp -> sigma_0 = 1e-10


def test_small_chain_absolute_bar(golden):
    """north_star's literal bar end to end: a 100-step chain recorded from the reference whose x0 stays O(1) (output layer and
    inputs scaled down, fixture sample_n100_small), so `<= 1e-3 per latent element` is an ABSOLUTE check here."""
    from cld_amd.engine import Engine
    from tests.test_oracle_golden import small_chain_inputs
    meta, g = golden("sample_n100_small")
    w, x_T, noise = small_chain_inputs(meta)
    e = Engine(n_timesteps=meta["n_timesteps"], device="cuda:0", precision=PRECISION)
    e.load_state_dict(w)
    e.finalize()
    cond = torch.from_numpy(synth.make_inputs(meta["B"], meta["in_seed"])["cond_feat"])
    x0, x1, logp = e.sample(x_T, cond, noise=noise)
    scale = float(np.abs(g["pred_traj"]).max())
    assert scale <= 10.0
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"small chain {k}: max|d|={err:.3e} max|ref|={scale:.3e}")
        assert err <= 1e-3
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


def test_guidance_loss_values_golden(golden, eng_jitter):
    """cld_guidance_losses against the reference's own loss classes (fixture guide_losses) and NaN for the terms that are off."""
    meta, g = golden("guide_losses")
    B, N = meta["B"], meta["N"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "gl_traj", (B, N, 52, 6))) * torch.tensor([20.0, 5.0, 6.0, 0.5, 3.0, 0.2])
    tgt = torch.from_numpy(synth.uniform(meta["in_seed"], "gl_tgt", (B, 52), 0.0, 12.0))
    wp = torch.from_numpy(synth.uniform(meta["in_seed"], "gl_wp", (B, 2), -5.0, 25.0))
    rep = lambda v: v.repeat_interleave(N, dim=0)
    one = torch.ones(B * N)
    flat = x.reshape(B * N, 52, 6)
    a = eng_jitter.guidance_losses(flat, {"target_speed": rep(tgt), "speed_limit": (meta["speed_limit"], one), "acc_limit": (meta["acc_limit"], one),
                                          "target_pos": (rep(wp), rep(torch.tensor(meta["target_time"])), one)}).reshape(B, N, 4).cpu().numpy()
    m = int(meta["min_target_time"] * 52)
    b = eng_jitter.guidance_losses(flat, {"target_pos": (rep(wp), torch.full((B * N,), -(m + 1)), one)}).reshape(B, N, 4).cpu().numpy()
    for col, key in ((0, "target_speed"), (1, "speed_limit"), (2, "acc_limit"), (3, "target_pos_at_time")):
        assert np.abs(a[..., col] - g[key]).max() <= 2e-6 * max(1.0, np.abs(g[key]).max()), key
    assert np.abs(b[..., 3] - g["target_pos"]).max() <= 2e-5 * max(1.0, np.abs(g["target_pos"]).max())
    assert np.isnan(b[..., :3]).all()
    # a zero scale switches the term off for that agent
    half = one.clone(); half[::2] = 0.0
    c = eng_jitter.guidance_losses(flat, {"speed_limit": (4.0, half)}).cpu().numpy()
    assert np.isnan(c[::2, 1]).all() and not np.isnan(c[1::2, 1]).any()


def _policy(eng):
    from cld_amd.dm_model import DmModel
    from cld_amd.policy import CldPolicy
    from cld_amd.vae_model import VaeModel
    return CldPolicy(DmModel(None, None, n_timesteps=eng.n_timesteps, engine=eng), VaeModel(engine=eng))


def test_get_action_selects_the_sample_upstream_would(golden, eng10):
    """get_action with guidance active (algos.py:2053-2064): per-sample guidance losses come back as info['guide_losses'] (equal to
    the oracle's on the same trajectories), the executed sample is the oracle's `choose_action_from_guidance` of them -- whose
    behaviour is pinned by fixture `select` recorded from the reference's own function -- and `guide_as_filter_only` samples
    without guidance and only filters (algos.py:1815)."""
    from oracle import cld_oracle as O
    pol = _policy(eng10)
    B, N = 6, 4
    inp = synth.make_inputs(B, 13)
    cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
    tgt = synth.uniform(13, "tgt", (B, 52), 0.0, 12.0)
    cfg = [[{"name": "target_speed", "weight": 1.0, "params": {"target_speed": tgt}, "agents": None}],
           [{"name": "target_speed", "weight": 2.0, "params": {"target_speed": tgt}, "agents": None},
            {"name": "speed_limit", "weight": 1.0, "params": {"speed_limit": 5.0}, "agents": [0, 2]}]]
    scene_index = torch.tensor([0, 0, 1, 1, 1, 1])
    pol.set_guidance(cfg, scene_index, lr=0.3, optimizer="adam")
    nz = synth.make_noise(B * N, 10, 5)
    noise = {"x_T": torch.from_numpy(nz["x_T"]), "noise": torch.from_numpy(nz["noise"])}
    act, info = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise, step_index=0)
    traj = info["trajectories"]
    gl = info["guide_losses"]
    assert list(gl) == ["target_speed_scene_000_00", "target_speed_scene_001_00", "speed_limit_scene_001_01"]
    # loss values vs the oracle on the same decoded trajectories; NaN outside each loss's agents
    tr = traj.reshape(B * N, 52, 6).cpu()
    rep = lambda v: torch.as_tensor(v).repeat_interleave(N, dim=0)
    ref = O.guidance_losses(tr, rep(tgt), None, (5.0, torch.ones(B * N)), None, None).reshape(B, N, 4)
    for key, col, members in (("target_speed_scene_000_00", 0, [0, 1]), ("target_speed_scene_001_00", 0, [2, 3, 4, 5]),
                              ("speed_limit_scene_001_01", 1, [2, 4])):
        got = gl[key].cpu()
        out = [b for b in range(B) if b not in members]
        assert bool(torch.isnan(got[out]).all()), key
        assert float((got[members] - ref[members, :, col]).abs().max()) <= 1e-5 * max(1.0, float(ref[members, :, col].abs().max())), key
    want = O.choose_action_from_guidance({k: v.cpu() for k, v in gl.items()}, [["target_speed"], ["target_speed", "speed_limit"]])
    assert torch.equal(info["act_idx"].cpu(), want)
    assert bool((want[:2] == 0).all()) and bool((want[2:] != 0).any())           # upstream: the last scene's choice, sample 0 elsewhere
    ar = torch.arange(B)
    assert torch.equal(act.positions.cpu(), traj[..., :2].cpu()[ar, want]) and torch.equal(act.yaws.cpu(), traj[..., 3:4].cpu()[ar, want])
    # the evident intent (every scene chooses for its own agents) is available as an option
    pol.select_per_scene = True
    _, info2 = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise)
    s0 = torch.nansum(torch.stack([gl["target_speed_scene_000_00"]], 2), 2).argmin(dim=-1).cpu()
    assert torch.equal(info2["act_idx"].cpu()[:2], s0[:2]) and torch.equal(info2["act_idx"].cpu()[2:], want[2:])
    pol.select_per_scene = False
    # filter only: the samples are the UNGUIDED chain's, the selection still follows the guidance losses
    _, info3 = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise, guide_as_filter_only=True)
    pol.clear_guidance()
    _, info0 = pol.get_action({"cond_feat": cond, "curr_states": cs}, num_action_samples=N, noise=noise)
    assert torch.equal(info3["trajectories"], info0["trajectories"]) and not torch.equal(info3["trajectories"], traj)
    assert "guide_losses" in info3 and "guide_losses" not in info0 and bool((info0["act_idx"] == 0).all())
    # the sample closest to the ground-truth future (algos.py:2055-2056; upstream's own helper raises NameError as written)
    gt = info0["trajectories"][:, 2, :40, :2].clone()                      # sample 2 is the truth for every agent ...
    avail = torch.ones(B, 40, dtype=torch.bool); avail[1] = False          # ... except agent 1, which has no valid step
    _, info4 = pol.get_action({"cond_feat": cond, "curr_states": cs, "target_positions": gt, "target_availabilities": avail},
                              num_action_samples=N, noise=noise, guide_with_gt=True)
    assert info4["act_idx"].cpu().tolist() == [2, 0, 2, 2, 2, 2]
    assert torch.equal(info4["act_idx"].cpu(), O.choose_action_from_gt(info0["trajectories"][..., :2].cpu(), gt.cpu(), avail))


def test_get_action_rejects_what_it_does_not_implement(eng10):
    """No silent **kwargs: options of upstream's get_action that are not built raise, unknown names are a TypeError, and
    classifier-free guidance without unconditional features is an error instead of an unguided run."""
    from cld_amd._lib import CldError
    pol = _policy(eng10)
    B = 3
    inp = synth.make_inputs(B, 2)
    obs = {"cond_feat": torch.from_numpy(inp["cond_feat"]).cuda(), "curr_states": torch.from_numpy(inp["curr_states"]).cuda()}
    with pytest.raises(NotImplementedError):
        pol.get_action(obs, guide_clean="video_diff")      # only the boolean form of guide_clean is built (tests/test_gpu_collision.py)
    with pytest.raises(NotImplementedError):
        pol.get_action(obs, plan=object())
    with pytest.raises(TypeError):
        pol.get_action(obs, not_an_option=1)
    with pytest.raises(CldError):                      # eng10 has no ContextEncoder weights to build non_cond_feat from
        pol.get_action(obs, class_free_guide_w=2.0)
    with pytest.raises(CldError):
        pol.dm({"history_positions": obs["cond_feat"]}, obs, {"num_samp": 1}, class_free_guide_w=2.0)
    with pytest.raises(CldError):                      # per-row timesteps are range-checked on the host
        eng10.unet_forward_rows(torch.zeros(B, 52, 4), obs["cond_feat"], torch.tensor([0, 3, 10]))


def test_guidance_on_the_output_step_vs_oracle(eng10):
    """upstream apply_guidance_output (diffuser.py:877-880): the t = 0 posterior mean takes one more optimiser step with
    final_step_opt_params and no noise follows -- against the oracle's autograd chain; `intermediate=False` leaves only that step."""
    from oracle import cld_oracle as O
    B, n = 6, 10
    inp = synth.make_inputs(B, 17)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    tgt = torch.from_numpy(synth.uniform(17, "tgt", (B, 52), 0.0, 12.0))
    nz = synth.make_noise(B, n, 6)
    xT, noise = torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"])
    gd = {"curr_states": cs, "target_speed": tgt, "lr": 0.3, "optimizer": "adam", "output": {"lr": 0.2, "optimizer": "adam", "perturb_th": 1.0}}
    x0, _, _ = eng10.sample(xT, cond, noise=noise, guidance=gd)
    torch.set_num_threads(8)
    w, wd = O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))
    ref = O.sample_guided(w, wd, O.schedule(n), xT, noise, cond, cs, tgt, lr=0.3, optimizer="adam", output={"lr": 0.2, "optimizer": "adam"})["pred_traj"]
    plain = O.sample_guided(w, wd, O.schedule(n), xT, noise, cond, cs, tgt, lr=0.3, optimizer="adam")["pred_traj"]
    scale = float(ref.abs().max())
    assert float((x0.cpu() - ref).abs().max()) <= 1e-3 * scale
    assert float((plain - ref).abs().max()) > 0.15                         # the output step did something: Adam's first step is lr = 0.2 per element
    only, _, _ = eng10.sample(xT, cond, noise=noise, guidance=dict(gd, intermediate=False))
    base, _, _ = eng10.sample(xT, cond, noise=noise)
    mean_g = eng10.guidance_step(base, cond, {"curr_states": cs, "target_speed": tgt, "lr": 0.2, "optimizer": "adam"}, sigma=0.0)
    assert float((only - mean_g).abs().max()) <= 1e-5 * float(base.abs().max())


def test_closed_loop_vs_oracle(eng10):
    """configs[4]'s loop at oracle size: encode-free planning loop (cond_fn supplies cond_feat) -> sample -> decode -> world update,
    3 sim steps of 8 agents, against oracle.closed_loop (env_utils.py:255-304 + env_trajdata.py:452-468)."""
    from cld_amd.policy import closed_loop_rollout
    from oracle import cld_oracle as O
    pol = _policy(eng10)
    B, n, S = 8, 10, 3
    inp = synth.make_inputs(B, 23)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    conds = [cond, cond.flip(0).contiguous(), (cond * 0.5).contiguous()]
    ctr = torch.from_numpy(synth.normal(23, "ctr", (B, 2))) * 50.0
    yaw = torch.from_numpy(synth.uniform(23, "yaw", (B,), -3.1, 3.1))
    nz = synth.make_noise(B, n, 8)
    xT, noise = torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"])
    poses = closed_loop_rollout(pol, lambda s, wld, c: conds[s].cuda(), ctr, yaw, cs, n_sim_steps=S,
                                noise={"x_T": xT, "noise": noise}).cpu()
    torch.set_num_threads(8)
    w, wd = O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))
    with torch.no_grad():
        ref = O.closed_loop(w, wd, O.schedule(n), lambda s, wld, c: conds[s], ctr, yaw, cs, xT, noise, S)
    # poses are O(100 m); the plan's positions carry the chain's relative error (1e-3 of max|x0| through the decoder is far below this)
    assert float((poses - ref).abs().max()) <= 2e-3, float((poses - ref).abs().max())
    assert float((poses[-1, :, :2] - ctr).norm(dim=-1).min()) > 0.0


def test_standalone_vae_model_without_unet_weights():
    """VaeModel() on its own engine (no DmModel): cld_finalize accepts a handle without the U-Net; decoder / encoder calls
    work and match the oracle, U-Net calls report the missing weights instead of crashing."""
    from cld_amd._lib import CldError
    from cld_amd.vae_model import VaeModel
    from oracle import cld_oracle as O
    vae = VaeModel(device="cuda:0")
    vae.load_state_dict(dict(synth.make_decoder_weights(0), **synth.make_encoder_weights(0)))
    B = 5
    inp = synth.make_inputs(B, 71)
    cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
    z = torch.from_numpy(synth.normal(71, "z", (B, 52, 4)))
    act = vae.lstmvae.lstm_dec(z, cond)
    traj = vae.convert_action_to_state_and_action(act, cs, descaled_output=True)
    wd = O.to_torch(synth.make_decoder_weights(0))
    assert float((act.cpu() - O.lstm_decode(wd, z, cond)).abs().max()) <= 2e-5
    assert float((traj.cpu() - O.decode(wd, z, cond, cs, True)).abs().max()) <= 1e-4
    with pytest.raises(CldError, match="U-Net weights"):
        vae.engine.unet_forward(z, cond, 3)
    with pytest.raises(CldError, match="U-Net weights"):
        vae.engine.sample(z, cond, noise=torch.zeros(100, B, 52, 4))
