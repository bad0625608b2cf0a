"""GPU parity of the layer chains (csrc/conv_chain.hip, csrc/chain_wino.hip): the 64-channel levels of a U-Net evaluation as two launches
whose tiles stay in LDS from layer to layer (exact-fp32 handles; the default at every batch size, with the 64 -> 64 k5 layers in Winograd F(4, 5) form: one-agent tiles up to
944 rows per launch set, two-agent tiles above).  Every golden case runs in the direct form ("chain1" / "chain4": one- / four-agent tiles) and with
the Winograd tiles of four, two and one agents forced ("chainw", "chainw2", "chainw1").  Same bars as
tests/test_gpu_parity.py: every case is run with the chains forced on at sizes the golden fixtures and the oracle cover, and the
automatic choice is checked at a launch size that takes it.
"""
import numpy as np
import pytest
import torch

from cld_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _f32_only(precision):
    if precision != "f32":
        pytest.skip("layer chains exist in the exact-fp32 mode only (the split-precision mode keeps one launch per layer)")


def _engine(n=100, jitter=True, form="chain"):
    from cld_amd.engine import Engine
    e = Engine(n_timesteps=n, device="cuda:0")
    e.load_state_dict(synth.make_unet_weights(0, affine_jitter=jitter))
    e.finalize()
    e.force_kernel("unet", form)
    return e


FORMS = ["chain1", "chainw", "chainw2", "chainw1"]     # direct one-agent tiles; Winograd tiles of four / two / one agents


@pytest.fixture(scope="module", params=FORMS)
def eng(request):
    return _engine(form=request.param)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("tag", ["default", "jitter"])
def test_chains_unet_forward_golden(golden, tag, form):
    meta, g = golden(f"unet_forward_{tag}")
    e = _engine(100, meta["affine_jitter"], form)
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "unet_x", (B, 52, 4))) * 3.0
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    for r, t in enumerate(meta["t"]):
        eps = e.unet_forward(x, cond, t).cpu().numpy()
        assert np.abs(eps[r] - g["eps"][r]).max() <= 2e-5, t


def test_chains_ragged_batch_and_per_row_timesteps_vs_oracle(eng):
    from oracle import cld_oracle as O
    B = 37                                           # 2.3 sixteen-agent tiles, 9.25 four-agent chain tiles
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=True))
    x = torch.from_numpy(synth.normal(7, "rag_x", (B, 52, 4))) * 2.0
    cond = torch.from_numpy(synth.make_inputs(B, 7)["cond_feat"])
    for t in (73, 3):
        ref = O.unet_forward(w, x, cond, torch.full((B,), t, dtype=torch.long)).numpy()
        got = eng.unet_forward(x, cond, t).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5
    tt = torch.from_numpy(np.random.default_rng(5).integers(0, 100, B))          # time half folded into the per-agent vectors
    ref = O.unet_forward(w, x, cond, tt).numpy()
    got = eng.unet_forward_rows(x, cond, tt).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5


def test_chains_ddpm_step_golden(golden, eng):
    meta, g = golden("ddpm_step")
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "step_x", (B, 52, 4)))
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    z = torch.from_numpy(synth.normal(meta["noise_seed"], "step_z", (B, 52, 4)))
    for i in meta["t"]:
        xn, mean, _ = eng.ddpm_step(x, cond, i, z)
        scale = max(1.0, float(np.abs(g[f"mean_t{i}"]).max()))
        assert np.abs(mean.cpu().numpy() - g[f"mean_t{i}"]).max() <= 1e-4 * scale
        assert np.abs(xn.cpu().numpy() - g[f"x_next_t{i}"]).max() <= 1e-4 * scale


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n,jitter", [(10, True), (50, True), (100, True)])
def test_chains_full_chain_golden(golden, n, jitter, form):
    meta, g = golden(f"sample_n{n}_{'jitter' if jitter else 'default'}")
    e = _engine(n, jitter, form)
    B = meta["B"]
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, logp = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]))
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"chains: n={n} {k}: max|d|={err:.3e} max|ref|={scale:.3e} rel={err/scale:.2e}")
        assert err <= 1e-3 * scale
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


@pytest.mark.parametrize("form", FORMS)
def test_chains_small_chain_absolute_bar(golden, form):
    """north_star's literal bar (<= 1e-3 ABSOLUTE per latent element over 100 steps) with the chains on."""
    from cld_amd.engine import Engine
    from tests.test_oracle_golden import small_chain_inputs
    meta, g = golden("sample_n100_small")
    w, x_T, noise = small_chain_inputs(meta)
    e = Engine(n_timesteps=meta["n_timesteps"], device="cuda:0")
    e.load_state_dict(w)
    e.finalize()
    e.force_kernel("unet", form)
    cond = torch.from_numpy(synth.make_inputs(meta["B"], meta["in_seed"])["cond_feat"])
    x0, x1, logp = e.sample(x_T, cond, noise=noise)
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"chains ({form}): absolute-bar chain {k}: max|d| = {err:.3e}")
        assert err <= 1e-3
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


@pytest.mark.parametrize("form", FORMS)
def test_chains_cfg_golden(golden, form):
    """CFG: the head combines the two halves of the chain's noise prediction [2B,52,4]."""
    meta, g = golden("sample_cfg_n10")
    B, n = meta["B"], meta["n_timesteps"]
    e = _engine(n, True, form)
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    non_cond = torch.from_numpy(synth.normal(meta["in_seed"], "non_cond_feat", (B, 256)))
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]),
                         non_cond=non_cond, guidance_w=meta["guidance_w"])
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        assert float(np.abs(got.cpu().numpy() - g[k]).max()) <= 1e-3 * scale
    a, _, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]))
    b, _, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]), non_cond=non_cond, guidance_w=0.0)
    assert torch.equal(a, b)           # w = 0 reproduces the plain chain bit for bit (a tile's result does not depend on the batch around it)


@pytest.mark.parametrize("B", [600, 1024, 2100])
def test_chains_are_the_default_and_agree_with_the_layer_launches(B):
    from oracle import cld_oracle as O
    e = _engine(100, True, "auto")
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 52, 4, generator=g) * 2.0
    cond = torch.randn(B, 256, generator=g)
    auto = e.unet_forward(x, cond, 41).clone()
    e.force_kernel("unet", "chain")
    forced = e.unet_forward(x, cond, 41).clone()
    e.force_kernel("unet", "layers")
    layers = e.unet_forward(x, cond, 41).clone()
    assert torch.equal(auto, forced)
    assert not torch.equal(auto, layers)             # two different kernels ran (their GroupNorm sums associate differently)
    assert float((auto - layers).abs().max()) <= 1e-5
    tiles = {}
    for form in ("chain1", "chain4", "chainw", "chainw2", "chainw1"):      # one- and four-agent direct tiles; Winograd tiles of four / two / one agents
        e.force_kernel("unet", form)
        tiles[form] = e.unet_forward(x, cond, 41).clone()
    assert torch.equal(auto, tiles["chainw1" if B <= 944 else "chainw2"])      # the library's own choice: Winograd tiles of one agent up to 944 rows, of two above
    assert float((tiles["chain1"] - tiles["chain4"]).abs().max()) <= 1e-5
    assert not torch.equal(tiles["chainw"], tiles["chain4"])
    dw = float((tiles["chainw"] - tiles["chain4"]).abs().max())
    print(f"Winograd chain vs direct chain at {B} rows: max|d eps| = {dw:.3e}")
    assert dw <= 1e-5
    for form in ("chainw2", "chainw1"):              # the same Winograd layers on smaller tiles: every agent's sums are formed in the same order
        assert torch.equal(tiles[form], tiles["chainw"]), form       # whatever the tile -- the result does not depend on the tile size, bit for bit
    rows = torch.tensor([0, 3, 4, B // 2 + 1, B - 2, B - 1])      # first / last tiles of the launch, both sides of a tile boundary
    ref = O.unet_forward(O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), x[rows], cond[rows],
                         torch.full((len(rows),), 41, dtype=torch.long))
    assert float((auto.cpu()[rows] - ref).abs().max()) <= 2e-5
    # rows of the big launch re-run as their own small batch (other tilings of the 128- / 256-channel layers at that size)
    e.force_kernel("unet", "chain")
    small = e.unet_forward(x[rows], cond[rows], 41)
    assert float((small.cpu() - auto.cpu()[rows]).abs().max()) <= 1e-5


def test_chains_on_device_noise_matches_the_layer_launches():
    """noise=None: the counter-based on-device generator is a function of (seed, step, row) only, so the tail chain -- which applies
    the DDPM update itself on plain steps and draws its own z -- and the head kernel behind one launch per layer walk the same
    chain; deterministic per seed, different across seeds."""
    e = _engine(10, True, "auto")
    B = 40
    g = torch.Generator().manual_seed(3)
    xT, cond = torch.randn(B, 52, 4, generator=g), torch.randn(B, 256, generator=g)
    a, a1, _ = e.sample(xT, cond, noise=None, seed=7)
    b, _, _ = e.sample(xT, cond, noise=None, seed=7)
    c, _, _ = e.sample(xT, cond, noise=None, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    e.force_kernel("unet", "layers")
    l, l1, _ = e.sample(xT, cond, noise=None, seed=7)
    scale = float(l.abs().max())
    assert float((a - l).abs().max()) <= 1e-3 * scale and float((a1 - l1).abs().max()) <= 1e-3 * float(l1.abs().max())
