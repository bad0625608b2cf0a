"""GPU parity of the Winograd F(4, 5) form of the 256 -> 256 Conv1d(k5) + GroupNorm + Mish launches (csrc/wino1d_kernels.hip; seven of
the 22 launches of a U-Net evaluation, half its FLOPs; exact-fp32 handles, the default from 384 rows per launch set).  Same bars as
tests/test_gpu_parity.py: every case runs with the form forced on at sizes the golden fixtures and the oracle cover, and the automatic
choice is checked at launch sizes that take it.
"""
import numpy as np
import pytest
import torch

from cld_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _f32_only(precision):
    if precision != "f32":
        pytest.skip("the Winograd form exists in the exact-fp32 mode only (the split-precision mode keeps its own loop)")


def _engine(n=100, jitter=True, form="winograd"):
    from cld_amd.engine import Engine
    e = Engine(n_timesteps=n, device="cuda:0")
    e.load_state_dict(synth.make_unet_weights(0, affine_jitter=jitter))
    e.finalize()
    e.force_kernel("conv5", form)
    return e


# "winograd": half items (wino1d_kernels.hip) at these sizes; "winograd_whole": wino1d_edge.hip's whole items, which a launch takes by itself
# only from two workgroups per CU (the large-batch cases below) -- forced here so that the golden fixtures and the oracle cover that kernel too
ITEM_FORMS = ["winograd", "winograd_whole", "winograd_ksplit"]


@pytest.fixture(scope="module", params=ITEM_FORMS)
def eng(request):
    return _engine(form=request.param)


@pytest.mark.parametrize("form", ITEM_FORMS)
@pytest.mark.parametrize("tag", ["default", "jitter"])
def test_winograd_unet_forward_golden(golden, tag, form):
    meta, g = golden(f"unet_forward_{tag}")
    e = _engine(100, meta["affine_jitter"], form)
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "unet_x", (B, 52, 4))) * 3.0
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    for r, t in enumerate(meta["t"]):
        eps = e.unet_forward(x, cond, t).cpu().numpy()
        assert np.abs(eps[r] - g["eps"][r]).max() <= 2e-5, t


def test_winograd_ragged_batch_and_per_row_timesteps_vs_oracle(eng):
    from oracle import cld_oracle as O
    B = 37                                           # 2.3 sixteen-agent workgroups
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=True))
    x = torch.from_numpy(synth.normal(7, "rag_x", (B, 52, 4))) * 2.0
    cond = torch.from_numpy(synth.make_inputs(B, 7)["cond_feat"])
    for t in (73, 3):
        ref = O.unet_forward(w, x, cond, torch.full((B,), t, dtype=torch.long)).numpy()
        got = eng.unet_forward(x, cond, t).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5
    tt = torch.from_numpy(np.random.default_rng(5).integers(0, 100, B))
    ref = O.unet_forward(w, x, cond, tt).numpy()
    got = eng.unet_forward_rows(x, cond, tt).cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5


def test_winograd_ddpm_step_golden(golden, eng):
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


@pytest.mark.parametrize("form", ITEM_FORMS)
@pytest.mark.parametrize("n,jitter", [(10, True), (50, True), (100, True)])
def test_winograd_full_chain_golden(golden, n, jitter, form):
    meta, g = golden(f"sample_n{n}_{'jitter' if jitter else 'default'}")
    e = _engine(n, jitter, form)
    B = meta["B"]
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, logp = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]))
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"winograd: n={n} {k}: max|d|={err:.3e} max|ref|={scale:.3e} rel={err/scale:.2e}")
        assert err <= 1e-3 * scale
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


@pytest.mark.parametrize("form", ITEM_FORMS)
def test_winograd_small_chain_absolute_bar(golden, form):
    """north_star's literal bar (<= 1e-3 ABSOLUTE per latent element over 100 steps) with the Winograd form on."""
    from cld_amd.engine import Engine
    from tests.test_oracle_golden import small_chain_inputs
    meta, g = golden("sample_n100_small")
    w, x_T, noise = small_chain_inputs(meta)
    e = Engine(n_timesteps=meta["n_timesteps"], device="cuda:0")
    e.load_state_dict(w)
    e.finalize()
    e.force_kernel("conv5", form)
    cond = torch.from_numpy(synth.make_inputs(meta["B"], meta["in_seed"])["cond_feat"])
    x0, x1, logp = e.sample(x_T, cond, noise=noise)
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        err = float(np.abs(got.cpu().numpy() - g[k]).max())
        print(f"winograd: small chain {k}: max|d|={err:.3e}")
        assert err <= 1e-3
    assert np.allclose(logp.cpu().numpy(), g["log_prob_final"], atol=1e-4)


def test_winograd_cfg_golden(golden):
    meta, g = golden("sample_cfg_n10")
    B, n = meta["B"], meta["n_timesteps"]
    e = _engine(n, True)
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    non_cond = torch.from_numpy(synth.normal(meta["in_seed"], "non_cond_feat", (B, 256)))
    nz = synth.make_noise(B, n, meta["noise_seed"])
    x0, x1, _ = e.sample(torch.from_numpy(nz["x_T"]), cond, noise=torch.from_numpy(nz["noise"]),
                         non_cond=non_cond, guidance_w=meta["guidance_w"])
    for got, k in ((x0, "pred_traj"), (x1, "x1")):
        scale = float(np.abs(g[k]).max())
        assert float(np.abs(got.cpu().numpy() - g[k]).max()) <= 1e-3 * scale


@pytest.mark.parametrize("B", [300, 600, 1024, 2100, 3200, 3205, 4096])
def test_winograd_is_the_default_for_large_launch_sets_and_agrees_with_the_direct_form(B):
    from oracle import cld_oracle as O
    e = _engine(100, True, "auto")
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 52, 4, generator=g) * 2.0
    cond = torch.randn(B, 256, generator=g)
    auto = e.unet_forward(x, cond, 41).clone()
    e.force_kernel("conv5", "winograd")
    wino = e.unet_forward(x, cond, 41).clone()
    e.force_kernel("conv5", "direct")
    direct = e.unet_forward(x, cond, 41).clone()
    # (rows are padded to 16; launches of fewer than 512 whole items run as half items: up to 2,100 rows every instance does; at 3,200 rows
    # the 256-channel launches run 800 whole items -- a full generation of 512 and a partly filled one --, at 4,096 rows 1,024 = two full ones)
    if B >= 384:
        assert torch.equal(auto, wino)
    else:      # below 384 rows the L = 13 / 26 launches take the direct form, the layer chains the Winograd form (at every size, round 4): a mix of the two forced runs
        assert not torch.equal(auto, wino) and not torch.equal(auto, direct)
        assert float((auto - wino).abs().max()) <= 1e-5 and float((auto - direct).abs().max()) <= 1e-5
    assert not torch.equal(wino, direct)             # two different kernels ran
    e.force_kernel("conv5", "winograd_whole")
    whole = e.unet_forward(x, cond, 41).clone()
    if B == 4096:                                    # 512 whole items or more in every k5 launch of the two levels: what the size rule takes is the whole-item kernel
        assert torch.equal(whole, wino)
        perm = torch.randperm(B, generator=g)        # a row's result does not depend on its place in a workgroup (16 agent rows; at L = 26 a pair of rows per agent): bit for bit
        assert torch.equal(e.unet_forward(x[perm], cond[perm], 41).cpu(), whole.cpu()[perm])
    elif B <= 2100:                                  # every launch in half items (the other Winograd kernel: its fourth tile where this one has a direct column)
        assert not torch.equal(whole, wino)
    assert float((whole - wino).abs().max()) <= 1e-5
    assert float((wino - direct).abs().max()) <= 1e-5
    rows = torch.tensor([0, 3, 15, 16, B // 2 + 1, B - 2, B - 1])      # first / last workgroups, both sides of a 16-agent boundary
    ref = O.unet_forward(O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), x[rows], cond[rows],
                         torch.full((len(rows),), 41, dtype=torch.long))
    assert float((wino.cpu()[rows] - ref).abs().max()) <= 2e-5
    # a row's result does not depend on the batch around it: the same rows as their own small batch, bit for bit
    e.force_kernel("conv5", "winograd")
    small = e.unet_forward(x[rows], cond[rows], 41)
    e.force_kernel("conv5", "direct")
    small_d = e.unet_forward(x[rows], cond[rows], 41)
    assert float((small.cpu() - wino.cpu()[rows]).abs().max()) <= 1e-5
    assert float((small_d.cpu() - direct.cpu()[rows]).abs().max()) <= 1e-5
