"""CPU tests of the host logic: the C-ABI library loads and exports every symbol of include/cld.h,
the synthetic generator is deterministic, config plumbing, scene sharding, and the world_size-2
gloo path of the trajectory all-gather.  No GPU compute is called here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from cld_amd import _lib
    lib_path = _lib.LIB_PATH
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    hdr = open(os.path.join(ROOT, "include", "cld.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cld_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/cld.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with include/cld.h"
    lib.cld_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cld_version()


def test_default_config_matches_reference_yaml():
    from cld_amd import _lib
    lib = _lib.load()
    cfg = _lib.CldConfig()
    lib.cld_default_config(ctypes.byref(cfg))
    assert (cfg.horizon, cfg.latent_dim, cfg.cond_dim, cfg.base_dim, cfg.hidden, cfg.n_timesteps) == (52, 4, 256, 32, 64, 100)
    assert list(cfg.dim_mults) == [2, 4, 8]
    assert list(cfg.acce_bound) == [-10.0, 8.0] and list(cfg.v_bound) == [-10.0, 30.0]
    assert abs(cfg.max_yawvel - 2 * np.pi) < 1e-6 and cfg.max_steer == 0.5
    from oracle import cld_oracle as O
    assert np.allclose(list(cfg.norm_mean), O.NORM_MEAN) and np.allclose(list(cfg.norm_std), O.NORM_STD)


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cld_amd._lib import CldError
    from cld_amd.engine import Engine
    with pytest.raises(CldError):
        Engine(device="cuda:0")
    with pytest.raises(CldError):
        Engine(device="cpu")


def test_synth_is_deterministic_and_shaped_like_the_reference_state_dict():
    from cld_amd import synth
    a, b = synth.make_unet_weights(0), synth.make_unet_weights(0)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert sum(v.size for v in a.values()) == 4_349_284          # SURVEY section 6 [measured]
    assert len(a) == 148
    assert sum(v.size for v in synth.make_decoder_weights(0).values()) == 67_778
    c = synth.make_unet_weights(1)
    assert not np.array_equal(a["model.mid_block1.blocks.0.block.0.weight"], c["model.mid_block1.blocks.0.block.0.weight"])
    z = synth.normal(5, "z", (200000,))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    n8, n16 = synth.make_noise(8, 3, 1), synth.make_noise(8, 3, 1)
    assert np.array_equal(n8["noise"], n16["noise"])


def test_scene_sharding_partitions_exactly():
    from cld_amd.parallel import shard_agents, shard_scenes
    for n, w in ((1024, 8), (512, 8), (10, 4), (3, 8), (32, 1)):
        spans = [shard_scenes(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert shard_agents(1024, 64, 8, 3) == (3 * 128 * 64, 4 * 128 * 64)


def test_cfg_get_reads_dict_and_attr_configs():
    from cld_amd.dm_model import cfg_get, repeat_by_expand_at

    class A:
        pass
    a = A(); a.vae = A(); a.vae.latent_size = 4
    assert cfg_get(a, "vae.latent_size") == 4 and cfg_get({"vae": {"latent_size": 4}}, "vae.latent_size") == 4
    assert cfg_get(a, "missing.key", 7) == 7
    t = torch.arange(6).reshape(3, 2)
    r = repeat_by_expand_at({"x": t, "k": 3}, 2, 0)
    assert r["x"].tolist() == [[0, 1], [0, 1], [2, 3], [2, 3], [4, 5], [4, 5]] and r["k"] == 3


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CLD_ROOT"])
from cld_amd.parallel import gather_trajectories, gather_ragged, shard_scenes
dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
r, w = dist.get_rank(), dist.get_world_size()
B = 6
local = torch.full((B, 52, 6), float(r)) + torch.arange(B).reshape(B, 1, 1)
full = gather_trajectories(local)
assert full.shape == (w * B, 52, 6)
for q in range(w):
    assert torch.equal(full[q * B:(q + 1) * B], torch.full((B, 52, 6), float(q)) + torch.arange(B).reshape(B, 1, 1))
spans = [shard_scenes(5, w, q) for q in range(w)]            # uneven split: 3 + 2 scenes (2 + 2 + 1 over three ranks)
sizes = [(h - l) * 4 for l, h in spans]
lo, hi = spans[r]
mine = torch.arange(lo * 4, hi * 4, dtype=torch.float32).reshape(-1, 1, 1).expand(-1, 52, 6).contiguous()
allr = gather_ragged(mine, sizes)
assert allr.shape == (20, 52, 6) and torch.equal(allr[:, 0, 0], torch.arange(20, dtype=torch.float32))
# BASELINE configs[3]: the fixed job of 1,024 scenes sharded by scene over this world (one row per scene here): every rank
# ends up with every scene's row in scene order, whether the split is even or not (bench.py --gpus N for N in 3, 5, 6, 7)
spans = [shard_scenes(1024, w, q) for q in range(w)]
sizes = [h - l for l, h in spans]
lo, hi = spans[r]
mine = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).expand(-1, 52, 6).contiguous()
allr = gather_ragged(mine, sizes) if len(set(sizes)) > 1 else gather_trajectories(mine)
assert allr.shape == (1024, 52, 6) and torch.equal(allr[:, 51, 5], torch.arange(1024, dtype=torch.float32))
dist.barrier(); dist.destroy_process_group()
print("rank", r, "ok")
'''


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_gather_even_and_uneven_splits(tmp_path, world):
    """The N > 1 exchange on the CPU (gloo): equal shards, a 3 + 2 scene split, and configs[3]'s 1,024 scenes over 2 ranks
    (512 + 512, one all_gather_into_tensor) and over 3 ranks (342 + 341 + 341, the padded gather)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, CLD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29611 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_scene_shards_cover_the_job_for_every_world_size():
    """parallel.shard_scenes / shard_agents for BASELINE configs[3] (1,024 x 64) and configs[4] (512 x 64) at every rank count
    the driver may use: contiguous, disjoint, complete, and at most one scene apart in size."""
    from cld_amd.parallel import shard_agents, shard_scenes
    for scenes in (1024, 512, 5):
        for world in range(1, 9):
            spans = [shard_scenes(scenes, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == scenes
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            assert [shard_agents(scenes, 64, world, r) for r in range(world)] == [(l * 64, h * 64) for l, h in spans]


def test_context_host_mirror_and_synthetic_inputs():
    """Host logic of the ContextEncoder row: curr_states assembly (batch_utils.py:46-65), state_dict key set, raster shape."""
    from cld_amd import synth
    from cld_amd.context_utils import get_current_states
    B = 5
    batch = {"history_positions": torch.randn(B, 31, 2), "history_yaws": torch.randn(B, 31, 1), "curr_speed": torch.rand(B)}
    cs = get_current_states(batch)
    assert cs.shape == (B, 4)
    assert torch.equal(cs[:, :2], batch["history_positions"][:, -1]) and torch.equal(cs[:, 2], batch["curr_speed"])
    assert torch.equal(cs[:, 3], batch["history_yaws"][:, -1, 0])
    w = synth.make_context_weights(0)
    assert len(w) == 130 and "context_encoder.map_encoder.encoder_heads.map_model.conv1.weight" in w
    assert w["context_encoder.map_encoder.encoder_heads.map_model.conv1.weight"].shape == (64, 34, 7, 7)
    assert w["context_encoder.process_cond_mlp._model.12.weight"].shape == (256, 256)
    assert w["context_encoder.agent_state_encoder._model.1.weight"].shape == (64,)         # LayerNorm
    r = synth.make_raster(2, 3)
    assert r.shape == (2, 34, 224, 224) and float((r[:, :31] != 0).mean()) < 1e-3          # near-empty history planes
    assert set(np.unique(r[:, 31:])) <= {0.0, 1.0}


def test_guidance_struct_matches_header():
    """ctypes mirror of `cld_guidance` (include/cld.h): three device pointers, two floats, one int32."""
    from cld_amd import _lib
    g = _lib.CldGuidance
    assert [n for n, _ in g._fields_] == ["curr_states", "target_speed", "loss_scale", "lr", "perturb_th", "optimizer",
                                          "speed_limit", "acc_limit", "speed_limit_scale", "acc_limit_scale",
                                          "target_pos", "target_time", "target_pos_scale", "ext_grad",
                                          "apply_output", "no_intermediate", "final_lr", "final_perturb_th", "final_optimizer",
                                          "grad_steps", "final_grad_steps", "guide_clean", "collision", "map_collision"]
    assert ctypes.sizeof(g) == 144 and g.apply_output.offset == 96 and g.final_optimizer.offset == 112 and g.lr.offset == 24 and g.optimizer.offset == 32 and g.speed_limit_scale.offset == 48
    assert g.grad_steps.offset == 116 and g.guide_clean.offset == 124 and g.collision.offset == 128      # appended in round 3: the earlier layout is untouched
    c = _lib.CldCollision
    assert [n for n, _ in c._fields_] == ["extent", "world_from_agent", "curr_speed", "scene_start", "scene_weight", "guided", "num_scenes",
                                          "num_samp", "num_disks", "max_scene_agents", "buffer_dist", "decay_rate", "moving_speed_th", "excluded"]
    assert ctypes.sizeof(c) == 88 and c.num_scenes.offset == 48 and c.buffer_dist.offset == 64 and c.excluded.offset == 80      # `excluded` appended in round 4
    cbody = open(os.path.join(ROOT, "include", "cld.h")).read()
    cbody = cbody[cbody.index("typedef struct cld_collision {"):cbody.index("} cld_collision;")]
    assert re.findall(r"^\s+(?:const\s+)?\w+\*?\s+\*?(\w+);", cbody, re.M) == [n for n, _ in c._fields_]
    m = _lib.CldMapCollision
    assert ctypes.sizeof(m) == 80 and m.num_scenes.offset == 48 and m.decay_rate.offset == 72
    hdr = open(os.path.join(ROOT, "include", "cld.h")).read()
    body = hdr[hdr.index("typedef struct cld_guidance {"):hdr.index("} cld_guidance;")]
    assert [m for m in re.findall(r"\b(curr_states|target_speed|loss_scale|lr|perturb_th|optimizer|speed_limit|acc_limit|speed_limit_scale|acc_limit_scale|target_pos|target_time|target_pos_scale|ext_grad|apply_output|no_intermediate|final_lr|final_perturb_th|final_optimizer|grad_steps|final_grad_steps|guide_clean|collision|map_collision);", body)] == [n for n, _ in g._fields_]


def test_force_kernel_tables_match_header():
    """`Engine.force_kernel(which, form)` goes through the name tables of `_lib`; the numbers are include/cld.h's CLD_KERNEL_* / CLD_FORM_*
    (the tests force every formulation of a kernel through them: a drifted number would silently test another form)."""
    import re
    from cld_amd import _lib
    text = open(os.path.join(ROOT, "include", "cld.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(CLD_(?:KERNEL|FORM)_[A-Z0-9_]+)\s+(\d+)", text)}
    kernels = {"guide": "GUIDE", "decode": "DECODE", "encode": "ENCODE", "unet": "UNET", "context": "CONTEXT", "conv5": "CONV5"}
    assert set(_lib.KERNELS) == set(kernels)
    for name, macro in kernels.items():
        assert _lib.KERNELS[name] == defs["CLD_KERNEL_" + macro], name
    forms = {"auto": "AUTO", "valu": "VALU", "mfma": "MFMA", "quad": "MFMA_QUAD", "layers": "LAYERS", "chain": "CHAIN", "chain1": "CHAIN_TILE1",
             "chain4": "CHAIN_TILE4", "chainw": "CHAIN_WINO", "chainw2": "CHAIN_WINO2", "chainw1": "CHAIN_WINO1", "direct": "DIRECT", "winograd": "WINOGRAD", "winograd_whole": "WINOGRAD_WHOLE", "winograd_ksplit": "WINOGRAD_KSPLIT", "winograd_f2": "WINOGRAD_F2"}
    assert set(_lib.FORMS) == set(forms)
    for name, macro in forms.items():
        assert _lib.FORMS[name] == defs["CLD_FORM_" + macro], name


def test_conv5_form_rule_and_its_32_bit_fallback():
    """Which form a k5 launch takes is a pure rule of the library (cld_debug_conv5_form; csrc/cld_api.hip conv5_takes_winograd): Winograd
    F(4, 5) from 384 rows per launch set for the layer shapes that have an instance, the direct form below, for other shapes, when forced
    -- and whenever the launch's widest tensor reaches 2 GiB, because the Winograd kernels address their tensors with 32-bit byte
    offsets (the direct kernels use 64-bit row bases): forcing Winograd does not override that."""
    from cld_amd import _lib
    f = _lib.load().cld_debug_conv5_form
    AUTO, DIRECT, WINO = _lib.FORMS["auto"], _lib.FORMS["direct"], _lib.FORMS["winograd"]
    shapes = [(13, 256, 0, 256), (13, 128, 0, 128), (13, 128, 0, 256), (13, 256, 256, 128), (26, 128, 0, 128), (26, 64, 0, 128), (26, 128, 128, 64)]
    for (l, c1, c2, co) in shapes:
        assert f(l, c1, c2, co, 383 // 16 * 16, AUTO) == DIRECT and f(l, c1, c2, co, 384, AUTO) == WINO      # the size threshold
        assert f(l, c1, c2, co, 369, AUTO) == WINO                                                           # 369 rows pad to 384
        assert f(l, c1, c2, co, 16, WINO) == WINO and f(l, c1, c2, co, 1 << 16, DIRECT) == DIRECT           # forcing
        widest = max(c1, co)
        edge = (1 << 31) // (l * widest * 4)                 # rows at which the widest tensor reaches 2 GiB
        below = edge // 16 * 16
        if below * l * widest * 4 >= 1 << 31:
            below -= 16
        assert f(l, c1, c2, co, below, AUTO) == WINO and f(l, c1, c2, co, below, WINO) == WINO
        assert f(l, c1, c2, co, below + 16, AUTO) == DIRECT and f(l, c1, c2, co, below + 16, WINO) == DIRECT, (l, c1, c2, co)
    for (l, c1, c2, co) in [(52, 64, 0, 64), (26, 64, 0, 64), (13, 256, 0, 128), (52, 4, 0, 64), (26, 128, 0, 256)]:      # no launch of its own in Winograd form
        assert f(l, c1, c2, co, 4096, AUTO) == DIRECT and f(l, c1, c2, co, 4096, WINO) == DIRECT
    WHOLE = _lib.FORMS["winograd_whole"]                 # whole items at every size: a Winograd form, the same rule
    assert f(13, 256, 0, 256, 16, WHOLE) == WINO and f(52, 64, 0, 64, 4096, WHOLE) == DIRECT
    assert f(13, 256, 0, 256, 16, _lib.FORMS["winograd_ksplit"]) == WINO
    assert f(13, 256, 0, 256, -1, AUTO) < 0 and f(13, 256, 0, 256, 64, 5) < 0


def test_timers_keep_the_reference_surface():
    """cld_amd.timer.Timers: tic / toc / timed / __str__ as src/tbsim/utils/timer.py:41-64 (host clock path, no GPU)."""
    import time as _t
    from cld_amd.timer import Timers
    t = Timers()
    for _ in range(3):
        with t.timed("network"):
            _t.sleep(0.002)
    t.tic("obs"); t.toc("obs")
    s = str(t)
    assert s.startswith("network: ") and ", obs: " in s
    assert t._timers["network"].calls == 3 and t._timers["network"].average_time >= 0.002
    assert t.gpu_ms("network") == 0.0 and t.gpu_ms("missing") == 0.0


def test_guidance_config_adapter_matches_the_golden_configurations():
    """policy.guidance_from_config folds upstream's per-scene guidance lists (name / weight / params / agents) into the per-agent
    scales of the kernel exactly as the golden fixtures were recorded from the reference's DiffuserGuidance (make_golden)."""
    from cld_amd.policy import guidance_from_config
    B, T = 8, 52
    tgt = np.random.RandomState(0).rand(B, T).astype(np.float32)
    scene_index = torch.tensor([4, 4, 4, 9, 9, 9, 9, 9])
    wp = np.arange(6, dtype=np.float32).reshape(3, 2)
    cfg = [[{"name": "target_speed", "weight": 1.0, "params": {"target_speed": tgt}, "agents": None},
            {"name": "speed_limit", "weight": 0.5, "params": {"speed_limit": 6.0}, "agents": None},
            {"name": "acc_limit", "weight": 4.0, "params": {"acc_limit": 0.1}, "agents": None},
            {"name": "target_pos_at_time", "weight": 2.0, "params": {"target_pos": wp, "target_time": [10, 51, 30]}, "agents": None}],
           [{"name": "speed_limit", "weight": 3.0, "params": {"speed_limit": 6.0}, "agents": None},
            {"name": "target_pos", "weight": 0.5, "params": {"target_pos": wp[:2], "min_target_time": 0.5}, "agents": [1, 3]}]]
    g = guidance_from_config(cfg, scene_index)
    assert torch.allclose(g["loss_scale"], torch.tensor([1.0 / (3 * 52)] * 3 + [0.0] * 5))
    assert torch.equal(g["target_speed"][:3], torch.from_numpy(tgt[:3])) and float(g["target_speed"][3:].abs().max()) == 0.0
    assert g["speed_limit"][0] == 6.0 and torch.allclose(g["speed_limit"][1], torch.tensor([0.5 / (3 * 52)] * 3 + [3.0 / (5 * 52)] * 5))
    assert g["acc_limit"][0] == 0.1 and torch.allclose(g["acc_limit"][1], torch.tensor([4.0 / (3 * 52)] * 3 + [0.0] * 5))
    pos, tt, sc = g["target_pos"]
    assert tt.tolist() == [10, 51, 30, 0, -27, 0, -27, 0]                    # scene 1: agents 1 and 3 of the scene, m = 26
    assert torch.allclose(sc, torch.tensor([2.0 / 3] * 3 + [0.0, 0.25, 0.0, 0.25, 0.0]))
    assert torch.equal(pos[4], torch.tensor([0.0, 1.0])) and torch.equal(pos[6], torch.tensor([2.0, 3.0]))
    with pytest.raises(ValueError):                                          # the loss reads observation fields: they must be handed over
        guidance_from_config([[{"name": "agent_collision", "weight": 1.0, "params": {}, "agents": None}], []], scene_index)
    with pytest.raises(ValueError):
        guidance_from_config([[{"name": "map_collision", "weight": 1.0, "params": {}, "agents": None}], []], scene_index)
    with pytest.raises(NotImplementedError):
        guidance_from_config([[{"name": "social_group", "weight": 1.0, "params": {}, "agents": None}], []], scene_index)
    db = {"extent": torch.ones(B, 3), "world_from_agent": torch.eye(3).expand(B, 3, 3), "curr_speed": torch.ones(B)}
    g = guidance_from_config([[{"name": "agent_collision", "weight": 3.0, "params": {"num_disks": 4}, "agents": [0, 2]}],
                              [{"name": "speed_limit", "weight": 3.0, "params": {"speed_limit": 6.0}, "agents": None}]], scene_index, data_batch=db)
    c = g["agent_collision"]
    assert c["weight"] == [3.0, 0.0] and c["agents"] == {0: [0, 2]} and c["num_disks"] == 4 and c["buffer_dist"] == 0.2 and "speed_limit" in g
    with pytest.raises(ValueError):
        guidance_from_config([[]], scene_index)


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table maps every symbol include/cld.h declares to the reference interface it replaces."""
    hdr = open(os.path.join(ROOT, "include", "cld.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in set(re.findall(r"\b(cld_[a-z_0-9]+)\s*\(", hdr)) if s not in doc]
    assert not missing, missing


def test_bench_cli_parses_without_a_gpu():
    """bench.py must import and parse its flags on a CPU-only host (the driver builds here before benching on the GPU box)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for flag in ("--gpus", "--steps", "--warmup", "--cfg-w", "--guide", "--closed-loop", "--precision", "--no-context"):
        assert flag in out.stdout


def test_minimal_filtering_matrices_of_the_kernels_reproduce_the_direct_form():
    """The transform matrices the Winograd kernels are written from (csrc/wino1d_kernels.hip / wino1d_edge.hip: F(4, 5) at {0, +-1, +-2, +-1/2, inf};
    csrc/wino44_kernels.hip: F(4x4, 3x3) at {0, 1, -1, 1/2, -2, inf}; G as csrc/cld_api.hip forms U), checked in fp64 against the direct
    correlation -- including wino1d_edge.hip's split of a 13-long row into three tiles and one direct output (27 products per channel pair)."""
    rng = np.random.default_rng(0)
    # ---- F(4, 5): y[4 t .. 4 t + 3] = A^T [(G g) (.) (B^T d)], d = x[4 t - 2 .. 4 t + 5]
    BT = np.array([[-1, 0, 21 / 4, 0, -21 / 4, 0, 1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                   [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0], [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0], [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0],
                   [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]])
    G = np.array([[-1, 0, 0, 0, 0], [-2 / 9] * 5, [-2 / 9, 2 / 9, -2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45, 4 / 45, 8 / 45],
                  [1 / 90, -1 / 45, 2 / 45, -4 / 45, 8 / 45], [32 / 45, 16 / 45, 8 / 45, 4 / 45, 2 / 45], [32 / 45, -16 / 45, 8 / 45, -4 / 45, 2 / 45], [0, 0, 0, 0, 1]])
    AT = np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0], [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 1]])
    g = rng.standard_normal(5)
    x = rng.standard_normal(13)
    xp = np.concatenate([np.zeros(2), x, np.zeros(6)])                 # pad 2 | 13 values | zeros past the end
    ref = np.array([sum(g[k] * xp[l + k] for k in range(5)) for l in range(13)])
    y = np.empty(13)
    for t in range(3):                                                  # outputs 0 .. 11: three tiles, 8 products each
        y[4 * t:4 * t + 4] = AT @ ((G @ g) * (BT @ xp[4 * t:4 * t + 8]))
    y[12] = sum(g[k] * x[10 + k] for k in range(3))                     # output 12 in the direct form: taps 3, 4 meet the padding
    assert np.abs(y - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    src = open(os.path.join(ROOT, "controllable-latent-diffusion-for-traffic-simulation_amd", "csrc", "cld_api.hip")).read()
    assert "{32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45}" in src and "{1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45}" in src
    # ---- F(4x4, 3x3): Y = A^T [(G g G^T) (.) (B^T d B)] A, d = the 6x6 patch at rows / columns 4 t - 1 .. 4 t + 4
    BT6 = np.array([[1, -3 / 2, -2, 3 / 2, 1, 0], [0, -1, 1 / 2, 5 / 2, 1, 0], [0, 1, -5 / 2, 1 / 2, 1, 0], [0, -2, -1, 2, 1, 0], [0, 1 / 2, -1, -1 / 2, 1, 0],
                    [0, 1, -3 / 2, -2, 3 / 2, 1]])
    G6 = np.array([[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15], [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]])
    AT6 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 1 / 2, -2, 0], [0, 1, 1, 1 / 4, 4, 0], [0, 1, -1, 1 / 8, -8, 1]])
    g2 = rng.standard_normal((3, 3))
    d = rng.standard_normal((6, 6))
    Y = AT6 @ ((G6 @ g2 @ G6.T) * (BT6 @ d @ BT6.T)) @ AT6.T
    ref2 = np.array([[sum(g2[a, b] * d[i + a, j + b] for a in range(3) for b in range(3)) for j in range(4)] for i in range(4)])
    assert np.abs(Y - ref2).max() <= 1e-12 * max(1.0, np.abs(ref2).max())
    assert "{-16.0 / 15, -8.0 / 15, -4.0 / 15}, {1.0 / 15, -2.0 / 15, 4.0 / 15}" in src
    for M in (BT, AT, BT6, AT6):                                        # exact in fp32: the kernels apply them with fp32 constants
        assert np.array_equal(M.astype(np.float32).astype(np.float64), M)
