"""Time of the collision-guided step: the AgentCollisionLoss kernel alone, and one guided denoising step (decode + collision + guidance
kernel) per optimiser step, at BASELINE configs[2]'s scene layout (32 scenes x 64 agents).
    python3 scripts/collision_time.py [scenes] [agents_per_scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cld_amd import synth
from cld_amd.engine import Engine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
A = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = S * A
e = Engine(100, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
inp = synth.make_inputs(B, 1)
cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
sc = synth.make_collision_scene([A] * S, 3, spacing=3.0)
sc["curr_speed"] = inp["curr_states"][:, 2].copy()
col = dict(extent=sc["extent"], world_from_agent=sc["world_from_agent"], curr_speed=sc["curr_speed"], scene_index=sc["scene_index"], weight=50.0)
mean = torch.randn(B, 52, 4, device="cuda") * 0.5
traj = e.decode(mean, cond, cs, descaled_output=True)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


loss, grad = e.agent_collision(traj, col)
print(f"{S} scenes x {A} agents = {B}: colliding pairs' value sum {float(loss.sum()):.3f}, max|grad| {float(grad.abs().max()):.3e}")
print(f"  agent_collision kernel (value + gradient, incl. the Python wrapper): {timed(lambda: e.agent_collision(traj, col)):.1f} us")
ms = synth.make_map_scene(B, 5, half_width_m=(0.8, 2.0))
ms["curr_speed"] = inp["curr_states"][:, 2].copy()
mcol = dict(extent=ms["extent"], raster_from_agent=ms["raster_from_agent"], drivable_map=ms["drivable_map"], curr_speed=ms["curr_speed"],
            scene_sizes=[A] * S, weight=1.0)
# everything resident on the device before the timed calls (the wrapper would otherwise upload the 100-MB map on every call)
mcol = {k: (torch.as_tensor(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in mcol.items()}
mcol["drivable_map"] = mcol["drivable_map"].to(torch.uint8)
col = {k: (torch.as_tensor(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in col.items()}
mloss, mgrad = e.map_collision(traj, mcol)
print(f"  map_collision kernel (value + gradient, 10 x 10 samples per box, incl. the Python wrapper): {timed(lambda: e.map_collision(traj, mcol)):.1f} us"
      f"  (plans with a partial overlap: {int((mloss > 0).sum())} of {B})")
tgt = torch.rand(B, 52, device="cuda") * 12
for steps in (1, 3):
    g0 = dict(curr_states=cs, target_speed=tgt, lr=0.3, optimizer="adam", grad_steps=steps)
    g1 = dict(g0, agent_collision=col)
    g2 = dict(g1, map_collision=mcol)
    t0 = timed(lambda: e.guidance_step(mean, cond, g0, sigma=0.5))
    t1 = timed(lambda: e.guidance_step(mean, cond, g1, sigma=0.5))
    t2 = timed(lambda: e.guidance_step(mean, cond, g2, sigma=0.5))
    print(f"  guided step, grad_steps = {steps}: target speed alone {t0:.1f} us; + agent_collision {t1:.1f} us; + map_collision {t2:.1f} us "
          f"(decode + loss kernels + guidance kernel per optimiser step)")
