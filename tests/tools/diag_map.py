"""Diagnostic: where the map-collision gradient of the kernel and of the oracle differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
from oracle import cld_oracle as O
grid = tuple(int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (1, 16)
e = Engine(10, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
sizes, N = [40, 7, 30], 2
B = sum(sizes)
sc = synth.make_map_scene(B, 17)
db = {k: torch.from_numpy(v) for k, v in sc.items()}
db["scene_index"] = torch.repeat_interleave(torch.arange(3), torch.tensor(sizes))
traj = torch.from_numpy(synth.make_map_trajectories(B, N, sc["curr_speed"], 17)).reshape(B * N, 52, 6)
wts = [1.5, 0.0, 0.7]
x = traj.clone().requires_grad_(True)
tot = O.scene_map_collision_total(x, dict(db, scene_weight=wts, num_points_lw=grid), N)
(gref,) = torch.autograd.grad(tot, x)
cfg = dict(extent=db["extent"], raster_from_agent=db["raster_from_agent"], drivable_map=db["drivable_map"], curr_speed=db["curr_speed"], weight=wts, num_samp=N, scene_sizes=sizes, num_points_lw=grid)
loss, grad = e.map_collision(traj, cfg)
d = (grad.cpu() - gref).abs()
print("max diff", float(d.max()), "max ref", float(gref.abs().max()))
idx = torch.nonzero(d > 0.02 * gref.abs().max())
print(len(idx), "elements differ by more than 2% of max; first:")
for r, t, c in idx[:12].tolist():
    print(f"  row {r} (agent {r // N}, sample {r % N}) t {t} ch {c}: kernel {float(grad[r, t, c]):.6e} oracle {float(gref[r, t, c]):.6e}")
