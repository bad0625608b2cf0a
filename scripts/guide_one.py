"""Run n guidance steps of ONE kernel formulation (for rocprofv3 counter passes):  python3 scripts/guide_one.py B form n"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
B, form, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
e = Engine(10, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
e._check(e.lib.cld_debug_force_kernel(e._h, 0, form), "force")
g = torch.Generator(device="cuda"); g.manual_seed(B)
mean = torch.randn(B, 52, 4, device="cuda", generator=g); cond = torch.randn(B, 256, device="cuda", generator=g)
cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = torch.rand(B, device="cuda", generator=g) * 15
z = torch.randn(B, 52, 4, device="cuda", generator=g)
gd = {"curr_states": cs, "target_speed": torch.rand(B, 52, device="cuda", generator=g) * 12, "lr": 0.3, "optimizer": "adam"}
for _ in range(n):
    e.guidance_step(mean, cond, gd, 0.5, z=z)
torch.cuda.synchronize()
