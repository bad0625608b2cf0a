"""Phase shares of the guidance kernel from in-kernel stamps (needs the -DCLD_STAMPS build via CLD_LIB_PATH):
    CLD_LIB_PATH=.../libcld_hip_stamps.so python3 scripts/guide_stamps.py 2048 [form]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FORM = int(sys.argv[2]) if len(sys.argv) > 2 else 3      # 3 = 8 agents per workgroup (4x4x1 MFMA), 2 = 16 agents (16x16x4 MFMA)
e = Engine(10, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
e._check(e.lib.cld_debug_force_kernel(e._h, 0, FORM), "force")
g = torch.Generator(device="cuda"); g.manual_seed(B)
mean = torch.randn(B, 52, 4, device="cuda", generator=g); cond = torch.randn(B, 256, device="cuda", generator=g)
cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = torch.rand(B, device="cuda", generator=g) * 15
z = torch.randn(B, 52, 4, device="cuda", generator=g)
gd = {"curr_states": cs, "target_speed": torch.rand(B, 52, device="cuda", generator=g) * 12, "lr": 0.3, "optimizer": "adam"}
for _ in range(20):
    e.guidance_step(mean, cond, gd, 0.5, z=z)
torch.cuda.synchronize()
out = np.zeros(2048, np.uint64)
e.lib.cld_debug_guide_stamps(out.ctypes.data)
s = out.reshape(256, 8).astype(np.int64)[: min(256, (B + 7) // 8 if FORM == 3 else (B + 15) // 16)]
d = np.diff(s[:, :7], axis=1)
names = ["load cond / mean + cond2hidden", "forward weights", "forward 52 steps", "actions + roll-out scan (chain_grad)", "backward weights + first fetch", "backward 52 steps"]
tot = (s[:, 6] - s[:, 0]).mean()
for k, nme in enumerate(names):
    print(f"{nme:38s} mean {d[:, k].mean():10.0f} cyc  {100 * d[:, k].mean() / tot:5.1f} %   per step {d[:, k].mean() / 52:8.0f}")
print(f"total {tot:.0f} cycles over {len(s)} workgroups")
