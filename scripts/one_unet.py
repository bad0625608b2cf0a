"""A few U-Net evaluations at a given batch size, for a per-launch kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 scripts/one_unet.py 4096 [precision]
then  python3 scripts/unet_breakdown.py OUT/**/t_kernel_trace.csv 4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
e = Engine(100, "cuda:0", precision=prec); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
g = torch.Generator(device="cuda"); g.manual_seed(B)
x = torch.randn(B, 52, 4, device="cuda", generator=g); c = torch.randn(B, 256, device="cuda", generator=g)
for _ in range(12):
    e.unet_forward(x, c, 50)
torch.cuda.synchronize()
print("done", B, prec)
