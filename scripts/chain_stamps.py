"""Diagnostic: in-kernel cycle stamps of the layer-chain launches (needs the -DCLD_STAMPS build via CLD_LIB_PATH).
    CLD_LIB_PATH=.../libcld_stamps.so python3 scripts/chain_stamps.py 4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
e.force_kernel("unet", "chain")
x = torch.randn(B, 52, 4, device=dev); c = torch.randn(B, 256, device=dev)
buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
for rep in range(3):
    buf.zero_()
    e._check(e.lib.cld_debug_stamps(e._h, C.c_void_p(buf.data_ptr()), 0), "stamps")
    e.unet_forward(x, c, 50); torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
names = ["staging + latent conv", "k5 loop 1", "epilogue 1", "k5 loop 2", "epilogue 2", "k5 loop 3", "epilogue 3", "k3s2 loop", "stores"]
d = np.diff(s[:, :10], axis=1)
clk = (s[:, 9] - s[:, 0]) / np.maximum((s[:, 15] - s[:, 14]), 1) * 100.0   # MHz
print(f"head chain: {len(s)} workgroups; entry skew max {s[:,0].max()-t0} cyc; kernel span {s[:,9].max()-t0} cyc; workgroup life mean {(s[:,9]-s[:,0]).mean():.0f}; clock ~{np.median(clk):.0f} MHz")
for k, nme in enumerate(names):
    print(f"   {nme:22s} mean {d[:,k].mean():9.0f}  min {d[:,k].min():8d}  max {d[:,k].max():8d}")
