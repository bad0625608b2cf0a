"""Diagnostic: in-kernel cycle stamps of the Winograd k5 launches (wino1d_kernels.hip; needs the -DCLD_STAMPS build via CLD_LIB_PATH).
    python3 scripts/wino1d_stamps.py <agents> <launch index> ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
layers = [int(a) for a in sys.argv[2:]] or list(range(2, 16))
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
x = torch.randn(B, 52, 4, device=dev); c = torch.randn(B, 256, device=dev)
buf = torch.zeros(16 * 8192, dtype=torch.int64, device=dev)
for _ in range(20): e.unet_forward(x, c, 50)      # clocks settle
for layer in layers:
    for rep in range(3):
        buf.zero_()
        e._check(e.lib.cld_debug_stamps(e._h, C.c_void_p(buf.data_ptr()), layer), "stamps")
        e.unet_forward(x, c, 50); torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
    s = s[(s[:, 0] != 0) & (s[:, 4] != 0)]
    if len(s) == 0:
        print(f"launch {layer}: no stamps (not a Winograd launch)"); continue
    t0 = s[:, 0].min()
    names = ["entry -> first image", "16 chunks (first pair: see below)", "output transform + GroupNorm sums", "affine + Mish + stores"]
    d = np.diff(s[:, :5], axis=1)
    clk = (s[:, 4] - s[:, 0]) / np.maximum((s[:, 9] - s[:, 8]), 1) * 100.0   # MHz
    print(f"launch {layer}: {len(s)} workgroups; kernel span {s[:,4].max()-t0} cyc; clock ~{np.median(clk):.0f} MHz; workgroup life mean {(s[:,4]-s[:,0]).mean():.0f}")
    for k, nme in enumerate(names):
        print(f"   {nme:36s} mean {d[:,k].mean():9.0f}  min {d[:,k].min():8d}  max {d[:,k].max():8d}")
    print(f"   first chunk pair {np.mean(s[:,5]-s[:,1]):.0f} cyc")
    st = np.sort(s[:, 0] - t0)
    print("   workgroup start times (cycles), deciles:", [int(st[int(q * (len(st) - 1))]) for q in np.linspace(0, 1, 11)])
    if len(sys.argv) > 2:            # placement: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
        full = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
        n = int((full[:, 0] != 0).sum())
        hw, xcc = full[:n, 10], full[:n, 11] & 15
        cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
        print("   first 24 ids -> (xcc, se, sh, cu, wave slot):", [(int(xcc[i]), int((hw[i] >> 13) & 7), int((hw[i] >> 12) & 1), int((hw[i] >> 8) & 15), int(hw[i] & 15)) for i in range(24)])
        first = {}
        for i in range(min(n, 512)):
            first.setdefault(int(cu[i]), []).append(i)
        pairs = [v for v in first.values()]
        print("   distinct CUs among the first 512 ids:", len(first), " ids sharing a CU (first 12 CUs):", pairs[:12])
        print("   |id difference| of the first two ids on a CU: ", np.bincount([abs(v[1] - v[0]) for v in pairs if len(v) > 1]).nonzero()[0][:20])
    en = np.sort(s[:, 4] - t0)
    print("   workgroup end times (cycles), deciles:  ", [int(en[int(q * (len(en) - 1))]) for q in np.linspace(0, 1, 11)])
