"""Layer chains (conv_chain.hip) against one launch per layer (conv_block.hip): max difference of a U-Net evaluation and the
time per evaluation in both forms.
    python3 scripts/chain_check.py 1024 2048 4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine

sizes = [int(a) for a in sys.argv[1:]] or [64, 1024, 2048, 4096]
e = Engine(100, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0, affine_jitter=True)); e.finalize()
for B in sizes:
    g = torch.Generator(device="cuda"); g.manual_seed(B)
    x = torch.randn(B, 52, 4, device="cuda", generator=g) * 2.0
    c = torch.randn(B, 256, device="cuda", generator=g)
    out, ms = {}, {}
    for form in ("layers", "chain", "chain1", "chain4", "chainw", "chainw2", "chainw1"):
        e.force_kernel("unet", form)
        out[form] = e.unet_forward(x, c, 37).clone()
        for _ in range(3):
            e.unet_forward(x, c, 37)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            e.unet_forward(x, c, 37)
        ev1.record(); torch.cuda.synchronize()
        ms[form] = ev0.elapsed_time(ev1) / 20
    # one CFG sampling step (2B rows per launch set; the head combines the two halves of the noise prediction)
    nc = torch.randn(B, 256, device="cuda", generator=g)
    z = torch.randn(B, 52, 4, device="cuda", generator=g)
    stp = {}
    for form in ("layers", "chain"):
        e.force_kernel("unet", form)
        stp[form] = e.sample_step(x, c, 37, z=z, non_cond=nc, guidance_w=2.0)["x_next"].clone()
    ds = (stp["chain"] - stp["layers"]).abs().max().item()
    print(f"B={B}: CFG step max|chain - layers| = {ds:.3e} (max|x'| = {stp['layers'].abs().max().item():.3e})")
    d = (out["chain"] - out["layers"]).abs().max().item()
    print(f"B={B}: max|chain - layers| = {d:.3e} (max|eps| = {out['layers'].abs().max().item():.3e}, finite={torch.isfinite(out['chain']).all().item()});"
          f" per evaluation incl. pack / cond-bias / head: layers {ms['layers']*1e3:.1f} us, chain {ms['chain']*1e3:.1f} us"
          f" (one-agent tiles {ms['chain1']*1e3:.1f}, four-agent tiles {ms['chain4']*1e3:.1f}, Winograd tiles of four / two / one agents {ms['chainw']*1e3:.1f} / {ms['chainw2']*1e3:.1f} / {ms['chainw1']*1e3:.1f};"
          f" max|chainw - chain4| = {(out['chainw'] - out['chain4']).abs().max().item():.3e})", flush=True)
