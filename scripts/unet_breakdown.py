"""Per-launch breakdown of one U-Net evaluation from a rocprofv3 kernel trace of scripts/one_unet.py:
    python3 scripts/unet_breakdown.py <t_kernel_trace.csv> <agents>
Prints, for every launch of the LAST evaluations in the trace (median over them), the kernel instance, its duration, its
algorithmic FLOP (2 * rows * K * N of the layers it computes, SURVEY 8(a) layer table) and the fp32-MFMA rate it reaches."""
import csv, statistics, sys

# MAC per agent of every launch of run_unet (csrc/cld_api.hip), in launch order; pairs are one launch
L = [("b0.c0 4->64@52 (K folded)", 66560), ("b0.c1 64->64@52 (+res 1x1)", 1064960 + 13312), ("b1.c0", 1064960), ("b1.c1", 1064960),
     ("down0 k3s2", 319488), ("b2.c0+res 64->128@26", 1064960 + 212992), ("b2.c1 128->128@26", 2129920), ("b3.c0", 2129920),
     ("b3.c1", 2129920), ("down1 k3s2", 638976), ("b4.c0+res 128->256@13", 2129920 + 425984), ("b4.c1 256->256@13", 4259840),
     ("b5.c0", 4259840), ("b5.c1", 4259840), ("b6.c0", 4259840), ("b6.c1", 4259840), ("b7.c0", 4259840), ("b7.c1", 4259840),
     ("b8.c0+res 512->128@13", 4259840 + 851968), ("b8.c1 128->128@13", 1064960), ("b9.c0", 1064960), ("b9.c1", 1064960),
     ("up0 convT 128@13->26", 2 * 425984), ("b10.c0+res 256->64@26", 2129920 + 425984), ("b10.c1 64->64@26", 532480),
     ("b11.c0", 532480), ("b11.c1", 532480), ("up1 convT 64@26->52", 2 * 212992), ("final_conv.0 64->64@52", 1064960),
     ("head 1x1 64->4 + update", 13312)]
# with the layer chains (conv_chain.hip, default from 1,024 rows): launches 0-4 and 24-29 are one launch each
LC = [("chain: downs.0 (5 layers 64ch@52 -> 64@26)", sum(m for _, m in L[0:5]))] + L[5:24] + \
     [("chain: ups.1 2nd half + final_conv (6 layers)", sum(m for _, m in L[24:30])), ("head: DDPM update on eps [B,52,4]", 0)]
# with the Winograd form of the k5 layers at L = 13 / 26 (wino1d_kernels.hip, default from 1,024 rows): a block's opening conv and its 1x1
# residual projection are two launches (the pair kernel ran them as one)
def split_pair(tag, mac, res_mac):
    return [(tag.split("+res")[0] + " (k5, Winograd)" + tag.split("+res")[1], mac - res_mac), (tag.split("+res")[0].split(".")[0] + ".res 1x1" + tag.split("+res")[1], res_mac)]
LW = [LC[0]] + split_pair(*LC[1], 212992) + LC[2:6] + split_pair(*LC[6], 425984) + LC[7:14] + split_pair(*LC[14], 851968) + LC[15:19] + \
     split_pair(*LC[19], 425984) + LC[20:]
path, B = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if "cld::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an evaluation = the conv launches between two head_kernel launches (pack / cond-bias launches in between are skipped)
ends = [i for i, n in enumerate(names) if "cld::head_kernel" in n]
evals = []
for a, e in zip(ends[:-1], ends[1:]):
    ev = [r for r in rows[a + 1:e + 1] if any(k in r["Kernel_Name"] for k in ("conv_block_kernel", "conv_pair_kernel", "cld::head_kernel", "chain_", "wino1d_"))]
    evals.append(ev)
evals = [ev for ev in evals if len(ev) == len(evals[-1])][-8:]
assert evals, "no complete U-Net evaluation in the trace"
nl = len(evals[-1])
if any("chain_" in r["Kernel_Name"] for r in evals[-1]):
    L = LW if nl == len(LW) else LC
labelled = nl == len(L)          # below ~2,048 rows some pairs run as two launches (their tilings differ): no per-launch FLOP then
tot_t = 0.0
tot_f = 2.0 * sum(m for _, m in L) * B
print(f"{len(evals)} evaluations of {nl} launches, {B} agents; peak 157.3 TFLOP/s")
for i in range(nl):
    d = statistics.median(int(ev[i]["End_Timestamp"]) - int(ev[i]["Start_Timestamp"]) for ev in evals) / 1e3
    k = evals[-1][i]["Kernel_Name"].replace("void cld::", "").replace("(cld::ConvArgs)", "").replace("(cld::ConvPairArgs)", "").replace("cld::", "")
    if labelled:
        tag, mac = L[i]
        fl = 2.0 * mac * B
        print(f"{i:2d} {tag:46s} {d:8.1f} us {fl/1e9:8.3f} GFLOP {fl/d/1e6:7.1f} TF/s {fl/d/1e6/157.3*100:5.1f}%  ideal {fl/157.3e6:6.1f} us  {k[:70]}")
    else:
        print(f"{i:2d} {d:8.1f} us  {k[:90]}")
    tot_t += d
wall = statistics.median(int(ev[-1]["End_Timestamp"]) - int(ev[0]["Start_Timestamp"]) for ev in evals) / 1e3
print(f"sum of kernels {tot_t:.1f} us, first start -> last end {wall:.1f} us, {tot_f/1e9:.1f} GFLOP, ideal {tot_f/157.3e6:.1f} us -> {tot_f/wall/1e6/157.3*100:.1f}% of peak")
