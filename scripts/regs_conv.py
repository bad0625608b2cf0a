"""Register / scratch / LDS use of every kernel in one .hip file of the library (compiler view):
    python3 scripts/regs_conv.py [csrc/conv_block.hip] [substring filter]
Runs hipcc -Rpass-analysis=kernel-resource-usage and prints one line per kernel; kernels with scratch (spills) first."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.environ.get("REGS_PKG") or os.path.join(ROOT, "controllable-latent-diffusion-for-traffic-simulation_amd")
src = sys.argv[1] if len(sys.argv) > 1 else "csrc/conv_block.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(PKG, src), "-o", "/tmp/_regs.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
rows = []
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = b.split("\n")[0].strip()
    g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
    rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
dem = subprocess.run(["c++filt"], input="\n".join(x[0] for x in rows), capture_output=True, text=True).stdout.split("\n")
out = []
for x, d in zip(rows, dem):
    d = d.replace("void cld::", "").replace("(cld::ConvArgs)", "").replace("(cld::ConvPairArgs)", "")
    if flt in d:
        out.append((x[3] <= 0, d, x))
for _, d, x in sorted(out):
    print(f"{d[:90]:90s} VGPR {x[1]:3d} AGPR {x[2]:3d} scratch {x[3]:4d} occ {x[4]} SGPR {x[6]}")
print(len(rows), "kernels,", sum(1 for x in rows if x[3] > 0), "with scratch")
