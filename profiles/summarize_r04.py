"""Condense gpurun_out/r04/ (written by profiles/run_r04.sh on the GPU box) into the committed summaries under profiles/r04/.
Counters are per-launch averages per kernel; FETCH_SIZE / WRITE_SIZE are KiB in rocprofv3's output and FETCH_SIZE is doubled
per MI355X_MICROARCH.md (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes)."""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles", "r04")
os.makedirs(DST, exist_ok=True)

for f in ("bench_n1.json", "bench_n1_configs1.json", "bench_n1_configs3.json", "bench_n1_configs4_shard.json", "bench_n2_gloo_rehearsal.json",
          "batch_sweep.txt", "batch_sweep_direct_form.txt", "ctx_time_1024.txt", "wino1d_stamps_4096.txt", "wino44_stamps.txt", "ctx_clock_mfma_busy.txt",
          "guide_time.txt", "unet_B64.txt", "unet_B1024.txt", "unet_B2048.txt", "unet_B4096.txt", "chain_check.txt", "collision_time.txt",
          "chainw_stamps_4096.txt", "mfma_covalu.txt"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
shutil.copy(os.path.join(SRC, "kstats", "bench_kernel_stats.csv"), os.path.join(DST, "kernel_stats_bench_configs2_steps1_warmup1.csv"))
shutil.copy(os.path.join(SRC, "kstats_ctx", "ctx_kernel_stats.csv"), os.path.join(DST, "kernel_stats_context_encoder_B256.csv"))


def per_launch(dirs):
    acc = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        p = os.path.join(SRC, d, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if "cld::" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


rows = per_launch(["pmc_sq", "pmc_fetch", "pmc_write", "pmc_ctx_fetch", "pmc_ctx_write"])
with open(os.path.join(DST, "pmc_per_launch_avg.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "avg_per_launch"])
    for (k, c), (s, n) in sorted(rows.items()):
        w.writerow([k, c, n, round(s / n, 1)])


def avg(kernel_sub, counter):
    v = [(s / n) for (k, c), (s, n) in rows.items() if kernel_sub in k and c == counter]
    return v[0] if v else None


out = {}
dom = "wino1d_edge_kernel<13, 256, 256, 256, 1>"     # the 256 -> 256 k5 launches in their Winograd form: the headline launches 4,096 rows (2 x 2,048 agents, CFG)
fe, wr = avg(dom, "FETCH_SIZE"), avg(dom, "WRITE_SIZE")
if fe is not None and wr is not None:
    out = {"kernel": "void cld::" + dom + "(cld::ConvArgs, int, int)", "batch_agents": 4096,
           "hbm_bytes_per_launch": int(2 * fe * 1024 + wr * 1024), "fetch_bytes_x2_corrected": int(2 * fe * 1024), "write_bytes": int(wr * 1024),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/run_r04.sh), KiB -> bytes, FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md; average over the launches of this instance (256 -> 256 channels, k5, L = 13)",
           "algorithmic_bytes_per_launch": {"activations_in": 54525952, "weights_8_transform_planes_plus_3_taps": 2883584, "output": 54525952, "residual_in_4_of_7_launches": 31157687}}
    lc, la = avg(dom, "SQ_LDS_BANK_CONFLICT"), avg(dom, "SQ_LDS_IDX_ACTIVE")
    if lc is not None and la:
        out["lds_bank_conflict_cycles"], out["lds_idx_active_cycles"], out["lds_conflict_share"] = lc, la, round(lc / la, 4)
    mb, bz = avg(dom, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(dom, "SQ_BUSY_CYCLES")
    if mb and bz:
        out["sq_valu_mfma_busy_cycles"], out["sq_busy_cycles"] = mb, bz
    json.dump(out, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
fe, wr = avg("stem_conv_kernel", "FETCH_SIZE"), avg("stem_conv_kernel", "WRITE_SIZE")
if fe is not None and wr is not None:
    json.dump({"kernel": "cld::stem_conv_kernel", "agents_per_launch": 256,
               "hbm_bytes_per_launch": int(2 * fe * 1024 + wr * 1024), "fetch_bytes_x2_corrected": int(2 * fe * 1024), "write_bytes": int(wr * 1024),
               "algorithmic_bytes_per_launch": {"raster_in": 256 * 34 * 224 * 224 * 4, "weights": 557056, "pooled_output": 256 * 56 * 56 * 64 * 4},
               "note": "average over the dense and the structured raster passes of scripts/ctx_time.py 256; the 9-row strips of "
                       "neighbouring workgroups overlap (2.25x re-read of the raster, served by L2 / Infinity Cache); round 4: the max-pool is "
                       "fused -- the kernel writes the pooled [n,56,56,64] tensor by atomic max (counted as write traffic; up to six updates "
                       "per pooled value) instead of the [n,112,112,64] conv output (822 MB), and the separate max-pool launch is gone"},
              open(os.path.join(DST, "traffic_stem.json"), "w"), indent=1)
chains = {}
# (round 4: the chains in Winograd form; weights = three / four k5 layers' G g at 131,072 B + the direct-form layers; kept slots 4,096 / 2,048 floats per agent)
for kname, algo in (("chain_head_wino_kernel", {"latent_in": 4096 * 52 * 4 * 4, "weights": 3 * 131072 + 8192 + 49152, "output": 4096 * 26 * 64 * 4, "kept_block_input_write_and_read": 2 * 4096 * 4096 * 4}),
                    ("chain_tail_wino_kernel", {"input": 4096 * 26 * 64 * 4, "residual_in": 4096 * 26 * 64 * 4, "weights": 4 * 131072 + 2 * 32768 + 4096, "eps_out": 4096 * 52 * 4 * 4, "kept_block_input_write_and_read": 2 * 4096 * 2048 * 4})):
    fe, wr = avg(kname, "FETCH_SIZE"), avg(kname, "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    chains[kname] = {"hbm_bytes_per_launch": int(2 * fe * 1024 + wr * 1024), "fetch_bytes_x2_corrected": int(2 * fe * 1024), "write_bytes": int(wr * 1024),
                     "algorithmic_bytes_per_launch": algo, "rows_per_launch": 4096}
    for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
        v = avg(kname, c)
        if v is not None:
            chains[kname][c.lower()] = v
if chains:
    json.dump(chains, open(os.path.join(DST, "traffic_chains.json"), "w"), indent=1)
    print(json.dumps(chains, indent=1))
print(open(os.path.join(DST, "traffic.json")).read() if out else "no traffic")
print(open(os.path.join(DST, "traffic_stem.json")).read())

# MFMA-busy share per kernel of the headline bench: SQ_VALU_MFMA_BUSY_CYCLES per launch / (1,024 SIMDs x average duration x 2.4 GHz)
ks = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(DST, "kernel_stats_bench_configs2_steps1_warmup1.csv")))}
with open(os.path.join(DST, "mfma_busy.txt"), "w") as f:
    f.write("MFMA pipe busy share per kernel of `bench.py --steps 1 --warmup 1` (configs[2], 4,096 rows per launch set): SQ_VALU_MFMA_BUSY_CYCLES per launch\n"
            "(pmc_per_launch_avg.csv) / (1,024 SIMDs x average kernel-trace duration x 2.4 GHz); the chip holds ~2.1-2.2 GHz under this load, so the share of the\n"
            "cycles it really ran is ~10 % higher\n")
    for (k, c), (sv, n) in sorted(rows.items()):
        if c != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        d = ks.get(k.split("(")[0])
        if d and any(t in k for t in ("wino1d", "chain", "guide", "conv_block", "conv_pair")):
            f.write(f"{k.split('(')[0][10:]:60s} {d / 1e3:8.1f} us  {sv / n / 1024 / (d * 1e-9 * 2.4e9) * 100:5.1f} %\n")
print(open(os.path.join(DST, "mfma_busy.txt")).read())
