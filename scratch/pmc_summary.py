"""Summarise the three PMC passes of one workload: python scratch/pmc_summary.py <dir> <c3|c4>
Reads <dir>/pmc_<wl>_{mfma,fetch,write}_raw.csv, writes <dir>/r03_pmc_traffic_<wl>.json."""
import collections, csv, json, sys
out, wl = sys.argv[1], sys.argv[2]
PASSES = 3        # --steps 1 --warmup 0: the pass that counts launches, one timed pass, the pass that times the halves apart
# kernels whose streaming reads are 16 bytes per lane: gfx950 tallies those at half their bytes
WIDE = ("gemm_kernel", "gemm_dma_kernel", "rows_gemm_kernel", "gru_mid_wgrad_kernel", "gru_mid_bwd_kernel", "gru_mid96_fwd_kernel", "gru_mid96_bwd_kernel", "gru_mid_fwd_kernel", "gru_persist_fwd_fused_kernel", "gru_persist_fwd_dma_kernel", "gru_persist_bwd_res_kernel")
DESC = {"c3": "c3 (B1024 T24 H96 L3)", "c4": "c4 (B2048 T256 H512 L3)"}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for name in ("mfma", "fetch", "write"):
    for r in csv.DictReader(open(f"{out}/pmc_{wl}_{name}_raw.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
kern = {}
tot_f = tot_w = 0.0
for k, v in agg.items():
    if not k.startswith("tgru::"):
        continue
    e = {"dispatches_per_pass": cnt[k] / PASSES}
    if v.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_fraction"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8), 4)
    f = v.get("FETCH_SIZE", 0.0) * 1024 / PASSES      # the counters are reported in KB
    w = v.get("WRITE_SIZE", 0.0) * 1024 / PASSES
    wide = any(n in k for n in WIDE)
    fc = 2 * f if wide else f
    e.update(fetch_bytes_per_pass_raw=f, fetch_doubled=wide, fetch_bytes_per_pass_corrected=fc, write_bytes_per_pass=w)
    tot_f += fc; tot_w += w
    kern[k] = e
res = {"workload": DESC[wl],
       "command": f"rocprofv3 --pmc <set> -- python3 bench.py --only --workload {wl} --steps 1 --warmup 0 --no-cpu, one run per counter set "
                  "(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE); three forward+backward passes per run",
       "normalisation": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); bytes per pass = counter KB x 1024 / 3; "
                        "FETCH_SIZE doubled for the kernels whose streaming reads are 16 bytes per lane (gfx950 tallies those at half their "
                        "bytes), taken as reported for the others (4-byte-per-lane reads: uncalibrated)",
       "kernels": kern, "hbm_bytes_per_step": tot_f + tot_w,
       "fetch_bytes_per_step_corrected": tot_f, "write_bytes_per_step": tot_w}
json.dump(res, open(f"{out}/r03_pmc_traffic_{wl}.json", "w"), indent=1)
for k, e in sorted(kern.items(), key=lambda kv: -kv[1].get("write_bytes_per_pass", 0)):
    print(k, e)
print(wl, "HBM bytes per step", tot_f + tot_w)
