#!/usr/bin/env python
"""Summarises the rocprofv3 PMC passes of tools/pmc_roofline.sh: per-launch FETCH_SIZE / WRITE_SIZE (KB as rocprofv3 reports them), the
gfx950 correction (FETCH_SIZE under-reports wide coalesced reads by 2x: MI355X_MICROARCH.md), MFMA-busy fraction, effective clock.
python tools/pmc_summary.py <dir with *_counter_collection.csv / *_kernel_trace.csv> <output prefix>"""
import csv, glob, json, os, sys
d, out = sys.argv[1], sys.argv[2]
PREC = sys.argv[3] if len(sys.argv) > 3 else "fp32"
X3 = PREC == "f32x3"
S16 = PREC in ("bf16", "fp16")           # 16-bit STORED operands (tf_gemm16_nt_f32): one v_mfma_f32_32x32x16_{bf16,f16} per 32x32x16 product
PEAK = 2500.0 / 6 if X3 else 2500.0 if S16 else 157.3       # f32x3: six bf16 MFMA products per fp32 product, dense bf16 / fp16 peak 2.5 PF
M, N, K = 1740, 6048, 1512
vals, durs, kname = {}, {}, None
for cc in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    tag = os.path.basename(cc)[:-len("_counter_collection.csv")]
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, tag + "_kernel_trace.csv")))}
    for r in csv.DictReader(open(cc)):
        if "gemm" not in r["Kernel_Name"]:
            continue
        kname = r["Kernel_Name"]
        k = kt[r["Dispatch_Id"]]
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        durs.setdefault(r["Counter_Name"], []).append((int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3)
avg = lambda v: sum(v[2:]) / max(1, len(v[2:]))          # skip the two warm-up launches
A = {k: avg(v) for k, v in vals.items()}
D = {k: avg(v) for k, v in durs.items()}
algo = (M * K + N * K) * (2 if S16 else 4) + (M * N + N) * 4
lines = ["# rocprofv3 --pmc <one pass per line> --kernel-trace -- python tools/gemm_tuned.py %d %d %d nt 10 %s   (tools/pmc_roofline.sh)" % (M, N, K, PREC),
         "# kernel: %s" % kname, "# per-launch averages over 10 launches (2 warm-ups dropped); algorithmic bytes %.1f MB (A + B + C + bias)" % (algo / 1e6)]
for k in sorted(A):
    lines.append("%-28s %16.1f   (avg kernel duration in that pass %.1f us)" % (k, A[k], D[k]))
res = dict(kernel=kname, algorithmic_bytes=algo)
if "FETCH_SIZE" in A and "WRITE_SIZE" in A:
    traffic = (2 * A["FETCH_SIZE"] + A["WRITE_SIZE"]) * 1024
    res.update(fetch_size_kb=A["FETCH_SIZE"], write_size_kb=A["WRITE_SIZE"], traffic_bytes_per_launch=traffic, traffic_over_algorithmic=traffic / algo,
               avg_duration_us=D["FETCH_SIZE"])
    lines.append("traffic = 2 x FETCH_SIZE + WRITE_SIZE = %.1f MB per launch = %.2f x algorithmic (fabric-side L2 requests incl. Infinity-Cache hits: "
                 "the operands, 47 MB, stay resident in the 256 MB MALL)" % (traffic / 1e6, traffic / algo))
if "SQ_VALU_MFMA_BUSY_CYCLES" in A and "GRBM_GUI_ACTIVE" in A:
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles the MFMA pipe is busy, summed over SIMDs... normalise by GUI_ACTIVE (summed over 8 XCDs) x SIMDs per XCD
    clk = A["GRBM_GUI_ACTIVE"] / 8 / (D["GRBM_GUI_ACTIVE"] * 1e3)
    res.update(effective_clock_ghz=clk, clock_adjusted_peak_tflops=PEAK * clk / 2.4, precision=PREC, peak_tflops=PEAK)
    lines.append("effective clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) -> clock-adjusted %s peak %.1f TF/s; achieved %.1f TF/s" %
                 (clk, "bf16-MFMA / 6 (f32x3)" if X3 else "16-bit MFMA" if S16 else "fp32 MFMA", PEAK * clk / 2.4, 2.0 * M * N * K / D["GRBM_GUI_ACTIVE"] / 1e6))
    busy = A["SQ_VALU_MFMA_BUSY_CYCLES"]
    ideal = 2.0 * M * N * K / 4096 * 64 / 1.0        # MFMA 32x32x2 = 4096 FLOP, 64 cycles of one SIMD's pipe
    if X3:
        ideal = 6 * 2.0 * M * N * K / 32768 * 32     # six v_mfma_f32_32x32x16_bf16 (32768 FLOP, 32 cycles) per 32x32x16 product
    if S16:
        ideal = 2.0 * M * N * K / 32768 * 32
    res.update(mfma_busy_cycles=busy, mfma_ideal_simd_cycles=ideal)
    for denom_name, denom in (("SQ_BUSY_CYCLES", A.get("SQ_BUSY_CYCLES")),):
        if denom:
            lines.append("SQ_VALU_MFMA_BUSY_CYCLES / %s = %.3f ; algorithmic MFMA SIMD-cycles %.3e vs counter %.3e (ratio %.2f)" % (denom_name, busy / denom, ideal, busy, busy / ideal))
    res["mfma_busy_frac"] = busy / (A["GRBM_GUI_ACTIVE"] / 8 * 1024) if A["GRBM_GUI_ACTIVE"] else None   # 1024 SIMDs; only meaningful if the counter is per-SIMD summed
res["note"] = "2*FETCH_SIZE + WRITE_SIZE of the tuned plan (rocprofv3 --pmc, separate passes; fabric-side incl. Infinity-Cache hits); algorithmic %.1f MB" % (algo / 1e6)
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
