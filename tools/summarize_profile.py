#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_summary.md, the kernel-stats CSVs, the bench
JSON lines and profiles/<tag>_hbm_traffic.json (HBM bytes per launch from the PMC counters, keyed
"<pass>:<precision>:<kernel>", pass in {fused, stream, sim})."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
out = ["# rocprofv3 summary `%s` (MI355X, collected by tools/collect_profiles.sh)\n" % tag]
KERNELS = ("k_fused", "k_cn", "k_vn", "k_biawgn", "k_count", "k_load", "k_unpack", "k_syndrome")


def short(name):
    name = name.replace("ldpc::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def pmc(dirname):
    f = glob.glob(os.path.join(src, dirname, "*counter_collection.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def kernel_durations(dirname):
    """Kernel name -> list of dispatch durations (ns) from the kernel trace of a pass."""
    f = glob.glob(os.path.join(src, dirname, "*kernel_trace.csv"))
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            acc[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


traffic, lds = {}, {}
for prec in ("f64", "f32"):
    for bk in ("fused", "stream"):
        f = glob.glob(os.path.join(src, "stats_%s_%s" % (bk, prec), "*kernel_stats.csv"))
        if not f:
            continue
        shutil.copyfile(f[0], os.path.join(dst, "%s_kernel_stats_%s_%s.csv" % (tag, bk, prec)))
        out.append("\n## `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --points --no-profile "
                   "--precision %s --backend %s`\n" % (prec, "auto" if bk == "fused" else "stream"))
        out.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(f[0])))[:7]:
            out.append("| %s | %s | %.3f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                           r["Percentage"]))
    # HBM traffic: calibrate on the known 1 GiB -> 1 GiB 4-byte-per-lane copy of the same run, then report per launch
    kf = kw = float("nan")
    for bk in ("fused", "stream"):
        fs, ws = pmc("pmc_%s_%s_FETCH_SIZE" % (bk, prec)), pmc("pmc_%s_%s_WRITE_SIZE" % (bk, prec))
        if not fs or not ws:
            continue
        cal_f = fs.get("k_copy4", {}).get("FETCH_SIZE", [0])[-1]
        cal_w = ws.get("k_copy4", {}).get("WRITE_SIZE", [0])[-1]
        gib = float(1 << 30)
        kf = gib / (cal_f * 1024) if cal_f else float("nan")  # true bytes per counted KiB, reads
        kw = gib / (cal_w * 1024) if cal_w else float("nan")
        out.append("\n## HBM traffic, %s, backend %s (`tools/prof_fused.py --reps 1`; separate --pmc passes; counters are KiB)\n" % (prec, bk))
        out.append("calibration on a known 1 GiB -> 1 GiB 4-byte-per-lane copy: FETCH_SIZE = %.0f KiB (true/reported = %.3f), WRITE_SIZE = %.0f KiB "
                   "(true/reported = %.3f)\n" % (cal_f, kf, cal_w, kw))
        out.append("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | corrected HBM bytes/launch |\n|---|---|---|---|---|")
        for k in fs:
            if k.startswith(KERNELS):
                f_ = fs[k]["FETCH_SIZE"]
                w_ = ws.get(k, {}).get("WRITE_SIZE", [0])
                fa, wa = sum(f_) / len(f_), sum(w_) / max(len(w_), 1)
                tot = fa * 1024 * kf + wa * 1024 * kw
                traffic["%s:%s:%s" % (bk, prec, k)] = tot
                out.append("| %s | %d | %.0f | %.0f | %.4g |" % (k, len(f_), fa, wa, tot))
    # the kernels of the bench step
    fs, ws = pmc("pmc_sim_%s_FETCH_SIZE" % prec), pmc("pmc_sim_%s_WRITE_SIZE" % prec)
    if fs and ws:
        out.append("\n## HBM traffic of the kernels of the bench step, %s (`bench.py --steps 2 --warmup 1 --no-profile --precision %s`; FETCH_SIZE x2 on "
                   "gfx950, WRITE_SIZE exact)\n" % (prec, prec))
        out.append("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch |\n|---|---|---|---|---|")
        for k in fs:
            if k.startswith(KERNELS):
                f_ = fs[k]["FETCH_SIZE"]
                w_ = ws.get(k, {}).get("WRITE_SIZE", [0])
                fa, wa = sum(f_) / len(f_), sum(w_) / max(len(w_), 1)
                tot = fa * 1024 * 2.0 + wa * 1024 * 1.0
                traffic["sim:%s:%s" % (prec, k)] = tot
                out.append("| %s | %d | %.0f | %.0f | %.4g |" % (k, len(f_), fa, wa, tot))
    for d in ("pmc_fused_%s_SQ" % prec, "pmc_fused_%s_SQ2" % prec):
        acc = pmc(d)
        for k, v in acc.items():
            if k.startswith("k_fused"):
                out.append("\n## SQ counters of %s (%s, `tools/prof_fused.py --reps 1 --precision %s`, summed over %d launches of 65 536 frames)\n" % (
                    k, d, prec, len(next(iter(v.values())))))
                for c, vals in sorted(v.items()):
                    out.append("- %s = %.4g" % (c, sum(vals)))
                info_path = os.path.join(src, d + ".info.json")
                if "SQ_LDS_IDX_ACTIVE" in v and os.path.exists(info_path):
                    # LDS roofline inputs of bench.py: LDS-array cycles per frame-sweep, and the clock the profiled launches ran at
                    info = json.load(open(info_path))
                    nl = len(v["SQ_LDS_IDX_ACTIVE"])
                    fs = info["frame_sweeps_per_launch"] * nl
                    dur = kernel_durations(d).get(k, [])
                    dur_s = sum(dur) * 1e-9
                    cus = info.get("cus", 256)
                    clk = None
                    if dur_s > 0 and "GRBM_GUI_ACTIVE" in v:
                        clk = sum(v["GRBM_GUI_ACTIVE"]) / 8 / dur_s  # the counter is summed over the 8 XCDs
                    elif dur_s > 0 and "SQ_BUSY_CYCLES" in v:
                        clk = sum(v["SQ_BUSY_CYCLES"]) / 32 / dur_s  # summed over the 32 shader engines
                    entry = {"kernel": k, "launches": nl, "frame_sweeps": fs, "workload": "%s %s %.1f dB batch %d" % (info["code"], info["alg"], info["snr"], info["batch"]),
                             "lds_idx_active_per_frame_sweep": round(sum(v["SQ_LDS_IDX_ACTIVE"]) / fs, 2),
                             "bank_conflict_per_frame_sweep": round(sum(v.get("SQ_LDS_BANK_CONFLICT", [0])) / fs, 2),
                             "insts_lds_per_frame_sweep": round(sum(v.get("SQ_INSTS_LDS", [0])) / fs, 2),
                             "insts_valu_per_frame_sweep": round(sum(v.get("SQ_INSTS_VALU", [0])) / fs, 2),
                             "kernel_ms_per_launch_in_pmc_pass": round(1e3 * dur_s / max(nl, 1), 4) if dur_s else None,
                             "effective_clock_hz": round(clk, 0) if clk else None,
                             "effective_clock_from": "GRBM_GUI_ACTIVE / 8 XCDs / kernel time" if (dur_s > 0 and "GRBM_GUI_ACTIVE" in v) else "SQ_BUSY_CYCLES / 32 SEs / kernel time",
                             "lds_pipe_busy_in_pmc_pass": round(sum(v["SQ_LDS_IDX_ACTIVE"]) / (dur_s * clk * cus), 4) if (dur_s and clk) else None}
                    lds["%s:%s" % (prec, k)] = entry
                    out.append("\nLDS roofline inputs: `%s`" % json.dumps(entry))

for name in ("bench.json", "bench_f32.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copyfile(p, os.path.join(dst, "%s_%s" % (tag, name)))
        d = json.load(open(p))
        keep = ("value", "unit", "dtype", "ms_per_step", "mean_sweeps", "roofline", "decode_from_hbm", "fp32_mode", "points", "roofline_streaming_backend",
                "cpu_baseline", "kernel_ms_per_step", "host_overhead_ms_per_step")
        out.append("\n## %s\n\n```json\n%s\n```" % (name, json.dumps({k: d[k] for k in keep if k in d}, indent=1)))
json.dump(traffic, open(os.path.join(dst, "%s_hbm_traffic.json" % tag), "w"), indent=1)
if lds:
    json.dump(lds, open(os.path.join(dst, "%s_lds_cycles.json" % tag), "w"), indent=1)
    json.dump(lds, open(os.path.join(dst, "lds_cycles.json"), "w"), indent=1)  # the copy bench.py reads
if traffic:
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:6000])
