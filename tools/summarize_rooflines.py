#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/collect_rooflines.sh) into

    profiles/<tag>_roofline_counters.json   one entry per (case, kernel): counters per frame-sweep, busy fractions, HBM bytes per launch
    profiles/roofline_counters.json         the copy bench.py and tools/measure_configs.py read (entries keyed by kernel name)
    profiles/<tag>_roofline_counters.md     the same as a table, with the arithmetic spelled out
    profiles/<tag>_kernel_stats_<case>.csv  rocprofv3 --kernel-trace --stats of the same command

Units (MI355X_MICROARCH.md): SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES
count quad-cycles (x4 = cycles) -- except that SQ_ACTIVE_INST_VALU equals the instruction count here; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB, calibrated here on the
known 1 GiB -> 1 GiB copy of the same pass (on gfx950 FETCH_SIZE reads half the bytes of a coalesced stream)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
PEAK_CLOCK = 2.4e9


def short(name):
    name = name.replace("ldpc::(anonymous namespace)::", "").replace("ldpc::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(name):  # cut the argument list: the first '(' outside template brackets
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def pmc(d):
    """{kernel: {counter: [value per dispatch, in dispatch order]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
    return {k: {c: [v for _, v in sorted(rows)] for c, rows in cs.items()} for k, cs in acc.items()}


def durations(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    return {k: [v for _, v in sorted(rows)] for k, rows in acc.items()}


def counted(info, series):
    """The dispatches the driver's frame-sweep count covers: the LAST info['launches'] of the kernel (tools/sim_driver.py runs its warm-up
    launches first and zeroes its counters behind them)."""
    n = int((info or {}).get("launches") or 0)
    if not n or not (info or {}).get("warm_launches"):
        return series
    if isinstance(series, dict):
        return {c: v[-n:] for c, v in series.items()}
    return series[-n:]


def info_of(case, which):
    p = os.path.join(src, case, which + ".info.json")
    return json.load(open(p)) if os.path.exists(p) else None


entries, md = {}, ["# Roofline counters `%s` (MI355X; tools/collect_rooflines.sh -> tools/summarize_rooflines.py)\n" % tag]
for case_dir in sorted(glob.glob(os.path.join(src, "*", ""))):
    case = os.path.basename(os.path.dirname(case_dir))
    stats = glob.glob(os.path.join(case_dir, "stats", "*kernel_stats.csv"))
    if not stats:
        continue
    shutil.copyfile(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, case)))
    inf = info_of(case, "stats") or {}
    md.append("\n## %s: %s %s over %s %.3g, %s, batch %d, backend %s (mean %.2f sweeps/frame)\n" % (
        case, inf.get("code"), inf.get("alg"), inf.get("channel"), inf.get("param", 0), inf.get("precision"), inf.get("batch", 0), inf.get("backend"),
        inf.get("mean_sweeps", 0)))
    md.append("`rocprofv3 --kernel-trace --stats`, un-instrumented clock:\n\n| kernel | calls | avg us | % |\n|---|---|---|---|")
    rows = list(csv.DictReader(open(stats[0])))
    for r in rows[:6]:
        md.append("| %s | %s | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    # ---- LDS / VALU counters of the LDS-resident kernel
    i1 = info_of(case, "sq1")
    if i1 and i1.get("kernel"):
        k = i1["kernel"]
        if k not in pmc(os.path.join(case_dir, "sq1")):  # a family name (sim_driver's "k_ml"): the instantiation the run launched
            fam_hits = sorted((x for x in pmc(os.path.join(case_dir, "sq1")) if x.startswith(k + "<")),
                              key=lambda x: -sum(pmc(os.path.join(case_dir, "sq1"))[x].get("SQ_INSTS_VALU", [0])))
            if fam_hits:
                k = fam_hits[0]
        i2 = info_of(case, "sq2") or i1
        c1, d1 = counted(i1, pmc(os.path.join(case_dir, "sq1")).get(k, {})), counted(i1, durations(os.path.join(case_dir, "sq1")).get(k, []))
        c2, d2 = counted(i2, pmc(os.path.join(case_dir, "sq2")).get(k, {})), counted(i2, durations(os.path.join(case_dir, "sq2")).get(k, []))
        if c1 and d1:
            fs, cus = float(i1["frame_sweeps"]), i1["cus"]
            tot = {c: sum(v) for c, v in c1.items()}
            dur_s = sum(d1) * 1e-9
            clk = tot["GRBM_GUI_ACTIVE"] / 8.0 / dur_s
            e = dict(case=case, kernel=k, workload="%s %s over %s %.3g, %s, batch %d, max_iter %d" % (i1["code"], i1["alg"], i1["channel"], i1["param"],
                                                                                                   i1["precision"], i1["batch"], i1["max_iter"]),
                     launches=len(d1), frame_sweeps=int(fs), mean_sweeps=round(i1["mean_sweeps"], 3),
                     lds_idx_active_per_frame_sweep=round(tot["SQ_LDS_IDX_ACTIVE"] / fs, 2),
                     bank_conflict_per_frame_sweep=round(tot.get("SQ_LDS_BANK_CONFLICT", 0) / fs, 2),
                     insts_lds_per_frame_sweep=round(tot.get("SQ_INSTS_LDS", 0) / fs, 2),
                     insts_valu_per_frame_sweep=round(tot.get("SQ_INSTS_VALU", 0) / fs, 2),
                     sq_active_inst_valu_per_frame_sweep=round(tot.get("SQ_ACTIVE_INST_VALU", 0) / fs, 2),
                     wave_cycles_per_frame_sweep=round(4.0 * tot.get("SQ_WAVE_CYCLES", 0) / fs, 1),
                     kernel_ms_per_launch_in_pmc_pass=round(1e3 * dur_s / len(d1), 4),
                     kernel_ms_per_launch_unprofiled=round(avg_us.get(k, 0) / 1e3, 4),
                     effective_clock_hz_in_pmc_pass=round(clk), cus=cus,
                     lds_busy_frac_in_pmc_pass=round(tot["SQ_LDS_IDX_ACTIVE"] / (dur_s * clk * cus), 4),
                     counters_from="profiles/%s_roofline_counters.json:%s (rocprofv3 --pmc, tools/sim_driver.py: the simulate kernel itself)" % (tag, case))
            if c2 and d2:
                fs2 = float(i2["frame_sweeps"])
                t2 = {c: sum(v) for c, v in c2.items()}
                wc = t2.get("SQ_WAIT_ANY", 0) + t2.get("SQ_WAIT_INST_ANY", 0) + t2.get("SQ_ACTIVE_INST_ANY", 0)
                e.update(wait_any_share=round(t2.get("SQ_WAIT_ANY", 0) / wc, 4) if wc else None,
                         wait_inst_any_share=round(t2.get("SQ_WAIT_INST_ANY", 0) / wc, 4) if wc else None,
                         active_inst_any_share=round(t2.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4) if wc else None,
                         insts_salu_per_frame_sweep=round(t2.get("SQ_INSTS_SALU", 0) / fs2, 2),
                         lds_active_inst_cycles_per_frame_sweep=round(4.0 * t2.get("SQ_ACTIVE_INST_LDS", 0) / fs2, 2))
            # VALU issue cycles.  SQ_ACTIVE_INST_VALU turns out to count one unit per instruction on gfx950 (== SQ_INSTS_VALU), so it is no
            # busy-cycle measure; the fraction below is an ISSUE MODEL on measured instruction counts: 2 cycles per wave64 instruction
            # (v_fma_f32: 2 cycles on a SIMD-32, MI355X_MICROARCH.md), 4 for fp64 add / mul / fma (half rate: 78.6 vs 157.3 TFLOP/s), 8 for
            # transcendentals (quarter rate) -- instruction-mix counters from the sq3 pass where this rocprofv3 has them
            i3 = info_of(case, "sq3") or i1
            c3 = counted(i3, pmc(os.path.join(case_dir, "sq3")).get(k, {}))
            mix = {}
            if c3:
                fs3 = float(i3["frame_sweeps"])
                mix = {c: sum(v) / fs3 for c, v in c3.items()}
            n_all = mix.get("SQ_INSTS_VALU", e["insts_valu_per_frame_sweep"])
            n_f64 = sum(mix.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
            n_tr = mix.get("SQ_INSTS_VALU_TRANS_F32", 0.0) + mix.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
            e["valu_mix_per_frame_sweep"] = {c: round(v, 2) for c, v in mix.items()} or None
            e["valu_issue_cycles_per_frame_sweep"] = round(2.0 * (n_all - n_f64 - n_tr) + 4.0 * n_f64 + 8.0 * n_tr, 1)
            e["valu_issue_model"] = "2 cycles per wave64 VALU instruction, 4 per fp64 add/mul/fma, 8 per transcendental; counts: SQ_INSTS_VALU*"
            e["valu_active_cycles_per_frame_sweep"] = e["valu_issue_cycles_per_frame_sweep"]  # the name bench.py reads
            e["valu_busy_frac_in_pmc_pass"] = round(e["valu_issue_cycles_per_frame_sweep"] * fs / (dur_s * clk * cus * 4), 4)
            # Which unit of the LDS path binds?  SQ_LDS_IDX_ACTIVE counts cycles of the LDS ARRAY; a store also has to move its address and data
            # registers to the LDS (MI355X_MICROARCH.md, LDS table: ds_write_b64 ~6 cycles of that transfer against 4 array cycles, ds_write_b32 4
            # against 2, ds_write_addtid_b32 2 against 2, reads 2), a path the array counter does not see.  Model: measured load / store
            # instruction counts (SQ_INSTS_LDS_LOAD / _STORE, sq4 pass) x the table's per-instruction cycles for the widths THIS kernel uses.
            i4 = info_of(case, "sq4") or i1
            c4 = counted(i4, pmc(os.path.join(case_dir, "sq4")).get(k, {}))
            if c4:
                fs4 = float(i4["frame_sweeps"])
                t4 = {c: sum(v) / fs4 for c, v in c4.items()}
                ld, st_, at = t4.get("SQ_INSTS_LDS_LOAD", 0.0), t4.get("SQ_INSTS_LDS_STORE", 0.0), t4.get("SQ_INSTS_LDS_ATOMIC", 0.0)
                # per-instruction cycles on the issue / transfer path by kernel family: (read, row store, note)
                fam = ("k_fused_f64", 2.0, 6.0, "ds_read_b64 2, ds_write_b64 6") if k.startswith("k_fused_f64") else \
                      ("k_admm_lds", 2.0, 6.0, "ds_read_b64 2, ds_write_b64 6") if k.startswith("k_admm_lds") else \
                      ("k_fused_becs", 2.0, 6.0, "ds_read_b64 2, ds_write_b64 6") if k.startswith("k_fused_becs") else \
                      ("k_fused_bp 16-wave", 2.0, 3.0, "ds_read_b32 2, ds_write2st64_b32 6 per two rows / addtid 2") if ", 16, " in k else \
                      ("k_fused_bp", 2.0, 2.0, "ds_read_b32 2, ds_write_addtid_b32 2")
                path = ld * fam[1] + st_ * fam[2] + at * 8.0
                # the same split with the constants MEASURED on this chip for 8-byte elements (profiles/r04_lds_store_path.txt: ds_read_b64 2.55,
                # ds_write_b64 6.3 cycles per wave-instruction per CU, ds_add_f64 8.07; a 2-load + 1-store mix runs at their sum: the path is additive)
                if fam[0] in ("k_fused_f64", "k_fused_becs", "k_admm_lds"):
                    e["lds_path_cycles_per_frame_sweep_measured_constants"] = round(ld * 2.55 + st_ * 6.3 + at * 8.07, 1)
                    e["lds_path_measured_constants"] = "ds_read_b64 2.55, ds_write_b64 6.3"
                elif fam[0] == "k_fused_bp":  # 4-byte elements: ds_read_b32 2.88, ds_write_addtid_b32 2.2 (third run of the same file)
                    e["lds_path_cycles_per_frame_sweep_measured_constants"] = round(ld * 2.88 + st_ * 2.2 + at * 8.0, 1)
                    e["lds_path_measured_constants"] = "ds_read_b32 2.88, ds_write_addtid_b32 2.2"
                e.update(lds_loads_per_frame_sweep=round(ld, 2), lds_stores_per_frame_sweep=round(st_, 2), lds_atomics_per_frame_sweep=round(at, 2),
                         lds_path_cycles_per_frame_sweep=round(path, 1), lds_path_model=fam[3],
                         lds_data_fifo_full_per_frame_sweep=round(t4.get("SQ_LDS_DATA_FIFO_FULL", 0.0), 2),
                         lds_cmd_fifo_full_per_frame_sweep=round(t4.get("SQ_LDS_CMD_FIFO_FULL", 0.0), 2),
                         lds_addr_conflict_per_frame_sweep=round(t4.get("SQ_LDS_ADDR_CONFLICT", 0.0), 2),
                         wait_inst_lds_quad_cycles_per_frame_sweep=round(t4.get("SQ_WAIT_INST_LDS", 0.0), 2))
            entries.setdefault(k, e)
            entries["%s:%s" % (case, k)] = e
            # what the un-instrumented launch makes of it: busy LDS-array cycles / (CUs x 2.4 GHz)
            if e["kernel_ms_per_launch_unprofiled"]:
                fsps = (fs / len(d1)) / (e["kernel_ms_per_launch_unprofiled"] * 1e-3)
                e["lds_frac_at_peak_clock_unprofiled"] = round(fsps * e["lds_idx_active_per_frame_sweep"] / (cus * PEAK_CLOCK), 4)
                e["valu_frac_at_peak_clock_unprofiled"] = round(fsps * e["valu_active_cycles_per_frame_sweep"] / (cus * 4 * PEAK_CLOCK), 4)
                e["frame_sweeps_per_s_unprofiled"] = round(fsps, 1)
            md.append("\n`%s`: %d launches, %.4g frame-sweeps.  Per frame-sweep: **%.1f LDS-array cycles** (SQ_LDS_IDX_ACTIVE; %.1f of them bank "
                      "conflicts), %.1f DS + %.1f VALU instructions, %.1f VALU-busy cycles.  In the PMC pass (%.3f ms/launch, %.3f GHz effective): LDS "
                      "array %.3f busy, VALU %.3f busy; un-instrumented launch %.3f ms => **LDS %.3f, VALU %.3f of peak at 2.4 GHz**." % (
                          k, len(d1), fs, e["lds_idx_active_per_frame_sweep"], e["bank_conflict_per_frame_sweep"], e["insts_lds_per_frame_sweep"],
                          e["insts_valu_per_frame_sweep"], e["valu_active_cycles_per_frame_sweep"], e["kernel_ms_per_launch_in_pmc_pass"], clk / 1e9,
                          e["lds_busy_frac_in_pmc_pass"], e["valu_busy_frac_in_pmc_pass"], e["kernel_ms_per_launch_unprofiled"],
                          e.get("lds_frac_at_peak_clock_unprofiled", 0), e.get("valu_frac_at_peak_clock_unprofiled", 0)))
            if "lds_path_cycles_per_frame_sweep" in e and e.get("frame_sweeps_per_s_unprofiled"):
                fsps = e["frame_sweeps_per_s_unprofiled"]
                arr, pth, val = e["lds_frac_at_peak_clock_unprofiled"], round(fsps * e["lds_path_cycles_per_frame_sweep"] / (cus * PEAK_CLOCK), 4), e["valu_frac_at_peak_clock_unprofiled"]
                e["lds_path_frac_at_peak_clock_unprofiled"] = pth
                e["binding_unit"] = max((("lds_array", arr), ("lds_store_path", pth), ("valu", val)), key=lambda t: t[1])[0]
                if "lds_path_cycles_per_frame_sweep_measured_constants" in e:
                    e["lds_path_frac_measured_constants"] = round(fsps * e["lds_path_cycles_per_frame_sweep_measured_constants"] / (cus * PEAK_CLOCK), 4)
                md.append("LDS split: %.1f loads + %.1f stores + %.1f atomics per frame-sweep -> issue / transfer path %.1f cycles (%s) = **%.3f** of "
                          "peak; array %.3f, VALU %.3f => binding unit: **%s**.  Queue counters per frame-sweep: data FIFO full %.2f, command FIFO "
                          "full %.2f, address conflicts %.2f, SQ_WAIT_INST_LDS %.2f quad-cycles." % (
                              e["lds_loads_per_frame_sweep"], e["lds_stores_per_frame_sweep"], e["lds_atomics_per_frame_sweep"], e["lds_path_cycles_per_frame_sweep"],
                              e["lds_path_model"], pth, arr, val, e["binding_unit"], e["lds_data_fifo_full_per_frame_sweep"],
                              e["lds_cmd_fifo_full_per_frame_sweep"], e["lds_addr_conflict_per_frame_sweep"], e["wait_inst_lds_quad_cycles_per_frame_sweep"]))
                if "lds_path_frac_measured_constants" in e:
                    md.append("With the per-instruction cycles MEASURED on this chip (profiles/r04_lds_store_path.txt: %s) the same split is %.1f cycles per "
                              "frame-sweep = **%.3f** of the path's capacity." % (
                                  e.get("lds_path_measured_constants", ""), e["lds_path_cycles_per_frame_sweep_measured_constants"], e["lds_path_frac_measured_constants"]))
            if "wait_any_share" in e:
                md.append("Wave time: %.0f %% waiting (s_waitcnt / barrier), %.0f %% issue stalls, %.0f %% issuing." % (
                    100 * (e["wait_any_share"] or 0), 100 * (e["wait_inst_any_share"] or 0), 100 * (e["active_inst_any_share"] or 0)))
    # ---- HBM bytes per launch of every kernel of the case
    fs_, ws_ = pmc(os.path.join(case_dir, "FETCH_SIZE")), pmc(os.path.join(case_dir, "WRITE_SIZE"))
    if fs_ and ws_:
        gib = float(1 << 30)
        cal_f = (fs_.get("k_copy4", {}).get("FETCH_SIZE") or [0])[-1]
        cal_w = (ws_.get("k_copy4", {}).get("WRITE_SIZE") or [0])[-1]
        kf = gib / (cal_f * 1024) if cal_f else 2.0
        kw = gib / (cal_w * 1024) if cal_w else 1.0
        md.append("\nHBM traffic (separate --pmc passes; calibration copy of 1 GiB: FETCH_SIZE %.0f KiB => x%.3f, WRITE_SIZE %.0f KiB => x%.3f):\n" % (cal_f, kf, cal_w, kw))
        md.append("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch | algorithmic bytes/launch | traffic / algorithmic |\n|---|---|---|---|---|---|---|")
        ih = info_of(case, "FETCH_SIZE") or inf
        s = 8 if ih.get("precision") == "f64" else 4
        E, n = ih.get("E", 0), ih.get("n", 0)
        sweeps = None
        for k in sorted(fs_):
            if not k.startswith("k_") or k.startswith("k_copy4") or len(fs_[k].get("FETCH_SIZE", [])) == 0:
                continue
            f_, w_ = fs_[k]["FETCH_SIZE"], ws_.get(k, {}).get("WRITE_SIZE", [0])
            fa, wa = sum(f_) / len(f_), sum(w_) / max(len(w_), 1)
            tot_b = fa * 1024 * kf + wa * 1024 * kw
            alg = None
            if k.startswith("k_cn<") or k.startswith("k_vn<"):
                # frames live in a launch vary with early termination: use launches at full batch only when no frame leaves (1.0 dB cases)
                frames = ih.get("batch", 0)
                alg = frames * s * ((2 * E + n) if k.startswith("k_cn<") else (E + 2 * n))
            elif k.startswith("k_cn16<") or k.startswith("k_vn16<"):  # fp16 storage, two-array sweep: 2-byte messages, 4-byte priors
                alg = ih.get("batch", 0) * ((4 * E) if k.startswith("k_cn16<") else (4 * E + 4 * n))
            elif k.startswith("k_becs_cn") or k.startswith("k_becs_vn"):  # bit-sliced erasure passes: 8-byte elements per 32 frames
                m_ = ih.get("m", 0)
                alg = ih.get("batch", 0) * ((E + m_) if k.startswith("k_becs_cn") else (3 * E + 3 * n)) // 4
            ent = dict(case=case, kernel=k, launches=len(f_), fetch_kib_per_launch=round(fa, 1), write_kib_per_launch=round(wa, 1),
                       hbm_bytes_per_launch=int(tot_b), calibration=dict(fetch=round(kf, 4), write=round(kw, 4)),
                       compulsory_bytes_per_launch=alg, traffic_over_compulsory=round(tot_b / alg, 4) if alg else None,
                       workload="%s %s %s batch %d" % (ih.get("code"), ih.get("alg"), ih.get("precision"), ih.get("batch", 0)))
            entries["hbm:%s:%s" % (case, k)] = ent
            md.append("| %s | %d | %.0f | %.0f | %.4g | %s | %s |" % (k, len(f_), fa, wa, tot_b, ("%.4g" % alg) if alg else "-", ("%.3f" % (tot_b / alg)) if alg else "-"))
        md.append("\n(check / variable pass: compulsory bytes = batch x s x (2E + n) / batch x s x (E + 2n) -- c2v in + out + each marginal once; "
                  "c2v in + prior in + marginal out -- valid where no frame leaves early; the section-8(d) model prices the pair at s(4E + n).)")

# Stamp every entry with the identity of the code it was measured on: sha256 of the library, and -- per kernel -- a hash of that kernel's
# machine code + descriptor (tools/kernel_resources.py kernel_code_hashes).  bench.py compares the latter with the library it has loaded
# and prints `counters_stale: true` (frac null) when the kernel body changed under an unchanged name.  This script runs ON THE BOX right
# behind the collection (tools/round_measure.sh), i.e. on the very library the counters were collected with.
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

lib = os.environ.get("LDPC_LIB_PATH") or kernel_resources.DEFAULT_LIB
hashes, lib_sha = kernel_resources.kernel_code_hashes(lib), kernel_resources.lib_sha256(lib)
head = os.environ.get("LDPC_HEAD")  # .git does not travel to the GPU box: tools/round_measure.sh passes the HEAD it was started from
if not head:
    try:
        import subprocess

        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        head = None
for key, ent in entries.items():
    ent["kernel_code_sha"] = hashes.get(ent["kernel"])
    ent["lib_sha256"] = lib_sha
    ent["head"] = head
os.makedirs(dst, exist_ok=True)
json.dump(entries, open(os.path.join(dst, "%s_roofline_counters.json" % tag), "w"), indent=1)
json.dump(entries, open(os.path.join(dst, "roofline_counters.json"), "w"), indent=1)
open(os.path.join(dst, "%s_roofline_counters.md" % tag), "w").write("\n".join(md) + "\n")
print("\n".join(md)[:8000])
