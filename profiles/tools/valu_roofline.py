#!/usr/bin/env python
"""VALU-issue roofline of the blend kernels from one profiles/tools/valu_roofline.sh run.

  issue cycles per launch = sum over instruction classes of  count(class) x cycles(class)
     count:  SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F32, _INT32, _CVT and the remainder of SQ_INSTS_VALU (moves, selects, compares,
             DPP / permlane: "other"), per launch (hardware counters, --pmc pass A); a packed v_pk_*_f32 counts once, so each
             fp32 class is split by the packed fraction of the kernel's blend loop (profiles/r5_isa_static_mix.json, isa_mix.py)
     cycles: SIMD cycles per wave64 instruction at 8 waves per SIMD from profiles/tools/valu_issue_rate.hip, run in the same
             session (wall time x 2.4 GHz: the same normalisation as the peak below)
  peak    = 256 CUs x 4 SIMDs x 2.4 GHz x (isolated launch duration from the kernel trace, one raster stream)
  frac    = issue cycles / peak      -- the fraction of the chip's VALU issue slots the launch fills
Beside it the hardware's own view: SQ_ACTIVE_INST_VALU (quad-cycles a wave spends in VALU instructions) x 4 / (1024 SIMDs x
GRBM_GUI_ACTIVE per XCD) and SQ_ACTIVE_INST_VALU2 (quad-cycles in which two VALU instructions issue).
Usage: python profiles/tools/valu_roofline.py gpurun_out/valu_<tag> > profiles/r5_valu_roofline.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNELS = {"fwd": "fs::sort_blend_kernel<false, false>", "trn": "fs::render_bwd_kernel<false, false>"}
ALSO = {"trn": ["fs::sort_blend_kernel<false, true>"]}


def kname(s):
    return s.split("(")[0].replace("void ", "").strip()


def counters(d):
    """{kernel: {counter: mean per launch}}; a dispatch's value = the sum of its rows (one per instance the tool reports)."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[(kname(r["Kernel_Name"]), r["Counter_Name"], r.get("Dispatch_Id", r.get("Correlation_Id", "0")))][0] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, c, _), v in per.items():
        agg[k][c].append(v[0])
    return {k: {c: sum(v) / len(v) for c, v in d_.items()} for k, d_ in agg.items()}


def durations(d):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[kname(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}      # median, ms


def main(src):
    rate_txt = open(os.path.join(src, "valu_issue_rate.txt")).read()
    rates = json.loads(next(l for l in rate_txt.splitlines() if l.startswith("JSON "))[5:])
    cyc = rates["cycles_at_2.4GHz"]
    mix = json.load(open(os.path.join(ROOT, "profiles", "r5_isa_static_mix.json")))["kernels"]
    # "other" = moves, selects, compares, DPP moves.  Selects are priced in the VOP3 form with an SGPR-pair mask (2.7 cycles: what the
    # blend loops use, profiles/r5_isa_static_mix.json); the VOP2 form reading VCC measured 11.3 cycles and is listed, not used.
    cnd = cyc.get("v_cndmask_b32_e64 (SGPR mask)", cyc["v_cndmask_b32"])
    other_cost = (cyc["v_mov_b32"] + cnd + cyc["v_cmp_gt_f32"] + cyc["v_mov_b32_dpp"]) / 4
    int_cost = sum(cyc[k] for k in ("v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mad_u32_u24", "v_bfe_u32")) / 5
    trans_cost = sum(cyc[k] for k in ("v_exp_f32", "v_rcp_f32")) / 2
    out = {"what": __doc__.split("\n\n")[0], "issue_cycles_per_instruction_at_2.4GHz": cyc, "counter_ghz": rates["counter_ghz"],
           "class_cost_used": {"other (mov, cndmask, cmp, dpp: mean)": other_cost, "int32 (mean of 5)": int_cost,
                               "trans (mean of exp, rcp)": trans_cost}, "kernels": {}}
    for mode, kern in KERNELS.items():
        ca, cb = counters(os.path.join(src, mode + "_a")), counters(os.path.join(src, mode + "_b"))
        dur = durations(os.path.join(src, mode + "_trace"))
        for k in [kern] + ALSO.get(mode, []):
            a, b = ca.get(k), cb.get(k)
            if not a or k not in dur:
                print("missing", k, file=sys.stderr)
                continue
            short = k.replace("fs::", "")
            pk = (mix.get(short, {}).get("hot_loop") or {}).get("packed_fraction", {"fma_f32": 0, "mul_f32": 0, "add_f32": 0})
            n = {c: a.get("SQ_INSTS_VALU_" + c, 0.0) for c in ("FMA_F32", "MUL_F32", "ADD_F32", "TRANS_F32", "INT32", "CVT")}
            total = a["SQ_INSTS_VALU"]
            n_other = max(total - sum(n.values()), 0.0)
            cost = 0.0
            parts = {}
            for c, (s_name, p_name) in {"FMA_F32": ("v_fma_f32", "v_pk_fma_f32"), "MUL_F32": ("v_mul_f32", "v_pk_mul_f32"),
                                        "ADD_F32": ("v_add_f32", "v_pk_add_f32")}.items():
                f = pk[c.lower()]
                parts[c] = n[c] * ((1 - f) * cyc[s_name] + f * cyc[p_name])
            parts["TRANS_F32"] = n["TRANS_F32"] * trans_cost
            parts["INT32"] = n["INT32"] * int_cost
            parts["CVT"] = n["CVT"] * cyc["v_cvt_f32_u32"]
            parts["other"] = n_other * other_cost
            cost = sum(parts.values())
            t_ms = dur[k]
            peak = 1024 * 2.4e9 * t_ms * 1e-3
            e = {"isolated_launch_ms": t_ms, "SQ_INSTS_VALU": total, "by_class": dict(n, other=n_other), "packed_fraction_of_blend_loop": pk,
                 "issue_cycles": cost, "issue_cycles_by_class": parts, "mean_cycles_per_valu_instruction": cost / max(total, 1),
                 "peak_simd_cycles": peak, "frac": cost / peak, "SQ_INSTS_SALU": a.get("SQ_INSTS_SALU")}
            if b:
                gui = b.get("GRBM_GUI_ACTIVE", 0.0)
                e["counters_pass_b"] = b
                # GRBM_GUI_ACTIVE is reported per XCD; the dispatch's sum over its rows / 8 = cycles the launch was active
                e["hw_valu_busy_frac"] = (b.get("SQ_ACTIVE_INST_VALU", 0.0) * 4) / (1024 * gui / 8) if gui else None
                e["hw_dual_issue_frac_of_valu_quads"] = b.get("SQ_ACTIVE_INST_VALU2", 0.0) / max(b.get("SQ_ACTIVE_INST_VALU", 1.0), 1.0)
                e["effective_clock_ghz_profiled"] = gui / 8 / (t_ms * 1e-3) / 1e9 if gui else None
            out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
