#!/usr/bin/env python
"""Static VALU instruction mix of the blend kernels (CPU only: hipcc -S --cuda-device-only on the shipped sources with the
Makefile's flags): per kernel the number of wave instructions by class, and per fp32 operation (fma / mul / add) the fraction
that is PACKED (v_pk_*_f32: two values per lane, twice the issue cost -- profiles/tools/valu_issue_rate.hip).  The hardware
counters (SQ_INSTS_VALU_FMA_F32 ...) count a packed instruction once; this fraction is what prices them in
profiles/tools/valu_roofline.py.  Usage: python profiles/tools/isa_mix.py > profiles/r5_isa_static_mix.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "freesplat_amd", "csrc")
KERNELS = {   # demangled-name substring -> (source, mangled-name regex)
    "sort_blend_kernel<false, false>": ("raster_fwd.hip", r"^_ZN2fs17sort_blend_kernelILb0ELb0EE"),
    "sort_blend_kernel<false, true>": ("raster_fwd.hip", r"^_ZN2fs17sort_blend_kernelILb0ELb1EE"),
    "render_bwd_kernel<false, false>": ("raster_bwd.hip", r"^_ZN2fs17render_bwd_kernelILb0ELb0EE"),
    "preprocess_kernel": ("raster_fwd.hip", r"^_ZN2fs17preprocess_kernelE"),
}


def classify(op: str) -> str:
    if op.startswith("v_pk_fma_f32"): return "pk_fma_f32"
    if op.startswith("v_pk_mul_f32"): return "pk_mul_f32"
    if op.startswith("v_pk_add_f32"): return "pk_add_f32"
    if re.match(r"v_(fma|fmac|mad|mac)_f32", op): return "fma_f32"
    if re.match(r"v_mul_f32", op): return "mul_f32"
    if re.match(r"v_(add|sub|subrev)_f32", op): return "add_f32"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", op): return "trans_f32"
    if re.match(r"v_cvt_", op): return "cvt"
    if re.match(r"v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|min|max|mbcnt|bcnt|ffb|alignbit|perm|lshlrev|lshrrev|ashrrev|add3|lshl_add|add_lshl|and_or|or3|xad|med3|addc|subb)_?[a-z]*_?(u|i|b)(16|24|32|64)", op):
        return "int"
    if re.match(r"v_cmpx?_", op): return "cmp"
    if re.match(r"v_cndmask", op): return "cndmask"
    if "_dpp" in op: return "dpp_mov"
    if re.match(r"v_(readlane|readfirstlane|writelane)", op): return "lane"
    if re.match(r"v_mov|v_accvgpr|v_swap", op): return "mov"
    if re.match(r"v_(max|min|med3|fract|floor|ceil|trunc|rndne|ldexp|frexp)", op): return "minmax_misc_f32"
    return "other_valu"


def count(lines):
    c = collections.Counter()
    salu = lds = vmem = 0
    for l in lines:
        t = l.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith((";", ".")):
            continue
        op = t[0]
        is_dpp = any(x.startswith(("row_", "quad_perm", "wave_", "row_bcast")) for x in t[1:])
        if op.startswith("v_") and not op.startswith("v_mfma"):
            c["dpp_mov" if (is_dpp and op.startswith("v_mov")) else classify(op)] += 1
        elif op.startswith("s_"):
            salu += 1
        elif op.startswith("ds_"):
            lds += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vmem += 1
    return c, salu, lds, vmem


def hot_loop(lines, start, end):
    """The blend loop: of the backward-branch spans (label ... s_cbranch label) that hold packed fp32 instructions, the one
    with the most of them, and of those the SHORTEST (the loop itself, not an enclosing one).  The kernels' fp32 arithmetic
    lives there (the per-tile sort of the forward is compares and selects), so its packed fractions price the dynamic
    SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 counts."""
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    best = None
    for i in range(start, end):
        t = lines[i].strip().split()
        if t and (t[0].startswith("s_cbranch") or t[0] == "s_branch") and len(t) > 1 and t[1] in labels and labels[t[1]] < i:
            a = labels[t[1]]
            c, salu, lds, vmem = count(lines[a:i + 1])
            npk = c["pk_fma_f32"] + c["pk_mul_f32"] + c["pk_add_f32"]
            if npk and (best is None or (npk, -(i - a)) > (best[0], -best[1])):
                best = (npk, i - a, c, salu, lds, vmem, a - start, i - start)
    if best is None:
        return None
    npk, span, c, salu, lds, vmem, a, b = best
    frac = lambda x, y: round(c[x] / max(c[x] + c[y], 1), 4)
    return {"asm_lines_from_kernel_start": [a, b], "valu": sum(c.values()), "packed": npk, "salu": salu, "lds": lds, "vmem": vmem,
            "by_class": dict(sorted(c.items(), key=lambda kv: -kv[1])),
            "packed_fraction": {"fma_f32": frac("pk_fma_f32", "fma_f32"), "mul_f32": frac("pk_mul_f32", "mul_f32"),
                                "add_f32": frac("pk_add_f32", "add_f32")}}


def main():
    out = {}
    asm_cache = {}
    for name, (src, rx) in KERNELS.items():
        if src not in asm_cache:
            contract = "on" if src in ("cost_volume.hip", "ptf_gru.hip") else "off"
            with tempfile.NamedTemporaryFile(suffix=".s") as f:
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-ffp-contract={contract}",
                                       "-fno-gpu-rdc", "--cuda-device-only", "-S", "-o", f.name, os.path.join(CSRC, src)],
                                      stderr=subprocess.DEVNULL)
                asm_cache[src] = open(f.name).read().splitlines()
        lines = asm_cache[src]
        start = next(i for i, l in enumerate(lines) if re.match(rx, l))
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        c = collections.Counter()
        salu = lds = vmem = 0
        for l in lines[start:end]:
            t = l.strip().split()
            if not t or t[0].endswith(":") or t[0].startswith((";", ".")):
                continue
            op = t[0]
            is_dpp = any(x.startswith(("row_", "quad_perm", "wave_", "row_bcast")) for x in t[1:])
            if op.startswith("v_") and not op.startswith("v_mfma"):
                c["dpp_mov" if (is_dpp and op.startswith("v_mov")) else classify(op)] += 1
            elif op.startswith("s_"):
                salu += 1
            elif op.startswith("ds_"):
                lds += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                vmem += 1
        tot = sum(c.values())
        hot = hot_loop(lines, start, end)
        frac = lambda a, b: round(c[a] / max(c[a] + c[b], 1), 4)
        out[name] = {"static_valu_instructions": tot, "by_class": dict(sorted(c.items(), key=lambda kv: -kv[1])),
                     "packed_fraction": {"fma_f32": frac("pk_fma_f32", "fma_f32"), "mul_f32": frac("pk_mul_f32", "mul_f32"),
                                         "add_f32": frac("pk_add_f32", "add_f32")},
                     "static_salu": salu, "static_lds": lds, "static_vmem": vmem, "hot_loop": hot}
    json.dump({"what": "static instruction mix of the shipped kernels' gfx950 code (hipcc -S); packed_fraction = v_pk_* share of "
                       "each fp32 operation", "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
