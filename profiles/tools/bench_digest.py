"""One line per section of a bench.py JSON line (value, roofline fraction, training time, CPU baseline, parity)."""
import json
import sys


def show(k, v, ind=0):
    if not isinstance(v, dict):
        return
    if "error" in v:
        print(" " * ind + k, "ERROR", v["error"][:300])
        return
    line = " " * ind + k + ": " + ", ".join(f"{x}={v[x]:.4g}" for x in ("value", "ms_per_call", "ms_per_step")
                                             if isinstance(v.get(x), (int, float)))
    r = v.get("roofline")
    if isinstance(r, dict):
        line += f" frac={r.get('frac', 0):.4g}"
        for x in ("frac_isolated", "pipeline_frac_wall"):
            if x in r:
                line += f" {x}={r[x]:.4g}"
    t = v.get("train_fwd_bwd")
    if isinstance(t, dict):
        line += " train_ms=" + str(t.get("ms", t.get("hip_ms")))
        r = t.get("roofline")
        if isinstance(r, dict):
            if "avg_launch_ms" in r:
                line += f" bwd_frac={r['frac']:.4g} bwd_ms={r['avg_launch_ms']:.4g}"
            else:
                line += f" step_frac={r['frac']:.4g} step_kernel_ms={r['kernel_ms_per_step']:.4g}"
    if "cpu_baseline" in v:
        line += f" cpu={v['cpu_baseline']['value']:.4g} ({v['cpu_baseline']['cores']} cores)"
    if "parity" in v:
        line += " parity=" + json.dumps(v["parity"])[:170]
    print(line)
    for kk, vv in v.items():
        if isinstance(vv, dict) and (kk in ("train", "c2", "c3_fp16_sh", "fast_exp", "cost_volume", "ptf", "encoder_tail", "multi_gpu") or "metric" in vv):
            show(kk, vv, ind + 2)


if __name__ == "__main__":
    for ln in open(sys.argv[1]):
        ln = ln.strip()
        if ln.startswith("{"):
            show("headline", json.loads(ln))
