#!/usr/bin/env python3
"""Prints the rows of DESIGN.md section 5's table from profiles/r02_*.json (run tools/collect_profiles.py first)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return json.load(open(os.path.join(ROOT, "profiles", f"r02_{name}.json")))


def row(label, tf, name):
    d = load(name)
    s, o, p = d["serial"], d["overlapped"], d.get("pipelined_one_frame_per_launch") or {}
    c = d["config"]
    return (f"| {label} | {tf} | {c['composited_samples_per_frame'] / 1e6:.1f} M / {c['fetched_samples_per_frame'] / 1e6:.1f} M | "
            f"{s['kernel_ms_median']:.3f}, {s['ms_per_step']:.3f}, {s['value']:.0f} | {p.get('ms_per_step', 0):.3f}, {p.get('value', 0):.0f} | "
            f"{o['ms_per_step']:.3f}, {o['fps']:.0f}, {o['value']:.0f} | {c.get('kernel_flavour_resolved')} |")


for label, tf, name in [("C1", "default", "C1_default"), ("C2", "default", "C2_default"), ("C2", "thin", "C2_thin"),
                        ("C3", "default", "bench_default"), ("C3 fused", "default", "c3_fused"), ("C3", "thin", "c3_thin"),
                        ("C4", "default", "C4_default"), ("C4", "thin", "C4_thin"), ("C5", "default", "C5_default"), ("C5", "thin", "C5_thin"),
                        ("C3 noisy", "default", "c3_noisy"), ("C3 flavour 1", "default", "c3_flavour1"), ("C3 no order", "default", "c3_noorder"),
                        ("C3 wpb4", "default", "c3_wpb4"), ("C3 otf", "default", "c3_otf")]:
    try:
        print(row(label, tf, name))
    except Exception as e:  # noqa: BLE001
        print(label, name, "missing", e)
for n in (2, 4, 8):
    for suffix in ("", "_k20"):
        try:
            d = load(f"c3_share{n}{suffix}")
            p = d.get("pipelined_one_frame_per_launch") or {}
            print(f"share {n}{suffix}: serial {d['serial']['ms_per_step']:.3f}  2x1 {p.get('ms_per_step', 0):.3f}  2x4 {d['overlapped']['ms_per_step']:.4f}")
        except Exception as e:  # noqa: BLE001
            print("share", n, suffix, "missing", e)
d = load("c3_selfgather")
print("selfgather", d["serial"]["ms_per_step"], (d.get("pipelined_one_frame_per_launch") or {}).get("ms_per_step"), d["overlapped"]["ms_per_step"])
d = load("c3_default")
r = d["roofline"]
print("roofline c3_default: kernel", r["kernel_ms"], "frac", r["frac"], "frac_over", r["frac_overlapped"], "traffic GB", r["traffic"] / 1e9,
      "valu", r["valu"]["busy_frac"], r["valu"]["busy_frac_overlapped"], "insts", r["valu"]["insts_per_launch"] / 1e6,
      "l1", r["l1"]["frac"], r["l1"]["frac_overlapped"], "ta", r["l1"]["ta_busy_frac"], r["l1"]["ta_busy_frac_overlapped"],
      "l1 bytes", r["l1"]["bytes_per_launch"] / 1e9, "gather", r["effective_gather"]["fetched_gbs"], "l2hit", r.get("l2_hit_rate"),
      "serial ms", d["serial"]["ms_per_step"], "over", d["overlapped"]["ms_per_step"])
if "regimes" in load("bench_default"):
    for g in load("bench_default")["regimes"]:
        print(g)
b = load("bench_default")
print("cpu", b.get("cpu_baseline"))
print("parity", {k: v for k, v in (b.get("parity") or {}).items() if not isinstance(v, (list, dict))})
