#!/usr/bin/env python3
"""Prints the rows of DESIGN.md section 5 / 6 from profiles/rNN_*.json (run tools/collect_profiles.py NN first):
    python tools/design_table.py 04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "04"


def load(name):
    return json.load(open(os.path.join(ROOT, "profiles", f"r{RND}_{name}.json")))


def kept(leg):
    k = (leg or {}).get("kernel_choice") or {}
    return k.get("kept") or k.get("ran_last") or "-"


def row(label, name):
    d = load(name)
    s, o, p = d["serial"], d["overlapped"], d.get("pipelined_one_frame_per_launch") or {}
    c, r = d["config"], d["roofline"]
    pp = s.get("kernel_ms_p10_p90") or [0, 0]
    tr = (r.get("traffic") or 0) / 1e9
    return (f"| {label} | {c['composited_samples_frame0'] / 1e6:.1f} M / {c['fetched_samples_frame0'] / 1e6:.1f} M | "
            f"{s['kernel_ms_median']:.3f} ({pp[0]:.3f}–{pp[1]:.3f}), {s['ms_per_step']:.3f}, **{s['value']:.0f}** | {p.get('ms_per_step', 0):.3f}, {p.get('value', 0):.0f} | "
            f"{o['ms_per_step']:.3f}, {o['value']:.0f} | {tr:.2f} | {kept(s)} / {kept(p)} / {kept(o)} |" if tr else
            f"| {label} | {c['composited_samples_frame0'] / 1e6:.1f} M / {c['fetched_samples_frame0'] / 1e6:.1f} M | "
            f"{s['kernel_ms_median']:.3f} ({pp[0]:.3f}–{pp[1]:.3f}), {s['ms_per_step']:.3f}, **{s['value']:.0f}** | {p.get('ms_per_step', 0):.3f}, {p.get('value', 0):.0f} | "
            f"{o['ms_per_step']:.3f}, {o['value']:.0f} | – | {kept(s)} / {kept(p)} / {kept(o)} |")


print("| config | frame 0: composited / fetched | one frame at a time: kernel ms (p10–p90), ms / frame, **Gsamples/s** | 2 launches × 1 frame: ms, Gs/s | 2 × 4 frames: ms, Gs/s | fabric GB / frame | kernel kept (serial / pipelined / batched) |")
print("|---|---|---|---|---|---|---|")
for label, name in [("C1 sphere-64, 256², unlit", "C1_default"), ("C2 phantom-256, 1024², unlit", "C2_default"), ("C2, thin table", "C2_thin"),
                    ("**C3 phantom-512, 1080p, lit — the driver's command** (`--steps 20 --warmup 5`)", "bench_default"),
                    ("C3, 40 frames (yaw 0.6 → 1.08)", "c3_default"), ("C3, `march_kernel` forced (`--flavour 6`)", "c3_f6"),
                    ("C3, two steps ahead forced in every leg (`--flavour 17`)", "c3_f17"), ("C3, fused arithmetic", "c3_fused"),
                    ("C3, thin table", "c3_thin"), ("C3, noisy air (nothing to skip)", "c3_noisy"), ("C3, noisy air, `march_kernel`", "c3_noisy_f6"),
                    ("**C4** C3 + mask + dose (3 volumes)", "C4_default"), ("C4, `march_kernel` forced", "C4_f6"), ("C4, thin", "C4_thin"),
                    ("**C5** phantom-1024 (16 GiB), 4K, lit, 1 GPU", "C5_default"), ("C5, two steps ahead forced (the moving window)", "C5_f17"), ("C5, thin", "C5_thin")]:
    try:
        print(row(label, name))
    except Exception as e:  # noqa: BLE001
        print(label, name, "missing", e)
print()
for w in ("c3", "C5"):
    for n in (2, 4, 8):
        try:
            d = load(f"{w}_share{n}")
            p = d.get("pipelined_one_frame_per_launch") or {}
            o = d["overlapped"]
            print(f"{w} share {n}: serial {d['serial']['ms_per_step']:.3f} (kernel {d['serial']['kernel_ms_median']:.3f}, {kept(d['serial'])})  2x1 {p.get('ms_per_step', 0):.3f} ({kept(p)})  "
                  f"2x4 {o['ms_per_step']:.4f} ({kept(o)})")
        except Exception as e:  # noqa: BLE001
            print("share", w, n, "missing", e)
d = load("c3_selfgather")
print("selfgather", d["serial"]["ms_per_step"], (d.get("pipelined_one_frame_per_launch") or {}).get("ms_per_step"), d["overlapped"]["ms_per_step"])
for name in ("bench_default", "c3_default", "c3_f6", "c3_noisy", "C4_default", "C5_default"):
    d = load(name)
    r = d["roofline"]
    v, l1 = r.get("valu") or {}, r.get("l1") or {}
    print(f"roofline {name}: kernel_ms {r['kernel_ms']} frac {r['frac']} achieved {r['achieved']} traffic GB {(r.get('traffic') or 0) / 1e9:.3f} frac_over {r.get('frac_overlapped')} "
          f"d2d {r.get('d2d_copy_gbs')} frac_d2d {r.get('frac_of_d2d_copy')} valu busy {v.get('busy_frac')} insts {(v.get('insts_per_launch') or 0) / 1e6:.1f} M "
          f"ta {l1.get('ta_busy_frac')} l1frac {l1.get('frac')} l2hit {r.get('l2_hit_rate')} gather {r['effective_gather']['fetched_gbs']} / {r['effective_gather']['composited_gbs']} clock {v.get('clock_ghz')}")
b = load("bench_default")
for k in ("sync_8d", "full_turn", "serial_with_present"):
    print(k, {kk: vv for kk, vv in (b.get(k) or {}).items() if kk != "note"})
for g in b.get("regimes", []):
    print(g)
print("cpu", {k: v for k, v in (b.get("cpu_baseline") or {}).items() if k not in ("cores_note", "speedup_note", "implementation")})
print("parity", {k: v for k, v in (b.get("parity") or {}).items() if not isinstance(v, (list, dict))})
print("arith_ab", b.get("arith_ab", {}).get("serial"), b.get("arith_ab", {}).get("overlapped"))
for name in ("c3_share8", "C5_share8"):
    d = load(name)
    print(name, "timeline", json.dumps(d.get("rank0_stage_timeline"))[:900])
