#!/usr/bin/env python3
"""Load-balance analysis of one march launch from the per-workgroup trace (vr_last_block_trace):

    python tools/block_trace.py [bench.py scene args: --workload C3 --tf thin --flavour N] > gpurun_out/trace.txt

Prints, per XCD and for the whole device: the span (first start .. last end), the summed workgroup time, the
time-averaged number of resident workgroups, and how the workgroup durations are distributed."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--tf", default="default")
    ap.add_argument("--air", default="exact0", choices=["exact0", "noisy"])
    ap.add_argument("--flavour", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1, help="trace rank's share of the tiles (tile t -> rank t mod world)")
    a = ap.parse_args()
    from volumerendering_amd import capi, host, workloads as wl
    n, W, H, vname = wl.WORKLOADS[a.workload]
    app = host.Application(W, H, 0)
    variant, vols = wl.build_scene(app, a.workload, a.tf, a.air)
    ctx = app.context()
    ctx.set_kernel_flavour(a.flavour)
    for _ in range(6):   # the launch order is three launches old: let it settle
        if a.world > 1:
            ctx.render_tiles(variant, a.rank, a.world)
        else:
            app.OnRender()
    tr = ctx.block_trace().astype(np.int64)
    t0, t1 = tr[:, 3], tr[:, 4]
    xcc = (tr[:, 5] >> 32) & 0xF
    crit = tr[:, 5] >> 40
    hw = tr[:, 5] & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    base = t0.min()
    span = (t1.max() - base) / 100.0  # us
    dur = (t1 - t0) / 100.0
    print(f"workgroups {len(tr)}  device span {span:.1f} us  sum of workgroup time {dur.sum():.0f} us  "
          f"mean resident workgroups {dur.sum() / span:.1f}")
    print(f"duration us: median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}")
    heavy = tr[:, 2] > 0
    print(f"workgroups that fetched samples: {heavy.sum()}  their mean duration {dur[heavy].mean():.1f} us, "
          f"mean fetched {tr[heavy, 2].mean():.0f}, ns per fetched sample-lane {1e3 * dur[heavy].sum() / tr[heavy, 2].sum():.3f}")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  XCD {x}: workgroups {m.sum():5d}  first start {(t0[m].min() - base) / 100.0:7.1f}  last end {(t1[m].max() - base) / 100.0:7.1f} us  "
              f"busy {dur[m].sum():9.0f} us  fetched {tr[m, 2].sum():10d}  composited {tr[m, 0].sum():11d}  CUs seen {len(set(zip(se[m].tolist(), cu[m].tolist())))}")
    order = np.argsort(-dur)[:12]
    print("longest workgroups: duration us / start us / fetched / composited / XCD / tile ordinal")
    per_tile = len(tr) // max(1, ctx.tile_count(a.rank, a.world) if a.world > 1 else ((W + 63) // 64) * ((H + 63) // 64))
    for i in order:   # records are indexed by LOGICAL block: consecutive blocks of a tile
        print(f"   {dur[i]:7.1f} {(t0[i] - base) / 100.0:7.1f} {tr[i, 2]:8d} {tr[i, 0]:8d}   {xcc[i]}  {i // max(1, per_tile)}  chain {crit[i]}")
    # how well does the start order follow the chain lengths?  (longest-first launch order)
    rank_start = np.argsort(np.argsort(t0))
    rank_chain = np.argsort(np.argsort(-crit))
    print(f"rank correlation of start time with descending chain length: {np.corrcoef(rank_start, rank_chain)[0, 1]:.3f}")
    print(f"longest sample chain {crit.max()}  p99 {np.percentile(crit[heavy], 99):.0f}  median of busy workgroups {np.median(crit[heavy]):.0f}")
    # what a packet's time is made of: duration against its longest sample chain, for packets that started while the machine
    # was filling (first 5 % of the span: they run beside a full machine for most of their life)
    early = heavy & ((t0 - base) / 100.0 < 0.05 * span)
    if early.sum() > 50:
        c, d = crit[early].astype(np.float64), dur[early]
        A = np.stack([c, np.ones_like(c)], 1)
        (slope, icpt), *_ = np.linalg.lstsq(A, d, rcond=None)
        print(f"packets started in the first 5 % that fetched: {early.sum()}  duration ~ {icpt:.1f} us + {slope:.3f} us x chain")
        qs = np.quantile(c, [0, 0.2, 0.4, 0.6, 0.8, 1.0])
        for lo, hi in zip(qs[:-1], qs[1:]):
            m = (c >= lo) & (c <= hi)
            steps = tr[early, 0][m] / np.maximum(1, tr[early, 1][m])  # composited per covered pixel = steps a ray takes in the box
            print(f"   chain {lo:4.0f}..{hi:4.0f}: n {m.sum():5d}  median duration {np.median(d[m]):6.1f} us  median steps in box per ray {np.median(steps):6.1f}"
                  f"  us per chain step {np.median(d[m] / np.maximum(1, c[m])):.3f}")
    if os.environ.get("VR_P2_DEBUG"):  # (a -DVR_P2_DEBUG=1 build: the fetched word holds march_p2_kernel's loop counters)
        f = tr[:, 2]
        trips, smp, shd, jmp = f & 0xfff, (f >> 12) & 0xfff, (f >> 24) & 0xfff, (f >> 36) & 0xfff
        busy = smp > 0
        for name, sel in (("all sampling packets", busy), ("duration > 350 us", busy & (dur > 350)), ("duration 200..350 us", busy & (dur > 200) & (dur <= 350))):
            if sel.sum() == 0:
                continue
            print(f"[loop counters] {name}: n {sel.sum()}  median duration {np.median(dur[sel]):.0f} us  trips {np.median(trips[sel]):.0f}  sampled steps {np.median(smp[sel]):.0f}"
                  f"  shaded steps {np.median(shd[sel]):.0f}  jumps {np.median(jmp[sel]):.0f}  chain {np.median(crit[sel]):.0f}"
                  f"  us per trip {np.median(dur[sel] / np.maximum(1, trips[sel])):.2f}")
        # what a step slot costs by kind: least squares over the sampling packets that start while the machine is full
        # (duration = a x shaded slots + b x sampled-but-not-shaded + c x idle slots + d x jumps + e)
        idle = 2 * trips.astype(np.float64) - smp
        st = (t0 - base) / 100.0
        mid = busy & (st > 0.05 * span) & (st < 0.6 * span) & (trips < 0xfff)
        if mid.sum() > 50:
            A = np.stack([shd[mid], (smp - shd)[mid], idle[mid], jmp[mid], np.ones(mid.sum())], axis=1).astype(np.float64)
            coef, *_ = np.linalg.lstsq(A, dur[mid].astype(np.float64), rcond=None)
            print(f"[slot costs] {mid.sum()} packets started at 5-60 % of the span: us per shaded slot {coef[0]:.3f}, sampled not shaded {coef[1]:.3f}, "
                  f"idle slot {coef[2]:.3f}, jump {coef[3]:.3f}, per packet {coef[4]:.1f}")
        airp = (trips > 0) & ~busy
        print(f"[packet time] packets with sampled slots {busy.sum()}: {dur[busy].sum() / 1e3:.1f} ms of wavefront time; in the pipelined loop without one {airp.sum()}: "
              f"{dur[airp].sum() / 1e3:.1f} ms (trips {trips[airp].sum()}, jumps {jmp[airp].sum()}); the rest {(~busy & ~airp).sum()}: {dur[~busy & ~airp].sum() / 1e3:.1f} ms")
        print(f"[slot totals] sampling packets {busy.sum()}: trips {trips[busy].sum()}  slots {2 * trips[busy].sum()}  sampled {smp[busy].sum()}  shaded {shd[busy].sum()}  "
              f"idle {int(idle[busy].sum())} ({100 * idle[busy].sum() / max(1, 2 * trips[busy].sum()):.1f} %)  jumps {jmp[busy].sum()}")
    if os.environ.get("VR_P2_DEBUG") == "2":  # (-DVR_P2_DEBUG=2: the covered word holds wait cycles / 64)
        c = tr[:, 1]
        wc, wb, lp = (c & 0xfffff) * 64.0, ((c >> 20) & 0xfffff) * 64.0, ((c >> 40) & 0xffffff) * 64.0
        busy = lp > 0
        print(f"[wait cycles] packets {busy.sum()}: pipelined loop {lp[busy].sum() / 1e6:.1f} Mcycles, waiting for corners {wc[busy].sum() / 1e6:.1f} "
              f"({100 * wc[busy].sum() / lp[busy].sum():.1f} %), for the distance-field bytes {wb[busy].sum() / 1e6:.1f} ({100 * wb[busy].sum() / lp[busy].sum():.1f} %); "
              f"median per packet: loop {np.median(lp[busy]):.0f} corners {np.median(wc[busy]):.0f} bytes {np.median(wb[busy]):.0f} cycles")
    if os.environ.get("VR_P2_DEBUG") == "3":  # (-DVR_P2_DEBUG=3: tail-loop cycles, whole-packet cycles, pipelined-loop cycles, / 64)
        c = tr[:, 1]
        tl, pk, lp = (c & 0xfffff) * 64.0, ((c >> 20) & 0xfffff) * 64.0, ((c >> 40) & 0xffffff) * 64.0
        busy = lp > 0
        print(f"[packet cycles] all packets {pk.sum() / 1e6:.1f} Mcycles; sampling packets ({busy.sum()}): whole {pk[busy].sum() / 1e6:.1f}, pipelined loop "
              f"{lp[busy].sum() / 1e6:.1f} ({100 * lp[busy].sum() / pk[busy].sum():.1f} %), last steps (plain loop) {tl[busy].sum() / 1e6:.1f} "
              f"({100 * tl[busy].sum() / pk[busy].sum():.1f} %), set-up and dequeue the rest; packets that never enter the pipelined loop: "
              f"{(~busy).sum()} with {pk[~busy].sum() / 1e6:.1f} Mcycles")
        long_ = busy & (pk > np.percentile(pk[busy], 75))
        print(f"[packet cycles] longest quarter of the sampling packets: pipelined {100 * lp[long_].sum() / pk[long_].sum():.1f} %, last steps {100 * tl[long_].sum() / pk[long_].sum():.1f} %")
    # residency over time (20 bins)
    edges = np.linspace(0, span, 21)
    res = []
    for i in range(20):
        lo, hi = edges[i], edges[i + 1]
        ov = np.clip(np.minimum((t1 - base) / 100.0, hi) - np.maximum((t0 - base) / 100.0, lo), 0, None)
        res.append(ov.sum() / (hi - lo))
    print("resident workgroups per 5 % of the span:", " ".join(f"{r:.0f}" for r in res))


if __name__ == "__main__":
    main()
