#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace (…_kernel_trace.csv): per window the fraction of wall-clock with at least one kernel
running (busy union), the mean number of kernels in flight, the launches of the dominant kernel -- and, for the steadiest windows, the idle gaps
and what the time with exactly ONE kernel in flight is spent on (a kernel that runs alone on the chip bounds the step whatever the lanes do).

    python tools/timeline_busy.py <kernel_trace.csv> [--window-ms 20] [--json out.json]"""
import argparse
import collections
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--window-ms", type=float, default=20.0)
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    W = int(a.window_ms * 1e6)
    nwin = (t1 - t0) // W + 1
    # sweep: events
    ev = []
    for s, e, k in rows:
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort(key=lambda x: (x[0], x[1]))
    busy = [0] * nwin          # ns with >= 1 kernel
    area = [0] * nwin          # kernel-ns
    alone = [collections.Counter() for _ in range(nwin)]      # ns with exactly one kernel in flight, by that kernel
    live = collections.Counter()
    n = 0
    prev = t0

    def add(lo, hi, n, live):
        while lo < hi:
            w = (lo - t0) // W
            end = min(hi, t0 + (w + 1) * W)
            if n >= 1:
                busy[w] += end - lo
                area[w] += n * (end - lo)
            if n == 1:
                k = next(k for k, c in live.items() if c > 0)
                alone[w][k] += end - lo
            lo = end

    for t, d, k in ev:
        if t > prev:
            add(prev, t, n, live)
            prev = t
        n += d
        live[k] += d
    roll = [0] * nwin
    for s, e, k in rows:
        if "creff_roll" in k or "creff_mfma" in k:
            roll[(s - t0) // W] += 1
    out = []
    for w in range(nwin):
        out.append({"t_ms": w * a.window_ms, "busy": busy[w] / W, "in_flight": area[w] / W, "creff_launches": roll[w]})
    mx = max(roll)
    steady = [w for w in range(nwin) if roll[w] >= 0.8 * mx] if mx else list(range(nwin))
    sb = sum(busy[w] for w in steady) / (len(steady) * W)
    sa = sum(area[w] for w in steady) / (len(steady) * W)
    al = collections.Counter()
    for w in steady:
        al.update(alone[w])
    tot_alone = sum(al.values()) / (len(steady) * W)
    print(f"{len(rows)} kernels, {nwin} windows of {a.window_ms} ms; {len(steady)} steady windows (>= 0.8 x the most CReFF launches per window):  busy {sb:.3f}  in flight {sa:.2f}  "
          f"exactly one kernel in flight {tot_alone:.3f} of the time")
    for k, v in al.most_common(12):
        print(f"   alone {v / (len(steady) * W):.3f}  {k}")
    for o in out:
        if o["creff_launches"]:
            print(f"  t={o['t_ms']:7.0f} ms  busy {o['busy']:.3f}  in flight {o['in_flight']:.2f}  creff launches {o['creff_launches']}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"note": __doc__, "windows": out, "steady": {"busy": sb, "in_flight": sa, "one_kernel_in_flight": tot_alone,
                                                                   "alone_by_kernel": {k: v / (len(steady) * W) for k, v in al.most_common(20)}}}, f, indent=1)


if __name__ == "__main__":
    main()
