#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / average / total, and -- for kernels that are launched several
times per training step with different work (the lazy K2's rounds) -- the average per position inside the step.
usage: kernel_trace_summary.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("ngp::", "").replace("void ", "")
    return name[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if len(sys.argv) > 2:
        rows = rows[int(sys.argv[2]):]
    by = defaultdict(list)
    for s, e, n in rows:
        by[n].append((e - s) / 1e3)
    total = sum(sum(v) for v in by.values())
    print(f"{len(rows)} dispatches, {total / 1e3:.2f} ms of kernel time")
    print(f"{'kernel':70s} {'count':>8s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:70s} {len(v):8d} {sum(v) / len(v):9.2f} {sum(v) / 1e3:9.2f} {100 * sum(v) / total:6.2f}")
    # timeline of the last 60 % of the trace (steady state): union of busy intervals vs wall time, overlap, gaps between consecutive kernels
    tail = rows[int(0.4 * len(rows)):]
    if tail:
        t0, t1 = tail[0][0], max(e for _, e, _ in tail)
        busy, cur_s, cur_e, ksum, gaps = 0, tail[0][0], tail[0][1], 0, []
        for s, e, n in tail:
            ksum += e - s
            if s > cur_e:
                busy += cur_e - cur_s; gaps.append((s - cur_e) / 1e3); cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        n_opt = sum(1 for _, _, n in tail if "k_optimizer" in n)
        gaps.sort()
        print(f"steady-state timeline: wall {(t1 - t0) / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.1f} %), sum of kernel durations {ksum / 1e6:.2f} ms, "
              f"{len(gaps)} idle gaps: total {sum(gaps) / 1e3:.2f} ms, median {gaps[len(gaps) // 2] if gaps else 0:.2f} us, p90 {gaps[int(0.9 * len(gaps))] if gaps else 0:.2f} us; "
              f"{n_opt} optimizer steps -> {(t1 - t0) / 1e3 / max(n_opt, 1):.1f} us wall, {busy / 1e3 / max(n_opt, 1):.1f} us busy, {ksum / 1e3 / max(n_opt, 1):.1f} us kernel time per step")
    # average step timeline: a step = (end of the previous k_optimizer, end of this k_optimizer]; kernels are placed by their START; steps that
    # contain occupancy-grid kernels are left out.  Offsets are relative to the previous optimizer's end.
    opt_ends = [e for _, e, n in tail if "k_optimizer" in n]
    if len(opt_ends) > 10:
        import bisect
        steps = defaultdict(list)
        for s, e, n in tail:
            k = bisect.bisect_left(opt_ends, e)  # this kernel ends inside step k (ends at or before opt_ends[k])
            if 0 < k < len(opt_ends):
                steps[k].append((s - opt_ends[k - 1], e - opt_ends[k - 1], n))
        clean = [v for v in steps.values() if not any("grid" in n or "bitfield" in n or "density_only" in n or "true, 1, false" in n for _, _, n in v)]
        if clean:
            agg = defaultdict(lambda: [0.0, 0.0, 0])
            for v in clean:
                seen = defaultdict(int)
                for s, e, n in v:
                    seen[n] += 1
                    key = f"{n}#{seen[n]}" if seen[n] > 1 else n
                    a = agg[key]; a[0] += s / 1e3; a[1] += e / 1e3; a[2] += 1
            print(f"average step timeline over {len(clean)} steps without occupancy-grid update (us after the previous optimizer's end): start .. end")
            for key, a in sorted(agg.items(), key=lambda kv: kv[1][0] / kv[1][2]):
                if a[2] >= len(clean) // 2:
                    print(f"  {key:60s} {a[0] / a[2]:8.1f} .. {a[1] / a[2]:8.1f}   ({(a[1] - a[0]) / a[2]:6.1f} us)")
    # K2 rounds: consecutive runs of the same tiles kernel = one step's rounds
    for key in [k for k in by if "k_inference_tiles" in k]:
        runs, cur = [], []
        for s, e, n in rows:
            if n == key:
                cur.append((e - s) / 1e3)
            elif cur:
                runs.append(cur); cur = []
        if cur:
            runs.append(cur)
        runs = [r for r in runs if len(r) == max(len(x) for x in runs)]
        if runs:
            k = len(runs[0])
            print(f"{key}: {len(runs)} steps x {k} rounds; average us per round: " + ", ".join(f"{sum(r[i] for r in runs) / len(runs):.1f}" for i in range(k)))


if __name__ == "__main__":
    main()
