#!/usr/bin/env python3
"""tools/timeline_summary.py <kernel_trace.csv> <frames> [skip_frames] — where a frame's wall time goes, from a rocprofv3
--kernel-trace csv of a frame loop (tools/tile_loop.py, bench.py): over the steady-state part of the trace (the last
`frames` frames; a frame = one dispatch of init_paths_kernel),
  wall per frame            (last kernel end - first kernel start) / frames
  GPU busy per frame        length of the UNION of all kernel intervals / frames  (the rest is idle: launch gaps, host)
  per kernel                launches per frame, summed duration per frame (kernels overlap: the sums exceed busy time),
                            mean duration, and how much of the kernel's time ran ALONE on the GPU
Usage on the GPU box: tools/prof_tile.sh writes the csv and calls this."""
import collections
import csv
import sys


def short(name):
    name = name[:name.index("(")] if "(" in name else name
    name = name.replace("void ", "").replace("bdpt::", "")
    return name[:name.index("<")] if "<" in name else name


def main():
    path, frames = sys.argv[1], int(sys.argv[2])
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [s for s, e, k in rows if k == "init_paths_kernel"]
    if len(starts) <= frames:
        raise SystemExit("trace holds %d frames, asked for the last %d" % (len(starts), frames))
    t0 = starts[-frames - 1]   # steady state: from the start of frame (last - frames) ...
    t1 = starts[-1]            # ... to the start of the last frame: exactly `frames` frame periods
    sel = [(max(s, t0), min(e, t1), k) for s, e, k in rows if e > t0 and s < t1]
    wall = (t1 - t0) / 1e6
    # union of intervals, and time during which exactly one kernel runs (attributed to it)
    ev = []
    for s, e, k in sel:
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = collections.Counter()
    busy = 0
    alone = collections.Counter()
    last = t0
    for t, d, k in ev:
        n = sum(active.values())
        if n > 0:
            busy += t - last
        if n == 1:
            alone[[kk for kk, c in active.items() if c > 0][0]] += t - last
        last = t
        active[k] += d
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, k in sel:
        per[k][0] += 1
        per[k][1] += e - s
    print("steady state: %d frames, wall %.3f ms per frame, GPU busy (union of kernels) %.3f ms per frame = %.1f %%, idle %.3f ms per frame" % (
        frames, wall / frames, busy / 1e6 / frames, 100.0 * busy / 1e6 / wall, (wall - busy / 1e6) / frames))
    print("%-28s %9s %12s %10s %14s" % ("kernel", "launches", "sum ms/frame", "mean us", "alone ms/frame"))
    tot = 0.0
    for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        tot += d / 1e6 / frames
        print("%-28s %9.2f %12.3f %10.1f %14.3f" % (k[:28], n / frames, d / 1e6 / frames, d / 1e3 / n, alone[k] / 1e6 / frames))
    print("%-28s %9s %12.3f   (kernels of %s frames in flight overlap: the sum exceeds the busy time)" % ("sum of kernel durations", "", tot, "several"))


if __name__ == "__main__":
    main()
