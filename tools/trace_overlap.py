"""Concurrency of a rocprofv3 --kernel-trace run: python tools/trace_overlap.py kernel_trace.csv [t_skip_frac]
Prints wall time covered by >= 1 kernel, the average number of kernels resident, idle gaps, and per-kernel totals."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
tot = collections.defaultdict(float)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((a, 1)); ev.append((b, -1))
    tot[r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]] += (b - a)
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3       # ignore the warm-up part of the trace
lo = t0 + (t1 - t0) * skip
cur = 0; last = ev[0][0]; busy = 0; area = 0; hist = collections.Counter(); gaps = []
for t, d in ev:
    if t > lo:
        dt = t - max(last, lo)
        if dt > 0:
            hist[min(cur, 8)] += dt
            if cur > 0: busy += dt
            else: gaps.append(dt)
            area += cur * dt
    cur += d; last = t
wall = t1 - lo
print("window %.1f ms: busy %.1f %%, mean kernels resident %.2f" % (wall / 1e6, 100 * busy / wall, area / wall))
print("resident-count histogram (%% of window):", {k: round(100 * v / wall, 1) for k, v in sorted(hist.items())})
gaps.sort(reverse=True)
print("idle: %d gaps, total %.2f ms, largest %s us" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3) for g in gaps[:8]]))
s = sum(tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-26s %6.1f %%" % (k, 100 * v / s))
