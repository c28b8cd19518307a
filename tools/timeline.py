"""Analyse a rocprofv3 --kernel-trace CSV of `bench.py`: take the last replayed step, report busy time per HW
queue, the union busy time, bubbles, and the kernels on the longest gaps.  usage: timeline.py <kernel_trace.csv>"""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                     r.get("Queue_Id", "0")))
rows.sort()
# a step starts with the stem kernel; take the window between the third- and second-last launches of it
sgd = [i for i, r in enumerate(rows) if "stem_conv_kernel" in r[2] or "stem_pool_kernel" in r[2]]
if len(sgd) < 3:
    sys.exit("need >= 3 steps in the trace")
lo, hi = sgd[-3], sgd[-2]
win = rows[lo:hi]
# the window runs from one launch of the stem kernel to the next: exactly one step period. (With the front-end pipeline
# the stem of step k+1 starts inside step k's weight-gradient tail, so the window holds the tail of one step and the
# body of the next; kernels that start inside it may end after it.)
t0, t1 = win[0][0], rows[hi][0]
print("kernels in step: %d   step period %.3f ms   (last kernel of the window ends at %.3f ms)" % (
    len(win), (t1 - t0) / 1e6, (max(r[1] for r in win) - t0) / 1e6))
per_q = collections.defaultdict(float)
for s, e, n, q, st in win:
    per_q[(q, st)] += (e - s) / 1e6
for k, v in sorted(per_q.items()):
    print("  queue/stream %s: busy %.3f ms" % (k, v))
# union busy
ev = sorted([(s, 1) for s, e, *_ in win] + [(min(e, t1), -1) for s, e, *_ in win])
busy = 0; depth = 0; last = None; gaps = []
for t, d in ev:
    if depth > 0:
        busy += t - last
    elif last is not None and t - last > 0:
        gaps.append((t - last, last))
    depth += d; last = t
print("union busy %.3f ms, idle %.3f ms in %d gaps" % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
gaps.sort(reverse=True)
for g, at in gaps[:12]:
    prev = max((r for r in win if r[1] <= at), key=lambda r: r[1])
    nxt = min((r for r in win if r[0] >= at + g), key=lambda r: r[0])
    print("  gap %.1f us after %-40.40s before %-40.40s" % (g / 1e3, prev[2], nxt[2]))
# coarse phase timeline: 0.25 ms bins, top kernel per bin per stream
print("timeline (0.5 ms bins): bin | per-stream busy fraction | top kernel")
nb = int((t1 - t0) / 5e5) + 1
for b in range(nb):
    a, z = t0 + b * 5e5, t0 + (b + 1) * 5e5
    acc = collections.defaultdict(float); top = collections.defaultdict(float)
    for s, e, n, q, st in win:
        o = min(e, z) - max(s, a)
        if o > 0:
            acc[st] += o; top[n.split("(")[0][:48]] += o
    tk = sorted(top.items(), key=lambda kv: -kv[1])[:2]
    print("  %4.1f ms | %s | %s" % (b * 0.5, " ".join("%s:%.2f" % (k, v / 5e5) for k, v in sorted(acc.items())),
                                 ", ".join("%s %.0fus" % (k, v / 1e3) for k, v in tk)))
print("per-queue kernel totals (us):")
tot = collections.defaultdict(lambda: [0.0, 0])
for s, e, n, q, st in win:
    k = (q, n.split("(")[0].replace("void ", "").replace("mxdet::", "")[:46])
    tot[k][0] += (e - s) / 1e3; tot[k][1] += 1
for (q, n), (us, c) in sorted(tot.items(), key=lambda kv: (kv[0][0], -kv[1][0])):
    if us >= 15:
        print("  q%s %-46s %8.1f  x%d" % (q, n, us, c))
