"""Join the rocprofv3 passes of tools/micro/wgrad_layers.sh into a per-layer table. usage: <dir>"""
import csv, glob, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1]
import importlib.util
spec = importlib.util.spec_from_file_location("wl", os.path.join(os.path.dirname(os.path.abspath(__file__)), "wgrad_layers.py"))
src = open(spec.origin).read()
ns = {}
exec(src[src.index("LAYERS = ["):src.index("if sys.argv[1]")], ns)
LAYERS, min_bytes = ns["LAYERS"], ns["min_bytes"]
per = collections.defaultdict(dict)      # dispatch id -> counters
names = {}
for f in sorted(glob.glob(os.path.join(root, "pass*", "c_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(os.path.join(root, "trace", "c_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = [i for i in sorted(names) if "wgrad" in names[i]]
# group dispatches per layer: a layer = one tile kernel (+ optionally one reduce kernel behind it)
groups = []
for i in ids:
    if "reduce" in names[i] and groups:
        groups[-1].append(i)
    else:
        groups.append([i])
n = len(LAYERS)
last = groups[-n:]
tids = [i for i in sorted(dur) if i in dur]
print("%-14s %3s | %8s %8s %8s  %5s | %8s %7s" % ("layer", "x", "min MB", "read MB", "write MB", "ratio", "us", "TF"))
tot = [0.0, 0.0, 0.0]
for (nm, H, W, Cin, Cout, k, s, c), g in zip(LAYERS, last):
    mb, fl = min_bytes(H, W, Cin, Cout, k, s)
    rd = sum(per[i].get("FETCH_SIZE", 0.0) for i in g) * 1024 * 2
    wr = sum(per[i].get("WRITE_SIZE", 0.0) for i in g) * 1024
    # durations come from a separate trace pass with the same dispatch order
    us = sum(dur.get(i, 0.0) for i in g)
    hit = sum(per[i].get("TCC_HIT_sum", 0.0) for i in g); miss = sum(per[i].get("TCC_MISS_sum", 0.0) for i in g)
    print("%-14s %3d | %8.1f %8.1f %8.1f  %5.2f | %8.1f %7.1f  L2 hit %.2f  %s" % (
        nm, c, mb / 1e6, rd / 1e6, wr / 1e6, (rd + wr) / mb, us, fl / us / 1e6 if us else 0.0,
        hit / (hit + miss) if hit + miss else 0.0, names[g[0]].split("(")[0][-28:]))
    tot[0] += mb * c; tot[1] += (rd + wr) * c; tot[2] += us * c
print("per step (x counts): min %.2f GB, measured %.2f GB, %.1f us" % (tot[0] / 1e9, tot[1] / 1e9, tot[2]))
