"""Every kernel of one replayed step (rocprofv3 kernel trace csv): start, duration, queue, grid, name."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mxdet::", "")
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:58], r["Queue_Id"], grid // max(wg, 1)))
rows.sort()
st = [i for i, r in enumerate(rows) if "stem_conv" in r[2] or "stem_pool" in r[2]]
lo, hi = st[-3], st[-2]
t0 = rows[lo][0]
print("step wall %.1f us" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3))
for s, e, n, q, g in rows[lo:hi]:
    print("%8.1f %7.1f q%s %6d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, n))
