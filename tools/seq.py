import csv,sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ","").replace("mxdet::","")[:44], r["Queue_Id"]))
rows.sort()
st=[i for i,r in enumerate(rows) if "stem_conv" in r[2]]
lo,hi=st[-3],st[-2]
t0=rows[lo][0]
print("step wall %.1f us"%((max(r[1] for r in rows[lo:hi])-t0)/1e3))
for s,e,n,q in rows[lo:hi]:
    if any(k in n for k in ("proposal_gather","anchor_iou","roi_align_bwd","nms_scan","f32_to_bf16")) :
        print("%8.1f %7.1f q%s %s"%((s-t0)/1e3,(e-s)/1e3,q,n))
