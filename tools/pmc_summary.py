"""Fold the rocprofv3 --pmc passes of tools/pmc_step.sh into per-kernel-family counters of ONE training step.

usage: pmc_summary.py <dir with pass*/c_counter_collection.csv> <out.json>

Per family: launches per step, HBM bytes (FETCH_SIZE and WRITE_SIZE are reported in KiB-units of 1 KB by rocprofv3 on
gfx950 -> bytes; FETCH_SIZE doubled for the kernels whose reads are 16 B per lane -- LDS-DMA and dwordx4 streams --
as /opt/skills/guides/MI355X_MICROARCH.md, section HBM, prescribes: the counter tallies 128-B requests at 64 B),
MFMA-busy fraction, LDS bank-conflict fraction, L2 hit rate. A step = the dispatches between two stem launches."""
import collections
import csv
import glob
import json
import os
import re
import sys


def family(name):
    n = name
    m = re.search(r"conv_igemm(_grouped)?_kernel<([^>]*)>", n)
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        return ("conv_igemm_dgrad" if args[5] == "true" else "conv_igemm_fwd") + " %sx%s" % (args[0], args[1]) + ("g" if m.group(1) else "")
    for key in ("wgrad_mixed_grouped_kernel", "wgrad3_grouped_kernel", "wgrad3_kernel", "wgrad_grouped_kernel", "wgrad_reduce_grouped_kernel", "wgrad_kernel", "stem_pool_kernel",
                "roi_align_bwd_gather_kernel", "roi_bwd_tab_kernel", "roi_bwd_rows_kernel", "roi_bwd_seg_kernel", "roi_bwd_box_kernel",
                "roi_align_fwd_kernel", "sgd_kernel",
                "filter_transpose_batched_kernel", "nms_scan_rows_kernel", "nms_mask_kernel", "proposal_", "anchor_", "rpn_loss_kernel",
                "rcnn_loss_kernel", "upsample2_bwd_kernel", "proposal_target_kernel"):
        if key in n:
            return key.rstrip("_")
    return "other"


WIDE = ("conv_igemm", "wgrad_mixed", "wgrad3", "wgrad_grouped", "wgrad_kernel", "wgrad_reduce", "sgd_kernel", "filter_transpose", "upsample2")


def main():
    root, out = sys.argv[1], sys.argv[2]
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for f in sorted(glob.glob(os.path.join(root, "pass*", "c_counter_collection.csv"))):
        rows = list(csv.DictReader(open(f)))
        # dispatches in order; one step = from the last stem launch to the end is incomplete, take the one before
        disp = collections.OrderedDict()
        for r in rows:
            disp.setdefault(int(r["Dispatch_Id"]), (r["Kernel_Name"], {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(disp)
        stems = [i for i in ids if "stem_pool_kernel" in disp[i][0] or "stem_conv_kernel" in disp[i][0]]
        if len(stems) < 2:
            continue
        lo, hi = stems[-2], stems[-1]
        first_pass = not launches
        for i in ids:
            if lo <= i < hi:
                name, c = disp[i]
                fm = family(name)
                if first_pass:
                    launches[fm] += 1
                for k, v in c.items():
                    fam[fm][k] += v
    res = {}
    for fm, c in sorted(fam.items()):
        e = {"launches_per_step": launches.get(fm, 0)}
        if "FETCH_SIZE" in c:
            corr = 2.0 if fm.startswith(WIDE) else 1.0
            e["hbm_read_bytes"] = int(c["FETCH_SIZE"] * 1024 * corr)
            e["fetch_size_correction"] = corr
        if "WRITE_SIZE" in c:
            e["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if c.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            e["gpu_cycles"] = int(cycles)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            e["wave_cycles_split"] = {"issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3), "waiting": round(c.get("SQ_WAIT_ANY", 0) / w, 3),
                                      "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)}
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        res[fm] = e
    agg = {}
    for grp in ("conv_igemm_fwd", "conv_igemm_dgrad", "conv_wgrad"):
        keys = [k for k in res if (k.startswith("wgrad") if grp == "conv_wgrad" else k.startswith(grp))]
        agg[grp] = {"hbm_bytes": sum(res[k].get("hbm_bytes", 0) for k in keys), "launches_per_step": sum(res[k]["launches_per_step"] for k in keys),
                    "members": keys}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"csrc_sha16": bench.csrc_sha16(),
               "note": "one eager training step, Faster R-CNN R50-FPN batch 2 (bench.py --no-graph); counters per kernel family, "
                       "rocprofv3 --pmc, one pass per counter group (tools/pmc_step.sh)", "families": agg, "kernels": res},
              open(out, "w"), indent=1)
    for k, v in agg.items():
        print(k, v["launches_per_step"], "launches", "%.1f MB" % (v["hbm_bytes"] / 1e6))
    for k, v in res.items():
        print("  %-34s x%-3d %8.1f MB  mfma %s  lds-conflict %s  l2-hit %s" % (k, v["launches_per_step"], v.get("hbm_bytes", 0) / 1e6,
              v.get("mfma_busy_frac"), v.get("lds_bank_conflict_frac"), v.get("l2_hit_rate")))


if __name__ == "__main__":
    main()
