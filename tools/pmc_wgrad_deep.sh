# issue / latency counters of the weight-gradient kernels of one eager step under an environment setting (paths in it must be
# absolute: the profiler runs from /tmp):
# library): tools/pmc_wgrad_deep.sh <tag> "ENV=1 ..."      (GPU box, repo root)
tag=$1; envs=${2:-MXDET_NOP=1}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "VmemLatency" "LdsLatency" "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && export $envs MXDET_REPO=$R && timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$tag/pass$i -o c -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-conv-timer --no-cpu-baseline --no-instep > $R/gpurun_out/$tag/pass$i.log 2>&1) || echo "pass $i failed"
done
python - $tag <<'P'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/%s/pass*/c_counter_collection.csv" % tag)):
    rows = list(csv.DictReader(open(f)))
    disp = {}
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), (r["Kernel_Name"], {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(disp)
    stems = [i for i in ids if "stem_pool_kernel" in disp[i][0]]
    lo, hi = stems[-2], stems[-1]
    for i in ids:
        if lo <= i < hi and "wgrad" in disp[i][0]:
            k = disp[i][0].split("(")[0].replace("void mxdet::", "")
            for c, v in disp[i][1].items():
                acc[k][c] += v
for k, c in sorted(acc.items()):
    print(k)
    for n, v in sorted(c.items()):
        print("   %-40s %16.1f" % (n, v))
P
