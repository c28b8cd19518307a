# Exchange schedule at world size 1 under rocprofv3 --kernel-trace: when does the NEXT step's front end (stem kernel) start
# relative to the previous step's last weight-gradient fold?   tools/xtrace.sh <tag> "ENV=.. ENV=.."      (GPU box, repo root)
tag=$1; envs=$2
R=$PWD
mkdir -p gpurun_out/$tag
(cd /tmp && export TMPDIR=/tmp MXDET_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 $envs && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/tl -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-conv-timer --no-cpu-baseline --no-instep > $R/gpurun_out/$tag/tl.log 2>&1)
python3 - $tag <<'P'
import csv,sys
rows=list(csv.DictReader(open('gpurun_out/%s/tl/tl_kernel_trace.csv'%sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
st=[i for i,r in enumerate(rows) if 'stem_pool' in r['Kernel_Name']]
red=[int(r['End_Timestamp']) for r in rows if 'wgrad_reduce' in r['Kernel_Name']]
qs=sorted(set(r['Queue_Id'] for r in rows))
for i in st[-2:]:
    s=int(rows[i]['Start_Timestamp']); prev=[e for e in red if e<=s]; nxt=[e for e in red if e>s]
    print(sys.argv[1], "stem q%s starts %.1f us after previous reduce end, %.1f us before the next reduce end; queues %s; period %.1f us"%(rows[i]['Queue_Id'],(s-prev[-1])/1e3,(nxt[0]-s)/1e3,qs,(int(rows[st[-1]]['Start_Timestamp'])-int(rows[st[-2]]['Start_Timestamp']))/1e3))
P
