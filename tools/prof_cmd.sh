# kernel trace of an arbitrary python tool: tools/prof_cmd.sh <tag> <script.py> args...   -> per-kernel avg durations
set -e
tag=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/kt -o kt -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$tag/kt.log 2>&1)
python - <<PY
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/$tag/kt/kt_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows: d[(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v=sorted(v); print('%-62s grid %8s wg %4s n=%3d min %8.1f med %8.1f us' % (k[0],k[1],k[2],len(v),v[0],v[len(v)//2]))
PY
