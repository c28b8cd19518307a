# PMC counters (one pass per group) of a python tool: tools/pmc_cmd.sh <tag> "<C1 C2 ...>" <script.py> args...
set -e
tag=$1; ctrs=$2; shift 2
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc -o c -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/$tag/pmc.log 2>&1)
python - <<PY
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/$tag/pmc/c_counter_collection.csv')))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: d[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'at::' in k or 'elementwise' in k: continue
    print(k)
    for c,vals in sorted(v.items()): print('   %-32s n=%d median %.4g' % (c,len(vals),sorted(vals)[len(vals)//2]))
PY
