# marginal cost of the step's components in the overlapped schedule: bench.py with one component left out at a time
# (timing only, MXDET_ABL_SKIP, see models/utils/detector.py).  tools/ablate_step.sh <tag>      (GPU box, repo root)
tag=${1:-abl}
mkdir -p gpurun_out/$tag
for rep in 1 2; do
for a in none front sgd transpose wgrad front,wgrad; do
  v=$a; [ $a = none ] && v=""
  MXDET_ABL_SKIP=$v timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-conv-timer --no-cpu-baseline --no-instep > gpurun_out/$tag/$rep.$a.log 2>&1
  python - "$a" gpurun_out/$tag/$rep.$a.log <<'P'
import json,sys
v=None
for l in open(sys.argv[2]):
    if l.startswith("{"): v=json.loads(l)
print("skip %-12s %s" % (sys.argv[1], "%.1f img/s  %.3f ms" % (v["value"], v["ms_per_step"]) if v else "FAILED"))
P
done
done | tee gpurun_out/$tag/summary.txt
