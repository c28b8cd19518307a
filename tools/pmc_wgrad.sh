# HBM-side traffic and L2 hit rate of the weight-gradient kernels of one eager step under an environment setting:
#   tools/pmc_wgrad.sh <tag> "ENV=1 ..."      (GPU box, repo root)
set -e
tag=$1; envs=$2
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && export $envs && timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pass$i -o c -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-conv-timer --no-cpu-baseline --no-instep > $R/gpurun_out/$tag/pass$i.log 2>&1)
done
python tools/pmc_summary.py gpurun_out/$tag gpurun_out/$tag/pmc_step.json | grep -i "wgrad"
