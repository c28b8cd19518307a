# HBM traffic and pipe counters of every kernel of ONE training step (eager launches: counters are per dispatch), one
# rocprofv3 --pmc pass per counter group:  tools/pmc_step.sh <tag> [bench args]      (GPU box, repo root)
#   -> gpurun_out/<tag>/pmc_step.json (tools/pmc_summary.py); copy it to profiles/<round>_pmc_step.json
set -e
tag=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pass$i -o c -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-conv-timer --no-cpu-baseline --no-instep "$@" > $R/gpurun_out/$tag/pass$i.log 2>&1)
  echo "pass $i ($ctrs) done"
done
python tools/pmc_summary.py gpurun_out/$tag gpurun_out/$tag/pmc_step.json
