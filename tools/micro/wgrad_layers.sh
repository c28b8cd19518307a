# per-layer HBM-side bytes of the single-layer weight-gradient launches:  tools/micro/wgrad_layers.sh <tag> ["ENV=.."]
set -e
tag=$1; envs=${2:-MXDET_NOP=1}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && export $envs && timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/$tag/pass$i -o c -- python3 $R/tools/micro/wgrad_layers.py run > $R/gpurun_out/$tag/pass$i.log 2>&1)
done
(cd /tmp && export $envs && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/trace -o c -- python3 $R/tools/micro/wgrad_layers.py run > $R/gpurun_out/$tag/trace.log 2>&1)
python tools/micro/wgrad_layers_report.py gpurun_out/$tag | tee gpurun_out/$tag/report.txt
