# One replayed step as a kernel trace + timeline: tools/prof_step.sh <tag> [extra bench args]   (GPU box, repo root)
set -e
tag=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/tl -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-conv-timer --no-cpu-baseline "$@" > $R/gpurun_out/$tag/tl.log 2>&1)
python tools/timeline.py gpurun_out/$tag/tl/tl_kernel_trace.csv > gpurun_out/$tag/timeline.txt
rm -rf gpurun_out/$tag/tl/*agent_info* 
tail -75 gpurun_out/$tag/timeline.txt
