# Regenerates the committed evidence under profiles/ (run on the GPU box from the repo root): tag = $1
set -e
tag=${1:-r01_g}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
# (1) default bench line (includes cpu_baseline)
timeout -k 10 900 python bench.py > gpurun_out/$tag/bench.json.log 2>&1
tail -1 gpurun_out/$tag/bench.json.log | cut -c1-300
# (2) kernel stats of the same command (no cpu leg under the profiler)
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -o st -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$tag/stats.log 2>&1)
# (3) per-shape conv report + timeline of one replayed step
timeout -k 10 600 python tools/conv_report.py > gpurun_out/$tag/conv_report.txt 2>&1
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/tl -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-conv-timer --no-cpu-baseline > $R/gpurun_out/$tag/tl.log 2>&1)
python tools/timeline.py gpurun_out/$tag/tl/tl_kernel_trace.csv > gpurun_out/$tag/timeline.txt
# (4) HBM traffic of the heaviest layer, one counter per pass
for kind in fwd dgrad wgrad; do for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && MXDET_EAGER_TIMING=1 timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc/${kind}_$ctr -o c -- python3 $R/tools/bench_one_conv.py $kind 2 200 336 256 256 3 1 3 0 > $R/gpurun_out/$tag/pmc_${kind}_$ctr.log 2>&1)
done; done
python tools/pmc_traffic.py gpurun_out/$tag/pmc gpurun_out/$tag/pmc_traffic.json > /dev/null
echo refreshed $tag
