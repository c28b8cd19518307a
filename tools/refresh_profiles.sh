# Regenerates the evidence under gpurun_out/<tag>/ (run on the GPU box from the repo root); copy what is to be tracked
# into profiles/ with the round prefix.   tag = $1 (default r02)
set -e
tag=${1:-r03}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
# (1) default bench line (includes cpu_baseline)
timeout -k 10 900 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
tail -1 gpurun_out/$tag/bench_default.json | cut -c1-200
# (2) kernel stats of the same command (no cpu leg under the profiler)
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/stats -o st -- python3 $R/bench.py --no-cpu-baseline --no-instep > $R/gpurun_out/$tag/stats.log 2>&1)
cp gpurun_out/$tag/stats/st_kernel_stats.csv gpurun_out/$tag/kernel_stats_bench_default.csv
# (3) per-shape conv report + timeline of one replayed step
timeout -k 10 600 python tools/conv_report.py > gpurun_out/$tag/conv_per_shape_report.txt 2>&1
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag/tl -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-conv-timer --no-cpu-baseline > $R/gpurun_out/$tag/tl.log 2>&1)
python tools/timeline.py gpurun_out/$tag/tl/tl_kernel_trace.csv > gpurun_out/$tag/timeline_one_replayed_step.txt
# (4) the other two configurations, same harness
for m in mask_rcnn retinanet; do
  timeout -k 10 600 python bench.py --model $m --no-cpu-baseline > gpurun_out/$tag/bench_$m.json 2> gpurun_out/$tag/bench_$m.err
  tail -1 gpurun_out/$tag/bench_$m.json | cut -c1-160
done
rm -rf gpurun_out/$tag/stats gpurun_out/$tag/tl
echo refreshed $tag
