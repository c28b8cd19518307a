#!/bin/bash
# whole-step A/B of the bucket boundaries at N = 1 (MXDET_TUNE_BUCKETS names the reduce points that do NOT close a bucket)
run() { echo -n "buckets='$1' $2: "; MXDET_TUNE_BUCKETS="$1" $2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 60 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
for b in 123 "" 1 12 13 23 2 3 123; do run "$b" "$@"; done
