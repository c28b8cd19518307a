#!/bin/bash
# one whole-step bench value per environment setting: tools/env_probe.sh "A=1" "B=2" ...   (never ROC_SYSTEM_SCOPE_SIGNAL=0: hangs)
for e in "$@"; do
  case "$e" in *ROC_SYSTEM_SCOPE_SIGNAL=0*) echo "[$e]: refused (hangs the graph replay: DESIGN 9b)"; continue;; esac
  echo -n "[$e]: "; env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 60 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1
done
