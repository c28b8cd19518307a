#!/bin/bash
# whole-step A/B of environment settings on one box: tools/ab_env.sh "A=1" "B=2 C=3" ...   ("-" = no override); two rounds
run() { if [ "$1" = "-" ]; then e=""; else e="$1"; fi
  echo -n "[$1]: "; env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 60 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
for r in 1 2; do for a in "$@"; do run "$a"; done; done
