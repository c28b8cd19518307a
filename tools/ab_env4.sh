#!/bin/bash
# as tools/ab_env.sh, four alternating rounds of 100 steps without the in-step profile (for differences of ~1 %):
#   tools/ab_env4.sh "A=1" "B=2 C=3" ...   ("-" = no override)
run() { if [ "$1" = "-" ]; then e=""; else e="$1"; fi
  echo -n "[$1]: "; env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --no-instep --steps 100 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
for r in 1 2 3 4; do for a in "$@"; do run "$a"; done; done
