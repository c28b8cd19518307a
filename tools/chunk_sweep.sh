#!/bin/bash
# wgrad grouped-launch persistent-grid sweep (run on the GPU box): MXDET_WGRAD_PERSIST = workgroups in the grid, 0 = off
for c in 0 256 384 512 768 1024; do
  echo "persist $c: $(MXDET_WGRAD_PERSIST=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-conv-timer --steps 30 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["losses_last_step"][:2])')" || exit 1
done
