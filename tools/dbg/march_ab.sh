#!/bin/bash
# A/B of block matching's march on one box: EXABM4D_BM_MARCH = 0 (one tile per workgroup), 1 (automatic), n blocks
cd "$(dirname "$0")/../.."
for m in ${@:-0 1 0 1 8 4}; do
  EXABM4D_BM_MARCH=$m timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bm4dnet 0 --cpu-sample 0 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('march $m', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['phase_ms'].items() if 'blockmatch' in k}, flush=True)" || exit 1
done
