#!/bin/bash
# A/B of one bench.py environment switch on one box:  bash tools/dbg/env_ab.sh VAR v1 v2 ...   (ENV_OPTIONS in bench.py)
cd "$(dirname "$0")/../.."
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bm4dnet 0 --cpu-sample 0 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['ms_per_step'],1), {k: round(x,1) for k,x in d['phase_ms'].items() if k.startswith(('stage','blockmatch'))}, flush=True)" || exit 1
done
