#!/bin/bash
# tools/dbg/sweep_chunks.sh SIZE N1 N2 ...: stage-kernel times for z chunk counts (one box)
size=$1; shift
for n in "$@"; do
  EXABM4D_STAGE_CHUNKS=$n python bench.py --size $size --steps 2 --warmup 1 --cpu-sample 0 --bm4dnet 0 --no-encode > gpurun_out/sw_$n.json 2> gpurun_out/sw_$n.err
  python -c "
import json;d=json.load(open('gpurun_out/sw_$n.json'));p=d['phase_ms'];print($n, round(d['ms_per_step'],1), round(p['stage_ht'],1), round(p['stage_wie'],1))"
done
