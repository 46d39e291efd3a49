#!/bin/bash
# hard-threshold stage only (bench --stages 1): tools/dbg/ab1.sh SIZE LIB ...
size=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset EXABM4D_LIB; else export EXABM4D_LIB=$PWD/tools/dbg/variants/libexabm4d_$v.so; fi
  python bench.py --size $size --stages 1 --steps 3 --warmup 1 --cpu-sample 0 --bm4dnet 0 --no-encode > gpurun_out/ab1_$v.json 2> gpurun_out/ab1_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab1_$v.err; continue; }
  python -c "
import json;d=json.load(open('gpurun_out/ab1_$v.json'));p=d['phase_ms'];print('$v', round(d['ms_per_step'],1), round(p['blockmatch_ht'],1), round(p['stage_ht'],1), round(d['residual_std'],3))"
done
