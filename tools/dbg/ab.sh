#!/bin/bash
# A/B on one box: tools/dbg/ab.sh SIZE LIB1 LIB2 ... (LIB = "default" or a variant name)
size=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset EXABM4D_LIB; else export EXABM4D_LIB=$PWD/tools/dbg/variants/libexabm4d_$v.so; fi
  python bench.py --size $size --steps 3 --warmup 1 --cpu-sample 0 --bm4dnet 0 --end-to-end 0 --no-encode > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
p=d["phase_ms"]
print("$v", round(d["ms_per_step"],1), {k:round(p[k],1) for k in ("blockmatch_ht","stage_ht","blockmatch_wie","stage_wie")}, round(d["residual_std"],3))
PY
done
