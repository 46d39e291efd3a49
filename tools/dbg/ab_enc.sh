#!/bin/bash
# encode legs A/B on one box: tools/dbg/ab_enc.sh SIZE LIB1 LIB2 ...
size=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset EXABM4D_LIB; else export EXABM4D_LIB=$PWD/tools/dbg/variants/libexabm4d_$v.so; fi
  python bench.py --size $size --steps 3 --warmup 1 --cpu-sample 0 --bm4dnet 0 > gpurun_out/abe_$v.json 2> gpurun_out/abe_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/abe_$v.err; continue; }
  python -c "
import json;d=json.load(open('gpurun_out/abe_$v.json'));p=d['phase_ms'];print('$v', round(d['ms_per_step'],1), {k:round(p[k],2) for k in ('encode_u16','dct_quantise','encode_idx')}, d['encoded']['lossless_bytes'], round(d['encoded']['dct_bits_per_voxel'],5))"
done
