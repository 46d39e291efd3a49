#!/bin/bash
# Build a variant of libexabm4d.so for A/B runs on one GPU box:
#   tools/dbg/build_variant.sh NAME [-DFLAG=VALUE ...]   ->  tools/dbg/variants/libexabm4d_NAME.so
# (select it at run time with EXABM4D_LIB=tools/dbg/variants/libexabm4d_NAME.so)
set -e
name=$1; shift
here=$(cd "$(dirname "$0")" && pwd)
src=${SRC:-$here/../../aind-exaspim-image-compression_amd/csrc}
out=$here/variants
mkdir -p "$out/obj_$name"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
for f in exabm4d_api comm_rccl bm_kernels stage_kernels elementwise_kernels metrics_kernels codec_kernels rans_kernels rans2_kernels nn_kernels; do
  /opt/rocm/bin/hipcc $flags "$@" -c "$src/$f.hip" -o "$out/obj_$name/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$out/libexabm4d_$name.so" "$out"/obj_$name/*.o -ldl
rm -rf "$out/obj_$name"
echo "built $out/libexabm4d_$name.so"
