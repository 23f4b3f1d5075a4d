#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags...]  -> audiodiffuser_amd/build/variants/libadf_hip_NAME.so
# A second build of the library (e.g. -DADF_PP_STAMP, -fno-slp-vectorize) to A/B against the product build inside one
# gpurun call: ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_NAME.so python bench.py ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/audiodiffuser_amd/build/variants; mkdir -p $out/$name
objs=""
for f in adf_api adf_net_unet1d adf_net_wavenet adf_net_adm adf_sampler adf_bench_replay adf_gemm adf_kernels adf_wavenet adf_conv2d; do
  extra=""; [ $f = adf_gemm ] && [ -z "$ADF_VARIANT_SLP" ] && extra="-fno-slp-vectorize"      # as audiodiffuser_amd/build.py EXTRA_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $extra "$@" -c $root/audiodiffuser_amd/csrc/$f.hip -o $out/$name/$f.o &
  objs="$objs $out/$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libadf_hip_$name.so $objs
echo $out/libadf_hip_$name.so
