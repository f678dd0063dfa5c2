#!/bin/bash
# variants/libzmi355_NAME.so from the working tree with extra compiler flags (-DEXPERIMENT ...): kernel A/B work on the GPU box
#   tools/build_variant.sh NAME [flags...]      then: ZMI_LIB=variants/libzmi355_NAME.so python tools/gpu_fast_probe.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p "$R/variants"
cd "$R/zlib_rs_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
  -o "$R/variants/libzmi355_$N.so" gen.hip checksum.hip lz77.hip parse.hip encode.hip encode_cp.hip inflate.hip resolve_jump.hip pack.hip blockscan.hip exchange.hip zmi_api.hip -ldl 2>&1 | grep -v "warning\|^$" || true
ls -la "$R/variants/libzmi355_$N.so"
