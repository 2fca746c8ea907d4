#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME "-DFLAG=..." [file.hip ...]
# recompiles the named translation units (default: the four fused-shape units) with the extra flags and links
# ldpc_decoders_amd/csrc/variants/libldpc_hip_NAME.so from them plus the regular objects; select it with LDPC_LIB_PATH.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
CSRC=$(cd "$(dirname "$0")/../ldpc_decoders_amd/csrc" && pwd)
UNITS=${@:-ldpc_fused_shapes_f32_dc6.hip ldpc_fused_shapes_f32_dcx.hip ldpc_fused_shapes_f64_dc6.hip ldpc_fused_shapes_f64_dcx.hip}
mkdir -p $CSRC/variants/$NAME
make -s -C $CSRC -j8 libldpc_hip.so
OBJS=""
for f in ldpc_api ldpc_stream ldpc_fused ldpc_fused_shapes_f32_dc6 ldpc_fused_shapes_f32_dcx ldpc_fused_shapes_f64_dc6 ldpc_fused_shapes_f64_dcx ldpc_fused_shapes_bec ldpc_bec_stream ldpc_channel ldpc_layout ldpc_ml ldpc_admm; do
  if echo " $UNITS " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $FLAGS -c $CSRC/$f.hip -o $CSRC/variants/$NAME/$f.o &
    OBJS="$OBJS $CSRC/variants/$NAME/$f.o"
  else
    OBJS="$OBJS $CSRC/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $CSRC/variants/libldpc_hip_$NAME.so $OBJS
echo $CSRC/variants/libldpc_hip_$NAME.so
