#!/bin/bash
# Build of the library with extra definitions, for A/B timing: tools/build_variant.sh <name> <defs...> -> tools/libvsloss_<name>.so
cd "$(dirname "$0")/../monocular-depth-estimation_amd"
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fno-slp-vectorize -shared -fPIC "$@" \
  -I ../include -o ../tools/libvsloss_$name.so csrc/vsl_loss.hip csrc/vsl_layers.hip csrc/vsl_pose.hip csrc/vsl_metrics.hip
