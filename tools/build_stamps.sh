#!/bin/bash
# Diagnostic build of the library with in-kernel clock stamps (tools/phase_stamps.py).
cd "$(dirname "$0")/../monocular-depth-estimation_amd"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fno-slp-vectorize -shared -fPIC -DVSL_STAMPS $VSL_EXTRA_DEFS \
  -I ../include -o ../tools/libvsloss_stamps.so csrc/vsl_loss.hip csrc/vsl_layers.hip csrc/vsl_pose.hip csrc/vsl_metrics.hip
