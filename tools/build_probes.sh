#!/bin/bash
# stand-alone measurement programs (HBM ceilings etc.); binaries land in tools/bin/ (git-ignored, travel with gpurun)
set -e
cd "$(dirname "$0")"
mkdir -p bin
for f in probe_hbm probe_pipe probe_bfly probe_issue; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/$f $f.hip
done
