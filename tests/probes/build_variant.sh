#!/bin/bash
# Builds vall-e_amd/csrc/<name>.so = the product library with extra defines (A/B knobs): build_variant.sh name [-DKNOB=v ...]
name=$1; shift
cd "$(dirname "$0")/../../vall-e_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc \
  -Wno-unused-function -Wno-unused-variable -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form "$@" -o $name.so engine.hip
