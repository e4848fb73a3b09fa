#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> tools/libsmr_hip_<name>.so (diagnostic builds for A/B timing)
N=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ismrender_amd/csrc -Wno-unused-value -Wno-unused-result "$@" -x hip -shared -o tools/libsmr_hip_$N.so smrender_amd/csrc/*.hip smrender_amd/csrc/*.cpp -ldl -lz -lpthread
