#!/bin/bash
# Diagnostic builds next to the shipped library:  tools/build_variant.sh NAME [-DFLAG ...]  ->  build/libaprilslam_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -Wno-unused-but-set-variable -ldl "$@" -o build/libaprilslam_$name.so aprilslam_amd/csrc/aprilslam.hip
echo build/libaprilslam_$name.so
