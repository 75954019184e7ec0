#!/bin/bash
# builds gpurun_out-independent A/B variants of the library: tools/build_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/../youtube-8m_amd/csrc"
name=$1; shift
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on "$@" -shared -o ../../tools/variants/lib_$name.so *.hip
echo built tools/variants/lib_$name.so
