#!/bin/bash
# builds A/B variants of the library (parallel, own object directory): tools/build_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/../youtube-8m_amd/csrc"
name=$1; shift
mkdir -p ../../tools/variants
make -j8 BUILD=build_$name LIB=../../tools/variants/lib_$name.so EXTRA="$*" > /dev/null
echo built tools/variants/lib_$name.so
