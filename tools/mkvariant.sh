#!/bin/bash
# Builds a variant of the library for A/B runs on the GPU box: tools/bin/libmdx_<tag>.so (git-ignored, travels with gpurun).
# usage: tools/mkvariant.sh <tag> [-DMDX_...=... more hipcc flags]      then e.g.  gpurun -- 'python tools/ab.py --packed cur <tag>'  or  MDX_LIBPATH=tools/bin/libmdx_<tag>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/bin
S=mapdamage_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -Wall -Wno-unused-function "$@" \
  $S/mdx_kernels.hip $S/mdx_capi.cpp $S/mdx_bamio.cpp $S/mdx_gbam.hip $S/mdx_libsort.hip $S/mdx_fasta.hip -lz -lpthread -ldl -o tools/bin/libmdx_$tag.so
echo "built tools/bin/libmdx_$tag.so ($*)"
