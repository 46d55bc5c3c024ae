#!/bin/bash
# A variant of the library for A/B runs, fast: only mdx_kernels.hip is compiled afresh (with the flags given), the other
# objects are the in-tree build's (mapdamage_amd/build/obj).  usage: tools/mkv.sh <tag> [-DMDX_...=...]  -> tools/bin/libmdx_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/bin
S=mapdamage_amd/csrc; O=mapdamage_amd/build/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-function -Wno-unused-but-set-variable "$@" -c $S/mdx_kernels.hip -o tools/bin/k_$tag.o
hipcc --offload-arch=gfx950 -fPIC -shared tools/bin/k_$tag.o $O/mdx_capi.cpp.o $O/mdx_bamio.cpp.o $O/mdx_gbam.hip.o $O/mdx_libsort.hip.o $O/mdx_fasta.hip.o -lz -lpthread -ldl -o tools/bin/libmdx_$tag.so
rm -f tools/bin/k_$tag.o
echo "built tools/bin/libmdx_$tag.so ($*)"
