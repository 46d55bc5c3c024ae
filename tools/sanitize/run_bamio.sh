#!/bin/bash
# ASan + UBSan over the native BAM decoder (the host part of mapdamage_amd/csrc/mdx_bamio.cpp, -DMDX_HOST_ONLY): one-piece and
# chunked decode of BAM files in both block layouts, with the parallel record scan forced on and off, plus a
# truncated and a garbage file.  CPU only.  Usage: tools/sanitize/run_bamio.sh
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
g++ -std=c++17 -O1 -g -DMDX_HOST_ONLY -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    "$ROOT/mapdamage_amd/csrc/mdx_bamio.cpp" "$ROOT/tools/sanitize/bamio_driver.cpp" -lz -lpthread -o "$TMP/driver"
cd "$ROOT"
python - "$TMP" <<'PY'
import sys
from mapdamage_amd import sam, synth
tmp = sys.argv[1]
ref, batch = synth.config1_batch()
big = synth.config3_batch(synth.make_genome(), 8000, seed=3)
g = synth.make_genome()
rgs = [{"ID": "rg1", "SM": "s", "LB": "l"}]
for name, b, r in (("small", batch, ref), ("big", big, g)):
    for layout in (True, False):
        sam.write_bam("%s/%s_%d.bam" % (tmp, name, layout), b, r.names, r.lengths, rgs, ["rg1"] * b.n, htslib_blocks=layout)
data = open(tmp + "/big_1.bam", "rb").read()
open(tmp + "/cut.bam", "wb").write(data[:len(data) // 2 + 7])
open(tmp + "/garbage.bam", "wb").write(b"\x1f\x8bnot a bam at all" * 10)
PY
export ASAN_OPTIONS=detect_leaks=1
for scan_min in 0 999999999999; do
    for f in small_1 small_0 big_1 big_0; do
        echo "== $f.bam  MDX_BAM_PARALLEL_SCAN_MIN=$scan_min"
        MDX_BAM_PARALLEL_SCAN_MIN=$scan_min "$TMP/driver" "$TMP/$f.bam"
    done
done
for f in cut garbage; do
    echo "== $f.bam (must fail cleanly)"
    MDX_BAM_PARALLEL_SCAN_MIN=0 "$TMP/driver" "$TMP/$f.bam" expect-error
done
echo "bamio: ASan/UBSan clean"
