#!/bin/bash
# Register / spill / LDS usage of every kernel of mdx_kernels.hip (gfx950), from the compiler's own report.
# usage: tools/kstats.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -c mapdamage_amd/csrc/mdx_kernels.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "remark:" | sed -e 's/^.*remark: //' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//' | \
  awk '/Function Name/ {if (line) print line; line=$3} /TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|Spill/ {gsub(/^ +/,""); line=line" | "$0} END {print line}' | \
  sed -e 's/ \[bytes\/lane\]//' -e 's/ \[waves\/SIMD\]//' | c++filt | cut -c1-260
