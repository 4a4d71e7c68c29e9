#!/bin/bash
# Register / spill / LDS figures of the kernels of one source file under extra -D flags (cross-compiles, no GPU):
#   tools/isa_stats.sh scan.hip "-DSLIDE_TILES=1 -DSLIDE_WGS=2" [kernel-name-pattern]
# The ISA is left in /tmp/isa_<hash>/ for reading.
src=$1; flags=$2; pat=${3:-.}
root=$(cd "$(dirname "$0")/.." && pwd)
dir=/tmp/isa_$(echo "$src $flags" | md5sum | cut -c1-8)
mkdir -p "$dir"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -x hip -c "$root/libbtbb_amd/csrc/$src" \
	-save-temps=obj -o "$dir/out.o" 2>&1 | grep -E "error|warning"
s=$(ls "$dir"/*gfx950.s)
echo "ISA: $s"
awk -v pat="$pat" '/^    \.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_spill_count:/ {sp=$2}
	/\.private_segment_fixed_size:/ {pr=$2} /\.group_segment_fixed_size:/ {lds=$2}
	/\.wavefront_size:/ { if (name ~ pat) printf "%-60s vgpr %3d sgpr %3d spill %d scratch %d lds %d\n", name, v, sg, sp, pr, lds }' "$s"
