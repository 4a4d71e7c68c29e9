#!/bin/bash
# Build A/B variants of the library for the GPU box: libbtbb_amd/variants/<name>.so = the normal build with the
# listed sources recompiled under extra -D flags (the other objects are taken from csrc/build, so `make` first).
#   tools/build_variants.sh "scan.hip context.cpp" prof "-DSCAN_PROFILE" x "-DSOME_SWITCH_UNDER_TEST"
#   tools/build_variants.sh "packet.hip" tlp "-DTL_PROFILE"
# Then, on the box (LIBBTBB_AMD_SO selects the library the Python view loads):
#   tools/ab_test.sh libbtbb_amd/variants/u3.so      scan tests against one variant
#   tools/ab_variants.sh [steps]                     headline bench for the normal build and every variant
#   tools/ab_trials.sh                               the bench's secondary block for each
#   tools/ab_profile.sh                              phase profile of variants named p*.so built with -DSCAN_PROFILE
# libbtbb_amd/variants/ is git-ignored; it travels to the box with the snapshot.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/libbtbb_amd/csrc"
srcs=$1; shift
mkdir -p ../variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  (
    tmp=$(mktemp -d)
    objs=""; skip=""
    for src in $srcs; do
      o=$tmp/$(basename "${src%.*}").o
      $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $flags -x hip -c "$src" -o "$o"
      objs="$objs $o"; skip="$skip|$(basename "${src%.*}").o"
    done
    rest=$(ls build/*.o | grep -v "/asan_" | grep -Ev "/(${skip#|})$")
    $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,-soname,libbtbb.so.1 -Wl,--version-script=exports.map $rest $objs -o ../variants/$name.so
    rm -rf "$tmp"
    echo "built $name"
  ) &
done
wait
