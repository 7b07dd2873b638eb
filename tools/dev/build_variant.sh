#!/bin/bash
# build vilgod_amd/libvilgod_hip_<tag>.so = the product objects with ONE source recompiled with extra -D flags (A/B runs on the GPU box:
# VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_<tag>.so).   usage: build_variant.sh <tag> <source> <flags...>
set -e
cd "$(dirname "$0")/../.."
tag=$1; src=$2; shift 2
python -m vilgod_amd.build > /dev/null
extra=""; case $src in vit.hip|api.hip) ;; *) extra="-ffp-contract=off";; esac
mkdir -p /tmp/vg_variant_$tag
hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Iinclude -Ivilgod_amd/csrc -Wno-unused-result -DNDEBUG $extra "$@" -c vilgod_amd/csrc/$src -o /tmp/vg_variant_$tag/${src%.*}.o
objs=""
for o in vilgod_amd/csrc/_obj/*.o; do b=$(basename $o); if [ "$b" = "${src%.*}.o" ]; then objs="$objs /tmp/vg_variant_$tag/$b"; else objs="$objs $o"; fi; done
hipcc -shared -fPIC --offload-arch=gfx950 -o vilgod_amd/libvilgod_hip_$tag.so $objs
echo vilgod_amd/libvilgod_hip_$tag.so
