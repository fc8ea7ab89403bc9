#!/bin/bash
# Container: one source of libvtmhip.so recompiled with extra -D flags and linked with the other (current) objects into vtm_amd/libvtmhip_<tag>.so (for scripts/gpu_lib_variants.sh).
# usage: scripts/build_variant.sh <tag> <source name without .hip> <flags...>
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift; shift
mkdir -p vtm_amd/_obj/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wall -Wno-unused-function "$@" -c vtm_amd/csrc/$SRC.hip -o vtm_amd/_obj/var/${SRC}_$TAG.o
OBJS=$(ls vtm_amd/_obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS vtm_amd/_obj/var/${SRC}_$TAG.o -o vtm_amd/libvtmhip_$TAG.so
echo built vtm_amd/libvtmhip_$TAG.so
