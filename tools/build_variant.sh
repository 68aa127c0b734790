#!/bin/bash
# build a variant of libmsmhip.so with one source recompiled under extra -D flags: tools/build_variant.sh NAME SRC.hip -DX=1 ...
# (the objects of the other sources come from pmarlo_amd/csrc/build; the variant lands in tools/probe/_bin/libmsmhip_NAME.so)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
obj=tools/probe/_bin/${src%.hip}_$name.o
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -c pmarlo_amd/csrc/$src -o $obj
others=$(ls pmarlo_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/_bin/libmsmhip_$name.so $obj $others
echo tools/probe/_bin/libmsmhip_$name.so
