#!/bin/bash
# usage: scripts/build_variant.sh <name> [extra hipcc flags]   -> tmp_variants/lib_<name>.so (kernels.hip rebuilt with the flags)
set -e
name=$1; shift
mkdir -p tmp_variants build/obj
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter "${WAVES_FLAG:--DRT_EXTEND_WAVES=4}" "$@" -c raytracer_amd/csrc/device/kernels.hip -o build/obj/kernels_$name.o 2>/dev/null
objs=$(ls build/obj/*.o | grep -v "kernels")  # (the variant is one unit: no RT_PART)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tmp_variants/lib_$name.so $objs build/obj/kernels_$name.o -lz
