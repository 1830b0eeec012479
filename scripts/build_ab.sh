#!/bin/bash
# usage: scripts/build_ab.sh <name> [extra hipcc flags]  -> ab/lib_<name>.so: the f64 kernels (RT_PART 0-2) rebuilt with the flags,
# linked with the objects of the last `make` (ab/ is git-ignored but travels to the GPU box; RT_MI355X_LIB=ab/lib_<name>.so selects it)
set -e
name=$1; shift
mkdir -p ab build/ab
for p in 0 1 2; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter -Wno-pass-failed -DRT_EXTEND_WAVES=4 -DRT_PART=$p "$@" \
    -c raytracer_amd/csrc/device/kernels.hip -o build/ab/kernels_${name}_$p.o 2>build/ab/${name}_$p.log &
done
wait
objs=$(ls build/obj/*.o | grep -v "kernels_[012]\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ab/lib_$name.so $objs build/ab/kernels_${name}_0.o build/ab/kernels_${name}_1.o build/ab/kernels_${name}_2.o -lz -ldl -lpthread
ls -la ab/lib_$name.so
