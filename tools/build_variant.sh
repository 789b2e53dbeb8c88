#!/bin/bash
# usage: tools/build_variant.sh name [-Dflags ...]  -> build_ab/libmpcgpu_<name>.so (solver object rebuilt with the flags, the
# environment / tracker objects of the product build reused).  For A/B runs with tools/quick_ab.py (MPCGPU_LIB=...).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab/obj_$name
H=/opt/rocm/bin/hipcc
[ -f build_obj/envgpu.o ] && [ -f build_obj/trackgpu.o ] || make -s -C trajtrack_mpcndqn_rlboost_amd/csrc
$H -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-variable -ffp-contract=on -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp \
   "$@" -c -o build_ab/obj_$name/mpcgpu.o trajtrack_mpcndqn_rlboost_amd/csrc/mpcgpu.hip
$H --offload-arch=gfx950 -shared -o build_ab/libmpcgpu_$name.so build_ab/obj_$name/mpcgpu.o build_obj/envgpu.o build_obj/trackgpu.o
echo built build_ab/libmpcgpu_$name.so
