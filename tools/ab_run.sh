#!/bin/bash
# usage: tools/ab_run.sh "B reps [N_hor [family]]" name1 name2 ...  -> tools/quick_ab.py with build_ab/libmpcgpu_<name>.so, in turn
args=$1; shift
for v in "$@"; do MPCGPU_LIB=$PWD/build_ab/libmpcgpu_$v.so python tools/quick_ab.py $args 2>&1 | grep kernel; done
