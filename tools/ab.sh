#!/bin/bash
# usage: tools/ab.sh lib1.so lib2.so ...  -> kernel ms for B=8192 with each library (2 reps)
for lib in "$@"; do
  echo "== $lib"
  MPCGPU_LIB=$PWD/$lib python tools/prof_solve.py ${ABB:-8192} 2 2>&1 | grep solve_ms
done
