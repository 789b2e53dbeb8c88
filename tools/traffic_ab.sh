#!/bin/bash
# L2 <-> fabric traffic and kernel time of the benchmark batch with 12 and with 16 resident wavefronts per CU (the 148-VGPR
# build forced by build_ab/three_waves.so = -DMPC_TRY_FOUR_WAVES=0, and the product library).  One rocprofv3 pass per counter.
#   build first (build container):  cd trajtrack_mpcndqn_rlboost_amd/csrc && make OUT=../../build_ab/three_waves.so OBJ=../../build_obj_w12 EXTRA=-DMPC_TRY_FOUR_WAVES=0
#   usage (GPU box): tools/traffic_ab.sh [outdir = gpurun_out/traffic_ab]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$REPO/gpurun_out/traffic_ab}"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-convergent"
for V in w16:$REPO/trajtrack_mpcndqn_rlboost_amd/libmpcgpu.so w12:$REPO/build_ab/three_waves.so; do
  name=${V%%:*}; export MPCGPU_LIB=${V#*:}
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/$name-$C" -o pmc -- python3 $ARGS > "$OUT/$name-$C.log" 2>&1
    echo "$name $C rc=$?"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name-kt" -o kt -- python3 $ARGS > "$OUT/$name-kt.log" 2>&1
  echo "$name kernel trace rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for name in ("w12", "w16"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"{out}/{name}-*SIZE/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    ms, calls = float("nan"), 0
    for f in glob.glob(f"{out}/{name}-kt/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Name"]:
                ms, calls = float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    launches = max(calls, 1)
    print(f"{name}: kernel {ms:.1f} ms per launch ({calls} launches); per launch FETCH_SIZE {acc['FETCH_SIZE'] / launches * 1024 / 1e9:.2f} GB, "
          f"WRITE_SIZE {acc['WRITE_SIZE'] / launches * 1024 / 1e9:.2f} GB (raw KiB sums {acc['FETCH_SIZE']:.4g} / {acc['WRITE_SIZE']:.4g}, counter rows {dict(n)})")
PY
