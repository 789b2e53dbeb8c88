#!/bin/bash
# Collects, on the GPU box, the raw rocprofv3 outputs behind bench.py's roofline object -- every pass on the SAME
# workload (bench.py defaults: N_hor = 20, 8 dynamic obstacles, B = 131072), one rocprofv3 run per pass (kernel trace and
# counters are never combined; FETCH_SIZE and WRITE_SIZE need a pass each).  Copy the directory to profiles/raw_r06/ and
# run `python tools/roofline.py rebuild`.
#   usage: tools/collect_profiles.sh [outdir = gpurun_out/raw_r06] [extra bench.py arguments]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$REPO/gpurun_out/raw_r06}"; shift || true
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-convergent --no-sweep --no-closed-loop --no-host-boundary --no-pmc $*"
python3 - "$OUT" $ARGS <<'PY'
import json, sys
sys.path.insert(0, sys.argv[2].rsplit("/", 1)[0])
import bench
a = bench.parse_args(sys.argv[3:])
json.dump({"N_hor": a.horizon, "n_dyn": a.n_dyn, "batch_per_gpu": a.batch, "steps": a.steps, "warmup": a.warmup,
           "command": "python3 bench.py " + " ".join(sys.argv[3:])}, open(sys.argv[1] + "/workload.json", "w"), indent=1)
PY
run() { name=$1; shift; rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run kt --kernel-trace --stats
run pmc_sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
# flatten: the CSVs tools/roofline.py reads
for n in kt pmc_sq pmc_fetch pmc_write; do
  find "$OUT/$n" -name "*.csv" | while read f; do cp "$f" "$OUT/$(basename "$f")"; done
done
ls -la "$OUT"
