#!/bin/bash
# Row f3 (env_step_kernel): the rocprofv3 passes behind its roofline figures on THIS round's build -- kernel trace (average duration),
# SQ counters (dynamic VALU instruction count, VALU-busy), FETCH_SIZE and WRITE_SIZE (one pass each, never combined with a trace) over
# the same `tools/bench_env.py` run -- and the summary tools/env_roofline.py makes of them.
#   usage (GPU box): bash tools/env_counters.sh [outdir = gpurun_out/env_r06] [batch = 32768]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$REPO/gpurun_out/env_r06}"; B="${2:-32768}"
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$REPO/tools/bench_env.py --batch $B --steps 100 --warmup 10 --cpu-seconds 0"
run() { name=$1; shift; rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run kt --kernel-trace --stats
run pmc_sq --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
for n in kt pmc_sq pmc_fetch pmc_write; do
  find "$OUT/$n" -name "*.csv" | while read f; do cp "$f" "$OUT/$(basename "$f")"; done
done
python3 $ARGS > "$OUT/bench_env_line.json" 2> "$OUT/bench_env.err"
python3 "$REPO/tools/env_roofline.py" "$OUT" "$B"
