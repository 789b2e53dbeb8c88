#!/bin/bash
# VALU instruction counts of the solve kernel in both lane layouts (one / two problems per wavefront), same batch.
# usage (GPU box): tools/pmc_pairing.sh [B = 8192] [outdir = gpurun_out/pmc_pairing]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
B="${1:-8192}"; OUT="${2:-$REPO/gpurun_out/pmc_pairing}"
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for P in 1 2; do
  export MPCGPU_PAIRING=$P
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d "$OUT/p$P" -o pmc -- python3 "$REPO/tools/valu_per_eval.py" "$B" 8 1 > "$OUT/p$P.log" 2>&1
  echo "pairing $P rc=$?"; grep EVALS "$OUT/p$P.log"
  find "$OUT/p$P" -name "*counter_collection.csv" | head -1 | xargs -I{} python3 - {} <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "solve_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in acc.items()})
PY
done
