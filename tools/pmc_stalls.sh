#!/bin/bash
# Where the solve kernel's cycles go: LDS pipe, waits, busy cycles (two counter passes on the same batch).
# usage (GPU box): tools/pmc_stalls.sh [B = 8192] [outdir = gpurun_out/pmc_stalls]
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
B="${1:-8192}"; OUT="${2:-$REPO/gpurun_out/pmc_stalls}"
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for SET in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/s$i" -o pmc -- python3 "$REPO/tools/valu_per_eval.py" "$B" 8 1 ${PMC_N:-20} > "$OUT/s$i.log" 2>&1
  echo "set $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc): print(f"{k:28s} {acc[k]:.4g}")
PY
