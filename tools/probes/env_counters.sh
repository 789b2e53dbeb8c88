# Probe: where the environment-step kernel's cycles go (SQ counters, two passes).  usage (GPU box): bash tools/probes/env_counters.sh
export TMPDIR=/tmp; REPO="$(cd "$(dirname "$0")/../.." && pwd)"; cd /tmp
i=0
for SET in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /tmp/envc/s$i -o pmc -- python3 "$REPO/tools/bench_env.py" --steps 20 --warmup 2 --cpu-seconds 0 > /tmp/envc_s$i.log 2>&1
  echo "set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/envc/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{k:28s} {acc[k]:.4g}  over {n[k]} launches")
PY
tail -2 /tmp/envc_s1.log | cut -c1-400
