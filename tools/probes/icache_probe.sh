export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|SQ_IFETCH\|INST_CACHE" | head -10
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/ic -o ic -- python3 /root/repo/tools/valu_per_eval.py ${IC_B:-32768} 8 1 ${IC_N:-20} > /tmp/ic.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(float)
for f in glob.glob("/tmp/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print(dict(acc))
PY
tail -3 /tmp/ic.log
