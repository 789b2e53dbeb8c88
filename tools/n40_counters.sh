export TMPDIR=/tmp MPCGPU_ORDER=as_given; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt40 -o kt40 -- python3 /root/repo/tools/prof_solve.py 16384 3 8 40 > /tmp/kt40.log 2>&1
f=$(find /tmp/kt40 -name "*kernel_stats.csv" | head -1); cp $f /root/repo/gpurun_out/kt40_kernel_stats.csv; head -3 $f | cut -c1-200
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm40 -o pm40 -- python3 /root/repo/tools/valu_per_eval.py 49152 8 1 40 > /tmp/pm40.log 2>&1
grep EVALS /tmp/pm40.log
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(float)
for f in glob.glob("/tmp/pm40/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
simdq=acc["GRBM_GUI_ACTIVE"]/8/4*1024
print({k:f"{v:.4g}" for k,v in acc.items()})
print("N_hor 40, B 49152: VALU busy %.3f, resident waves/SIMD %.2f, VALU instructions per solve %.3g" % (acc["SQ_ACTIVE_INST_VALU"]/simdq, acc["SQ_WAVE_CYCLES"]/simdq, acc["SQ_INSTS_VALU"]/acc["SQ_WAVES"]))
PY
