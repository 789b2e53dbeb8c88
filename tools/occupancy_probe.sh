export TMPDIR=/tmp; cd /tmp
for N in ${OCC_N:-40 20}; do
B=${OCC_B:-65536}
rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/occ$N -o occ -- python3 /root/repo/tools/valu_per_eval.py $B 8 1 $N > /tmp/occ$N.log 2>&1
python3 - $N <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(float)
for f in glob.glob(f"/tmp/occ{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); lds=r["LDS_Block_Size"]; vg=r["VGPR_Count"]; name=r["Kernel_Name"][:60]
print(sys.argv[1], name, "LDS", lds, "VGPR", vg, dict(acc), "avg resident waves/SIMD over the launch:", acc["SQ_WAVE_CYCLES"]*4/(acc["GRBM_GUI_ACTIVE"]/8*1024))
PY
done
