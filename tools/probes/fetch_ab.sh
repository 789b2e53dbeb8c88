export TMPDIR=/tmp; cd /tmp
for lib in prev new; do
  if [ $lib = prev ]; then export MPCGPU_LIB=/root/repo/build_ab/libmpcgpu_prev.so; else unset MPCGPU_LIB; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$lib$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$lib$c -o pm -- python3 /root/repo/tools/prof_solve.py 32768 2 8 20 > /tmp/pm_$lib$c.log 2>&1
    python3 - $lib $c <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(f"/tmp/pm_{sys.argv[1]}{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "pair" if "solve_kernel_pair" in r["Kernel_Name"] else ("team" if "solve_kernel_team" in r["Kernel_Name"] else None)
        if k: acc[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print(sys.argv[1], sys.argv[2], {k: f"{v/len(n[k])*1024/1e9:.1f} GB per launch ({len(n[k])} launches)" for k,v in acc.items()})
PY
  done
done
