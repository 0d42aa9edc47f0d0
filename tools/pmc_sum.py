import csv, glob, sys, collections
tag = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "advect_lds"
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob(f"gpurun_out/{tag}/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:.6g}  (rows {n[k]})")
