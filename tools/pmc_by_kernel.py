"""Per-kernel counter sums of rocprofv3 --pmc passes:  python tools/pmc_by_kernel.py <dir with p*/ sub-directories> [kernel substring ...]
Prints, for every kernel whose name contains one of the substrings (default: all), the mean per dispatch of each counter."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
pats = sys.argv[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(f"{root}/p*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")      # "void (anonymous namespace)::kernel<...>((anonymous namespace)::Args)"
        k = (k.split("(anonymous namespace)::", 1)[1] if "(anonymous namespace)::" in k else k).split("(")[0]
        if pats and not any(p in k for p in pats):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        n = max(len(disp[k][c]), 1)
        print(f"    {c:34s} {tot[k][c] / n:14.6g} per dispatch ({n} dispatches)")
