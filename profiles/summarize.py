"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python profiles/summarize.py <run_dir> <out_prefix> [workload-json]

<run_dir> holds  kt/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE), write/ (--pmc WRITE_SIZE)
as produced by the commands in profiles/README.md.  HBM traffic per launch follows
MI355X_MICROARCH.md "HBM": bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE/WRITE_SIZE are
in KiB and on gfx950 FETCH_SIZE reports half of the bytes fetched (checked here on the pack kernel,
whose reads are a known byte count).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd.build import csrc_hash  # noqa: E402


def stamp(run):
    """What the summary is stamped with: the build id of the LIBRARY that ran (config.build_id of the run's own bench
    line, = lc_build_id()), so that bench.py replays these counters only for that binary; runs recorded before the
    library carried an id fall back to the hash of the working tree's sources."""
    try:
        return json.load(open(os.path.join(run, "bench_stdout.json")))["config"]["build_id"]
    except Exception:
        return csrc_hash()


def short(name):
    n = name.replace("void ", "")
    if "(anonymous namespace)::" in n:
        n = n.split("(anonymous namespace)::", 1)[1]
    return n.split("(")[0]


def main():
    run, out = sys.argv[1], sys.argv[2]
    workload = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    ks = glob.glob(os.path.join(run, "kt", "*", "*_kernel_stats.csv"))
    stats = {}
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(out + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "avg_ms", "min_ms", "max_ms", "percent"])
            for r in rows:
                if "anonymous namespace)::" not in r["Name"] or "at::native" in r["Name"]:
                    continue
                w.writerow([short(r["Name"]), r["Calls"], "%.4f" % (float(r["AverageNs"]) / 1e6),
                            "%.4f" % (float(r["MinNs"]) / 1e6), "%.4f" % (float(r["MaxNs"]) / 1e6), r["Percentage"]])
                stats[short(r["Name"])] = float(r["AverageNs"]) / 1e6
    pmc = collections.defaultdict(dict)
    for name in ("fetch", "write"):
        for f in glob.glob(os.path.join(run, name, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "anonymous namespace)::" in r["Kernel_Name"] and "at::native" not in r["Kernel_Name"]:
                    agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in agg.items():
                pmc[k][c + "_KiB"] = sum(v) / len(v)
    for k, d in pmc.items():
        if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024
        if k in stats:
            d["avg_ms_kernel_trace"] = stats[k]
    json.dump({"workload": workload, "csrc_hash": stamp(run), "kernels": pmc,
               "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half)"},
              open(out + "_pmc_traffic.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
