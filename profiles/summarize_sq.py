"""Reduce the SQ / TCP / GRBM counter passes of one bench command (gpurun_out/<tag>/p*/ as written by the
--pmc commands in profiles/README.md) to <out_prefix>_pmc_sq_tcp.json, with the derived per-unit figures
DESIGN.md quotes.

    python profiles/summarize_sq.py <run_dir> <out_prefix> '<workload json>' <nsteps>

Derived (per launch of each advect kernel; W = waves, S = time steps per launch):
  valu_instr_per_wave_timestep   SQ_INSTS_VALU / (W*S)
  salu_instr_per_wave_timestep   (SQ_INSTS_SALU + SQ_INSTS_BRANCH) / (W*S)
  valu_issue_frac                SQ_ACTIVE_INST_VALU / (cycles * CUs)      -- each SIMD issues one VALU op per 4 cycles
  scalar_issue_frac              (SQ_INSTS_SALU + SQ_INSTS_BRANCH) / (cycles * CUs)   -- one scalar pipe per CU
  lds_active_frac                SQ_LDS_IDX_ACTIVE / (cycles * CUs)
  lds_bank_conflict_frac         SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  tcp_lookups_per_cu_cycle       TCP_TOTAL_CACHE_ACCESSES / (cycles * CUs)
with cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs) and CUs = 256.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd.build import csrc_hash  # noqa: E402
from bench import derived_unit_figures  # noqa: E402  (the same formulas bench.py applies to its own live passes)


def stamp(run):
    """What the summary is stamped with: the build id of the LIBRARY that ran (config.build_id of the run's own bench
    line, = lc_build_id()), so that bench.py replays these counters only for that binary; runs recorded before the
    library carried an id fall back to the hash of the working tree's sources."""
    try:
        return json.load(open(os.path.join(run, "bench_stdout.json")))["config"]["build_id"]
    except Exception:
        return csrc_hash()

CUS = 256


def short(name):
    n = name.replace("void ", "")
    if "(anonymous namespace)::" in n:
        n = n.split("(anonymous namespace)::", 1)[1]
    return n.split("(")[0]


def main():
    run, out, workload, nsteps = sys.argv[1], sys.argv[2], json.loads(sys.argv[3]), float(sys.argv[4])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(run, "p*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("advect"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, d in agg.items():
        c = {n: sum(v) / len(v) for n, v in d.items()}
        der = derived_unit_figures(c, nsteps, CUS)
        kernels[k] = {**c, "derived": der}
    json.dump({"workload": workload, "csrc_hash": stamp(run), "kernels": kernels,
               "note": "rocprofv3 --pmc passes (counter sets in profiles/README.md) over bench.py --steps 1 --warmup 0; "
                       "per-launch averages; GRBM_GUI_ACTIVE is summed over 8 XCDs"},
              open(out + "_pmc_sq_tcp.json", "w"), indent=1, sort_keys=True)
    for k, v in kernels.items():
        print(k, json.dumps(v["derived"], indent=1))


if __name__ == "__main__":
    main()
