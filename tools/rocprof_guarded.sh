#!/bin/bash
# The ONE way a script of this repository starts rocprofv3: under a wall-clock limit.
#     tools/rocprof_guarded.sh <rocprofv3 arguments ...> -- python3 <program> <args ...>
# Round 5 lost metered GPU time three times to the same sequence (gpurun_out/r5e/p2.log, r5z/p2.log, r5aa/prof/p3.log):
# a --pmc set with more SQ counters than one pass has registers for -> "rocprofiler_create_counter_config ... error code 38:
# Request exceeds the capabilities of the hardware to collect" -> rocprofv3 caught signal 6 -> the child sat there until the
# lease's own limit.  So: `timeout -k 10` around rocprofv3 itself (the program after `--` stays the program: no env / bash -c
# hop, the profiler's preload initialises the GPU), limit ROCPROF_LIMIT seconds (default 240), and counter sets only
# from tools/pmc_sets.txt (tests/test_profiler_guards.py holds every set to what a pass accepted: <= 8 SQ, <= 5 TCP/TCC/TA).
# Exit status: rocprofv3's, 124 when the limit ended it.
exec timeout -k 10 "${ROCPROF_LIMIT:-240}" rocprofv3 "$@"
