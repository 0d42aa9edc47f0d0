"""The library's HOST side under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.

Every translation unit of csrc/ is compiled host-only (hipcc --cuda-host-only: the kernels become launch stubs) with
-fsanitize=address,undefined and linked against tests/c/fake_hip.c, a recording stand-in for the HIP runtime whose device
memory is malloc'd host memory and whose kernel launches do nothing.  tests/c/host_orchestration.cpp then drives the
one-call routes (lc_lcs_host, lc_lcs_global_host with the truncation-operator cache), lc_advect_ex's launcher and the
audit counters through their plain paths, every refusal, and an injected failure at EVERY allocation and EVERY copy:
no leak, no use after free, no overrun, no crash.  No GPU sanitizer is involved (the pool offers none)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lagrangiancoherence_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
UNITS = ["api", "pack", "advect", "sigma", "ridges", "halo", "preprocess"]
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
TSAN = ["-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-O1"]


def _build_and_run(tmp_path, san, env_extra, timeout=900):
    objs = []
    procs = []
    for u in UNITS:     # host-only: seconds per file (no device code generation)
        o = str(tmp_path / f"{u}.o")
        procs.append(subprocess.Popen([HIPCC, "--cuda-host-only", "-std=c++17", "-fPIC", "-Wno-unused-function", *san,
                                       '-DLCS_BUILD_ID="san"', "-c", os.path.join(CSRC, u + ".hip"), "-o", o],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        objs.append(o)
    for p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out[-3000:]
    fake = str(tmp_path / "fake_hip.o")
    subprocess.run([HIPCC, "-x", "c", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *san, "-c",
                    os.path.join(ROOT, "tests", "c", "fake_hip.c"), "-o", fake], check=True)
    clangxx = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clangxx):
        clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    drv = str(tmp_path / "host_orchestration.o")
    subprocess.run([clangxx, "-std=c++17", "-Wall", "-Wextra", *san, "-c", os.path.join(ROOT, "tests", "c", "host_orchestration.cpp"), "-o", drv],
                   check=True)
    exe = str(tmp_path / "host_orchestration")
    # (each host object refers to its own __hip_fatbin_<hash>, which only a device link defines: left unresolved, never read --
    #  the stand-in's __hipRegisterFatBinary ignores its argument)
    r = subprocess.run([clangxx, *san, drv, *objs, fake, "-o", exe, "-ldl", "-lm", "-lpthread", "-Wl,--unresolved-symbols=ignore-all"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert r.stdout.startswith("OK ") and int(r.stdout.split()[1]) > 150, r.stdout
    return r


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_host_orchestration_under_asan_and_ubsan(tmp_path):
    r = _build_and_run(tmp_path, SAN, dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
                                           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_host_orchestration_under_thread_sanitizer(tmp_path):
    """Round 6 put threads into the host side (csrc/hostxfer.h: the staging ring's copy workers, the thread that touches the
    caller's output pages while the upload runs).  The same driver under ThreadSanitizer (race detection): every
    route, the staged copies of all sizes, every injected failure -- no data race between the workers, the calling thread and
    the background thread, no lock-order inversion, no thread leaked past lc_ctx_destroy."""
    r = _build_and_run(tmp_path, TSAN, dict(TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1:report_thread_leaks=1"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
