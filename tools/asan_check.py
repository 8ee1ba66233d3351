#!/usr/bin/env python3
"""Host-side sanitizer run (CPU only -- GPU AddressSanitizer is not available on this pool).

Builds, with gcc/g++ -fsanitize=address,undefined, into build/asan/:
  liboracle_port.so          oracle/tinympc_oracle.c            (the C restatement, 700 lines of raw pointers)
  libtinympc_emit.so         csrc/tinympc_codegen.hip as C++    (the host-only emitter of the C ABI, tinympc_codegen_emit)
  libtinympc_matlab_mock.so  the MEX shim + the mock MEX API    (argument validation / dispatch; linked against the product)
then runs the CPU tests that drive them -- oracle vs golden fixtures and vs the reference core, the emitter half of
test_codegen.py, the argument-validation half of test_mex_shim.py, the adaptive-rho oracle tests -- with the sanitizer run-time
preloaded into the (uninstrumented) python interpreter. Any report fails the run.
    python tools/asan_check.py [> profiles/r03_asan.txt]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "asan")
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g", "-O1"]


def run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def build():
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tinympc-matlab_amd", "csrc")]
    run(["gcc"] + SAN + ["-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", os.path.join(OUT, "liboracle_port.so"),
                         os.path.join(ROOT, "oracle", "tinympc_oracle.c"), "-lm"])
    run(["g++"] + SAN + ["-std=c++17", "-fPIC", "-shared", "-Wall"] + inc + ["-x", "c++", os.path.join(ROOT, "tinympc-matlab_amd", "csrc", "tinympc_codegen.hip"),
                         os.path.join(ROOT, "tools", "asan_host_stubs.cpp"), "-o", os.path.join(OUT, "libtinympc_emit.so")])
    mock = os.path.join(ROOT, "tests", "mock_mex")
    pkg = os.path.join(ROOT, "tinympc-matlab_amd")
    run(["g++"] + SAN + ["-std=c++17", "-fPIC", "-shared", "-I" + mock] + inc +
        [os.path.join(pkg, "matlab", "tinympc_matlab_mex.cpp"), os.path.join(mock, "mock_mex.cpp"), "-L" + pkg, "-ltinympc_hip",
         "-Wl,-rpath," + pkg, "-o", os.path.join(OUT, "libtinympc_matlab_mock.so")])


def main() -> int:
    if not shutil.which("gcc") or not shutil.which("g++"):
        print("gcc / g++ not available")
        return 2
    build()
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ,
               LD_PRELOAD=libasan + ":" + libubsan,
               # python itself is not instrumented: its arenas look like leaks; everything else is on
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=66:detect_odr_violation=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=67",
               TINYMPC_ORACLE_PORT_LIB=os.path.join(OUT, "liboracle_port.so"),
               TINYMPC_ASAN_EMIT_LIB=os.path.join(OUT, "libtinympc_emit.so"),
               TINYMPC_MOCK_MEX_LIB=os.path.join(OUT, "libtinympc_matlab_mock.so"))
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_vs_ref.py", "tests/test_adaptive_rho.py",
             "tests/test_codegen.py::test_emitter_matches_the_reference_emitter_line_by_line", "tests/test_codegen.py::test_emitter_sensitivity_block_and_errors",
             "tests/test_mex_shim.py", "tests/test_distributed_cpu.py::test_summary_without_process_group"]
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests
    print("+", " ".join(cmd), "   [LD_PRELOAD=%s]" % env["LD_PRELOAD"], flush=True)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    print(r.stdout[-6000:])
    reports = [l for l in (r.stdout + r.stderr).splitlines() if "ERROR: AddressSanitizer" in l or "runtime error:" in l]
    if r.stderr.strip():
        print("---- stderr ----")
        print(r.stderr[-4000:])
    print("sanitizer reports: %d" % len(reports))
    for l in reports[:20]:
        print("  ", l)
    ok = r.returncode == 0 and not reports
    print("RESULT:", "clean" if ok else "FAILED (rc %d)" % r.returncode)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
