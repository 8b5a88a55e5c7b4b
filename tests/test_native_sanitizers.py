"""Host logic of the PETSc-API mirror under AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer (host code only, planning-only
contexts: the GPU pool offers no device sanitizers).  tests/native/asan_driver.cpp assembles MPIAIJ matrices on 1 / 3 / 4 thread-ranks
through relocating rows and the stash, checks the plans, round-trips PetscBinary files, updates values, hits error paths."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mirror_under_sanitizers(tmp_path):
    exe = str(tmp_path / "asan_driver")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan",
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "native", "asan_driver.cpp"), os.path.join(ROOT, "petsc_amd", "csrc", "host", "petsc_host.cpp"),
           "-L" + os.path.join(ROOT, "petsc_amd", "lib"), "-lpetschip", "-Wl,-rpath," + os.path.join(ROOT, "petsc_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib",
           "-lpthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("libasan" in b.stderr or "libubsan" in b.stderr):
        pytest.skip("sanitizer runtimes not installed: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "asan driver ok" in r.stdout, (r.stdout + r.stderr)[-4000:]


def test_host_ring_under_tsan(tmp_path):
    """The shared-memory reduction ring (phip_comm_allreduce_host) under ThreadSanitizer: four thread-ranks, 40 000 rounds."""
    lib = str(tmp_path / "libpetschip_tsan.so")
    srcs = [os.path.join(ROOT, "petsc_amd", "csrc", f) for f in ("phip_runtime.hip", "phip_spmv.hip", "phip_vec.hip", "phip_comm.hip")]
    b = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-w", "-Xarch_host", "-fsanitize=thread",
                        "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), *srcs, "-o", lib, "-ldl"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-3000:]
    exe = str(tmp_path / "tsan_ring")
    b = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-g", "-O1", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "tsan_ring.cpp"), lib, "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "tsan" in b.stderr:
        pytest.skip("ThreadSanitizer runtime not available: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "tsan ring ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-4000:]
