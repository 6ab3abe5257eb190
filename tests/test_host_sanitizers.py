"""Host-side code (partition builder, graph.<id>.bin IO, halo plan, input readers) under AddressSanitizer + UBSan:
builds dorylus_amd/host/{partition,formats}.cpp with tests/host_asan_main.cpp and runs the driver (SURVEY.md 5:
the reference has ASan in Debug builds only)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-pthread", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "dorylus_amd", "host", "partition.cpp"), os.path.join(ROOT, "dorylus_amd", "host", "formats.cpp"),
           os.path.join(ROOT, "tests", "host_asan_main.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", DORY_BUILD_THREADS="3")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "host sanitizer run ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_parallel_partition_builder_under_tsan(tmp_path):
    """the vertex-ownership parallel passes of the builder (std::thread) under ThreadSanitizer"""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "dorylus_amd", "host", "partition.cpp"), os.path.join(ROOT, "dorylus_amd", "host", "formats.cpp"),
           os.path.join(ROOT, "tests", "host_asan_main.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, DORY_BUILD_THREADS="4", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot run in this container's address-space layout")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "host sanitizer run ok" in r.stdout and "ThreadSanitizer" not in r.stderr
