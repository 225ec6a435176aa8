"""AddressSanitizer + UBSan over the CPU-side code: oracle, host tables, host signal source.
(GPU sanitizers are not available on the pool; the device code is covered by the parity tests.)"""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpu_side_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "san_driver"
    src = [os.path.join(ROOT, "tests", "sanitize", "driver.cpp"),
           os.path.join(ROOT, "m17_sdr_amd", "csrc", "m17_txgen.cpp"),
           os.path.join(ROOT, "m17_sdr_amd", "csrc", "m17_tables.cpp")]
    obj = tmp_path / "oracle.o"
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
             "-ffp-contract=off", "-fopenmp"]
    subprocess.run(["gcc", "-std=gnu11", *flags, "-c", os.path.join(ROOT, "oracle", "m17_oracle.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++17", *flags, *src, str(obj), "-o", str(exe), "-pthread", "-lm"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
    sys.stdout.write(r.stdout[-2000:]); sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-4000:]
    assert "sanitizer driver ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
