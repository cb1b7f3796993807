"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side native code (the threaded SAH builder of the
product and the C oracle), CPU build only; the same harness checks that the builder's output is byte-identical for
1, 4 and 16 build threads, that the oracle's BVH equals its brute force, and the quantised node images
(csrc/lrc_qnodes.cpp): containment with the margin, decode, the four-wide collapse reaching the same leaves, and -- with the
kernel's box arithmetic restated -- that no box on the path to a brute-force hit fails the quantised test."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, REPO


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_builder_and_oracle_under_asan_ubsan(tmp_path):
    csrc = os.path.join(PKG, "csrc")
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    obj = tmp_path / "orc.o"
    subprocess.run(["gcc", *flags, "-ffp-contract=off", "-c", os.path.join(REPO, "oracle", "lrc_oracle.c"), "-o", str(obj)],
                   check=True, capture_output=True, text=True)
    exe = tmp_path / "harness"
    subprocess.run(["g++", "-std=c++17", *flags, "-ffp-contract=off", "-I", csrc, os.path.join(REPO, "tests", "native", "sanitize_harness.cpp"),
                    os.path.join(csrc, "bvh_build.cpp"), os.path.join(csrc, "lrc_qnodes.cpp"), str(obj), "-o", str(exe), "-pthread", "-lm"],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-3000:])
    assert r.stdout.count(" ok") == 6 and "MISMATCH" not in r.stdout and "NONDETERMINISTIC" not in r.stdout
    assert "QNODES" not in r.stdout and r.stdout.count("hit rays checked along their paths") >= 5, r.stdout[-1500:]
    assert "runtime error" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_seeded_stream_threads_under_sanitizers(tmp_path, sanitizer):
    """The streaming path of the native seeded stream (csrc/lrc_nprandom.cpp: generator thread, flag / transform workers,
    the caller walking the counts; chunk buffers recycled between them) under ThreadSanitizer and under ASan + UBSan: no
    report, and the same doubles and generator state as the sequential path for 3..6 threads, call after call."""
    csrc = os.path.join(PKG, "csrc")
    src = tmp_path / "lrc_nprandom_plain.cpp"
    with open(os.path.join(csrc, "lrc_nprandom.cpp")) as f:     # the ifunc dispatch of the clones runs before the sanitizer is up
        text = f.read().replace('__attribute__((target_clones("avx2", "default")))', "")
    src.write_text(text.replace('#include "../../include/lidarcast.h"', f'#include "{os.path.join(REPO, "include", "lidarcast.h")}"'))
    exe = tmp_path / "rng_harness"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-DLRC_TSAN", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
                    "-ffp-contract=off", str(src), os.path.join(REPO, "tests", "native", "rng_threads_harness.cpp"),
                    "-o", str(exe), "-pthread"], check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "rng threads harness ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
