"""CPU: the blob parser (the one piece of the library that reads untrusted bytes) built with AddressSanitizer +
UndefinedBehaviorSanitizer and fed valid blobs plus a few thousand corruptions of them.  GPU sanitizers are not available
on this pool; this is the host build only (SURVEY.md §5 "Race detection / sanitizers")."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from pyfaceanalysis_amd import blob
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_blob_parser_under_asan_ubsan(tmp_path, nets):
    exe = tmp_path / "asan_blob_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "pyfaceanalysis_amd", "csrc"), os.path.join(ROOT, "tests", "asan_blob_driver.cpp"),
           os.path.join(ROOT, "pyfaceanalysis_amd", "csrc", "hg_tree.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("sanitizer build not possible here: " + r.stdout.decode(errors="replace")[-300:])
    seeds = [blob.flow_to_blob(n) for n in (helpers.overlapping_net(1), helpers.product_net(1), helpers.linear_net(1), nets("T3L-8"),
                                            helpers.fuzz_igsfa_net(1), helpers.fuzz_igsfa_net(2), helpers.fuzz_igsfa_net(10), helpers.fuzz_product_net(3))]
    rng = np.random.default_rng(0)
    files, n_valid = [], 0
    for si, good in enumerate(seeds):
        p = tmp_path / ("good%d.bin" % si)
        p.write_bytes(good)
        files.append(str(p))
        n_valid += 1
        for k in range(250):
            ba = bytearray(good)
            mode = k % 5
            if mode == 0:                                           # one 32-bit word anywhere in the structural part
                off = int(rng.integers(6, min(len(ba) // 4, 4000))) * 4
                struct.pack_into("<I", ba, off, int(rng.integers(0, 2 ** 32)))
            elif mode == 1:                                         # small values where counts and dims live
                off = int(rng.integers(6, min(len(ba) // 4, 400))) * 4
                struct.pack_into("<I", ba, off, int(rng.choice([0, 1, 2, 3, 7, 16, 255, 2 ** 24, 2 ** 31, 2 ** 32 - 1])))
            elif mode == 2:                                         # truncation (size field fixed up, so the parser walks in)
                ba = ba[:int(rng.integers(24, len(ba)))]
                ba += b"\0" * ((-len(ba)) % 8)
                struct.pack_into("<Q", ba, 16, len(ba))
            elif mode == 3:                                         # random byte noise
                for _ in range(8):
                    ba[int(rng.integers(8, len(ba)))] = int(rng.integers(0, 256))
            else:                                                   # a run of 0xFF
                off = int(rng.integers(24, len(ba) - 16))
                ba[off:off + 12] = b"\xff" * 12
            p = tmp_path / ("m%d_%d.bin" % (si, k))
            p.write_bytes(bytes(ba))
            files.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = b""
    for i in range(0, len(files), 500):
        r = subprocess.run([str(exe)] + files[i:i + 500], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
        out += r.stdout
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    parsed = sum(int(l.split()[1]) for l in out.decode().splitlines() if l.startswith("parsed"))
    rejected = sum(int(l.split()[3]) for l in out.decode().splitlines() if l.startswith("parsed"))
    assert parsed + rejected == len(files) and parsed >= n_valid and rejected > 200


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_pool_under_tsan(tmp_path):
    """The packing pool of the host path (hg_hostpool.hpp) under ThreadSanitizer: alternating 4-task and large regions back
    to back from two caller threads (a worker still leaving the previous region took a ticket of the next one)."""
    exe = tmp_path / "tsan_pool_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "pyfaceanalysis_amd", "csrc"),
           os.path.join(ROOT, "tests", "tsan_pool_driver.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("TSAN build not possible here: " + r.stdout.decode(errors="replace")[-300:])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    out = r.stdout.decode(errors="replace")
    if "FATAL: ThreadSanitizer" in out and "unexpected memory mapping" in out:
        pytest.skip("TSAN cannot run in this container (address-space layout): " + out[-200:])
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in out and "bad 0 0 thrown 1 sum 4950" in out, out[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_pipeline_plain_and_under_tsan(tmp_path):
    """The host half of hg_flow_execute (hg_hostpipe.hpp: tickets, pinned ring or direct destinations, pieces, passes, the
    fall-through from exact narrowing to the caller's type, the pass planner) with a memcpy sink in the place of the GPU
    whose copies and passes complete late: 285 seeded cases, every row's sum computed from the wire rows the sink received.
    Once as a plain build (must pass), once under ThreadSanitizer (skipped where TSAN cannot run)."""
    srcs = [os.path.join(ROOT, "tests", "tsan_pipe_driver.cpp"), os.path.join(ROOT, "pyfaceanalysis_amd", "csrc", "hg_hostpack.cpp")]
    inc = "-I" + os.path.join(ROOT, "pyfaceanalysis_amd", "csrc")
    exe = tmp_path / "pipe_driver"
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", inc] + srcs + ["-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"bad 0" in r.stdout, r.stdout.decode(errors="replace")[-2000:]
    texe = tmp_path / "pipe_driver_tsan"
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", inc] + srcs + ["-o", str(texe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if b.returncode != 0:
        pytest.skip("TSAN build not possible here: " + b.stdout.decode(errors="replace")[-300:])
    r = subprocess.run([str(texe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    out = r.stdout.decode(errors="replace")
    if "FATAL: ThreadSanitizer" in out and "unexpected memory mapping" in out:
        pytest.skip("TSAN cannot run in this container (address-space layout): " + out[-200:])
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in out and "bad 0" in out, out[-3000:]
