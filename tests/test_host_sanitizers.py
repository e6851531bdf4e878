"""CPU: the product's HOST side under ThreadSanitizer and AddressSanitizer + UBSan (the reference builds its own
host code that way on request: /root/reference/Makefile:38-48, ASAN=yes / TSAN=yes / BSAN=yes).

`make -C roadsurf_amd tsan asan` compiles every translation unit of the library host-only (kernels become launch
stubs) with the sanitizer and links them against roadsurf_amd/sanitize/hip_stub.cpp, a device layer that never
computes (allocations are calloc, copies memcpy, launches counted).  roadsurf_amd/sanitize/harness.cpp then drives
it the way the reference driver's worker pool drives `runsimulation` (examples/example1/src/roadrunner.cpp:454-497):
64 caller threads, two groups of settings, the coalescer limited to 5 callers per batch (ROADSURF_HIP_COALESCE_MAX
< threads), short-lived threads whose caches are adopted, concurrent runsimulation_batch_ex calls of four sizes and
two concurrent rs_driver_run calls (shards, segment table, one worker thread per block).  Results mean nothing on
that device; a race, a use-after-free or an out-of-bounds access of the host code fails the test.  No GPU needed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AMD = os.path.join(ROOT, "roadsurf_amd")


def _build(kind):
    exe = os.path.join(ROOT, "tools", "bin", "host_sanitize_" + kind)
    from roadsurf_amd import provenance
    stamp = os.path.join(AMD, "build_" + kind, "sources.stamp")
    want = provenance.build_sha16()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if not os.path.exists(exe) or have != want:
        r = subprocess.run(["make", "-C", AMD, kind], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.fail("make %s failed:\n%s" % (kind, (r.stdout + r.stderr)[-3000:]))
        open(stamp, "w").write(want)
    return exe


def _run(exe, extra_env):
    env = dict(os.environ)
    env.update({"ROADSURF_HIP_COALESCE_MAX": "5", "ROADSURF_HIP_MIN_SHARD": "1024"})
    env.pop("ROADSURF_HIP_COALESCE_US", None)
    env.update(extra_env)
    r = subprocess.run([exe, "64", "640"], capture_output=True, text=True, env=env, timeout=900)
    return r.returncode, r.stdout + r.stderr


def test_host_side_is_clean_under_thread_sanitizer():
    exe = _build("tsan")
    supp = os.path.join(AMD, "sanitize", "tsan.supp")
    rc, out = _run(exe, {"TSAN_OPTIONS": "halt_on_error=0 ignore_noninstrumented_modules=1 suppressions=" + supp})
    assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
    assert rc == 0 and "sanitize harness ok" in out, out[-3000:]
    assert "phase 4: concurrent rs_driver_run calls done" in out


def test_host_side_is_clean_under_address_and_undefined_behaviour_sanitizers():
    exe = _build("asan")
    rc, out = _run(exe, {"ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]
    assert rc == 0 and "sanitize harness ok" in out, out[-3000:]
