"""The host code (6.6 k lines of C++: bump arena, page-locked staging, transcript, decommitment planner, verifier, level-2
handles, sharded collectives, the shared twiddle registry) under AddressSanitizer + UndefinedBehaviorSanitizer and under
ThreadSanitizer ("ASan on host code").  The instrument is the emulation build of the same sources
(tests/emu/build_emu.sh asan | tsan; emu_runtime.cpp announces its fiber switches to the sanitizers): the CPU suites that load
tests/emu/libluminair_emu.so are re-run in a pytest subprocess that has the sanitizer runtime preloaded and
LMN_EMU_SANITIZER set (tests/conftest.py redirects the library path).  Any report aborts the subprocess
(halt_on_error) and fails the test with the report's tail.

    python -m pytest tests -m sanitize -x -q          (about 10 minutes on 8 CPUs)

csrc/batch.cpp (the lock-step batch library: member fibers on hand-made stacks, wait-free rendezvous, argument tables, the
batched copy list) is reached through the emulated batch build (build_emu.sh batch asan | tsan; tests/test_batch_emu.py's
scenarios as a plain script)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    """LD_PRELOAD value: the sanitizer runtime AND libstdc++ (the interpreter does not link it: without it the runtime finds
    no __cxa_throw to forward to at start-up and aborts on the library's first exception)"""
    libs = []
    for n in (name, "libstdc++.so"):
        out = subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip()
        if not out or not os.path.isabs(out) or not os.path.exists(out):
            pytest.skip("%s not available" % n)
        libs.append(os.path.realpath(out))
    return " ".join(libs)


def _build(kind):
    so = os.path.join(ROOT, "tests", "emu", "libluminair_emu_%s.so" % kind)
    csrc = os.path.join(ROOT, "luminair_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_runtime.cpp", "build_emu.sh")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh"), kind], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    # the suites also want the plain build to be current (their own fixtures check it against the sources)
    plain = os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(plain) or any(os.path.getmtime(s) > os.path.getmtime(plain) for s in srcs):
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return so


def _run(kind, runtime, options, tests, timeout, extra_env=None):
    env = dict(os.environ, LMN_EMU_SANITIZER=kind, LD_PRELOAD=runtime, PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    env.update(options)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu and not sanitize", "-p", "no:cacheprovider"] + tests
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-6000:] + "\n---- stderr ----\n" + r.stderr[-6000:])
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and \
        "WARNING: ThreadSanitizer" not in r.stderr, tail
    return r.stdout


@pytest.mark.sanitize
def test_emulation_suite_under_address_and_undefined_behaviour_sanitizers():
    rt = _runtime("libasan.so")
    _build("asan")
    # detect_leaks=0: the interpreter's own allocations; the library's handles are checked by the suites' close() paths
    opts = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0:verify_asan_link_order=0",
            "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}
    out = _run("asan", rt, opts, ["tests/test_emu_hostlogic.py", "tests/test_product_verifier.py", "tests/test_level2_only_prove.py",
                                  "tests/test_settings.py", "tests/test_producer_scenarios.py", "tests/test_emu_threads.py",
                                  "tests/test_pin_variant.py"], 3600)
    assert " passed" in out, out[-2000:]


@pytest.mark.sanitize
def test_sharded_proofs_under_address_sanitizer():
    """world 2 over gloo (one emulation context per process, the lmn_collective callbacks) and over the built-in RCCL
    transport against the stub librccl: pack / unpack kernels, halo exchange, the row-parallel front end"""
    rt = _runtime("libasan.so")
    _build("asan")
    opts = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0:verify_asan_link_order=0",
            "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}
    out = _run("asan", rt, opts, ["tests/test_sharded_prove.py", "tests/test_sharded_rccl_stub.py", "tests/test_sharded_merkle.py",
                                  "-k", "not 8 and not world8 and not eight"], 3600)
    assert " passed" in out, out[-2000:]


@pytest.mark.sanitize
def test_concurrent_contexts_under_thread_sanitizer():
    rt = _runtime("libtsan.so")
    _build("tsan")
    opts = {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1:report_signal_unsafe=0"}
    # gcc 11's runtime does not know the address-space layout of recent kernels under full randomisation: setarch -R
    cmd_prefix = ["setarch", os.uname().machine, "-R"]
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.update(opts)
    # a plain interpreter, not pytest: conftest's `import torch` brings thread pools of its own, and the runtime of gcc 11
    # does not survive them (SIGSEGV at start-up); the scenario needs numpy and ctypes only
    cmd = cmd_prefix + [sys.executable, os.path.join(ROOT, "tests", "test_emu_threads.py"),
                        os.path.join(ROOT, "tests", "emu", "libluminair_emu_tsan.so")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
    tail = (r.stdout[-6000:] + "\n---- stderr ----\n" + r.stderr[-6000:])
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, tail
    assert "concurrent contexts ok" in r.stdout, tail


def _run_script(runtime, options, script, lib, marker, timeout=3600, no_aslr=False, args=()):
    env = dict(os.environ, LD_PRELOAD=runtime, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.update(options)
    cmd = (["setarch", os.uname().machine, "-R"] if no_aslr else []) + [sys.executable, os.path.join(ROOT, "tests", script), lib] + list(args)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-6000:] + "\n---- stderr ----\n" + r.stderr[-6000:])
    assert r.returncode == 0 and marker in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and \
        "WARNING: ThreadSanitizer" not in r.stderr, tail


@pytest.mark.sanitize
def test_lock_step_batches_under_address_and_undefined_behaviour_sanitizers():
    """csrc/batch.cpp + batch.h on the emulation runtime: KAT batches with more members than worker threads, a pie that fails
    alone (constraints, non-canonical word), refused shapes, growing batches, four operators"""
    import test_batch_emu
    rt = _runtime("libasan.so")
    lib = test_batch_emu._build("asan")
    _run_script(rt, {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0:verify_asan_link_order=0",
                     "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}, "test_batch_emu.py", lib, "emulated batches ok")


@pytest.mark.sanitize
def test_lock_step_batches_under_thread_sanitizer():
    """16-row pies (KAT batches on two worker threads, a pie that fails alone): up to eight worker threads carry the member
    fibers, the last arriver of every rendezvous launches for everybody.  One finding (round 6), fixed: every member stored
    the launch's table region into the group (`g.region`, the same value from all of them)"""
    import test_batch_emu
    rt = _runtime("libtsan.so")
    lib = test_batch_emu._build("tsan")
    _run_script(rt, {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1:report_signal_unsafe=0"}, "test_batch_emu.py", lib,
                "emulated batches ok", no_aslr=True, args=("small",))
