"""BurstGatherer (host/BurstGatherer.cpp) under ThreadSanitizer and AddressSanitizer, without a GPU.

The gather stage is the one concurrent component of the host side.  It is compiled here against a CPU stand-in of the
trxhip_hostpipe_* C ABI (tests/gatherer_stub/hostpipe_stub.cpp: results echoed back after a random delay) and driven by
tests/gatherer_stub/gatherer_stress.cpp: one producer and one consumer thread per channel, 16 channels on this
container's 8 CPUs, with the reservation path stalled now and then for longer than a batch round trip -- the window of
the stale-reservation race of round 2 (a producer that read `filling`, slept through that batch's whole flight and then
reserved a slot of the parked batch: bursts of one channel delivered out of order).

Contract checked (radioInterface.cpp:272-291, Transceiver.cpp:1229-1253): every accepted burst is delivered exactly
once, to its own channel, in push order; refused pushes == the gatherer's drop counter; per-channel TRXD header
versions; EDGE slots refused when 444-bit rows are not configured (they used to overflow the ring entry); stop() with
gathered-but-unsubmitted bursts, then start() again: one hostpipe alive, empty FIFOs, a full second run.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "gatherer_stub")
HOST = os.path.join(ROOT, "osmo_trx_amd", "host")
SRCS = [os.path.join(STUB, "gatherer_stress.cpp"), os.path.join(STUB, "hostpipe_stub.cpp"), os.path.join(HOST, "BurstGatherer.cpp")]
INC = ["-include", os.path.join(STUB, "stall_decl.h"), "-DTRX_GATHERER_TEST_STALL=gatherer_test_stall",
       "-I", os.path.join(HOST, "compat"), "-I", HOST, "-I", os.path.join(ROOT, "include")]


def build(tmp_path, name, san):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + san + INC + ["-o", exe] + SRCS,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def run(exe, *args, timeout=600):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                       timeout=timeout)
    return r


@pytest.mark.timeout(900)
def test_gatherer_order_and_exactly_once_under_tsan(tmp_path):
    exe = build(tmp_path, "gstress_tsan", ["-fsanitize=thread"])
    # 16 channels x 65536 pushes = 1,048,576 bursts (+ a restarted run of a quarter of that), float soft bits
    r = run(exe, 16, 65536, 64, 50, -1, 1)
    assert "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
    assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout[-4000:]
    acc = int(r.stdout.split("accepted ")[1].split()[0])
    assert acc > 900000                                            # the ordering check looked at ~1 M deliveries
    # TRXD mode with a different header version per channel (mVersionTRXD[chan]), small batches, short timeout
    r = run(exe, 16, 8192, 16, 20, 1, 0)
    assert "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
    assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout[-4000:]
    # bursts by reference (BurstGathererConfig::by_reference): producers push addresses into a registered ring of fifo_depth + 1
    # burst periods per channel; a ring position is rewritten only after its previous burst was pulled (TSan checks the
    # happens-before), an address outside the ring fails its batch with -EIO and nothing else
    r = run(exe, 16, 32768, 64, 50, -1, 1, 0, 0, 1)
    assert "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
    assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout[-4000:]


@pytest.mark.timeout(900)
def test_multi_device_dispatch_and_restart_under_fire_tsan(tmp_path):
    """The C++ multi-device dispatcher (BurstGathererConfig::devices[]): two and five fake devices under ThreadSanitizer --
    per-channel order and exactly-once delivery as on one device, batches dealt round-robin (every device within one batch of
    the others), contexts and pipes released; then stop() / start() called 40 times WHILE 8 producers and 8 consumers are
    inside push() / pull() (the advisor's use-after-free window of round 3)."""
    exe = build(tmp_path, "gstress_tsan_md", ["-fsanitize=thread"])
    # (the last two: three completion threads on one pipe, then on two devices with bursts by reference and restarts under fire)
    for args in ((16, 16384, 64, 50, -1, 1, 2, 0), (8, 8192, 32, 30, 1, 0, 3, 1, 1), (8, 8192, 32, 30, 1, 0, 5, 1),
                 (16, 32768, 64, 50, -1, 1, 0, 0, 0, 3), (8, 8192, 32, 30, 1, 0, 2, 1, 1, 3)):
        r = run(exe, *args)
        assert "ThreadSanitizer" not in r.stdout, r.stdout[-4000:]
        assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout[-4000:]
        assert args[6] == 0 or f"devices {args[6]} batches_per_device" in r.stdout
    assert "churn pushed" in r.stdout


@pytest.mark.timeout(600)
def test_gatherer_under_asan(tmp_path):
    exe = build(tmp_path, "gstress_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    for args in ((8, 20000, 64, 50, -1, 1), (8, 20000, 128, 100, 0, 1), (8, 8000, 64, 50, -1, 1, 3, 1), (8, 8000, 64, 50, -1, 1, 2, 1, 1)):
        r = run(exe, *args)
        assert "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-4000:]
        assert r.returncode == 0 and "errors 0" in r.stdout, r.stdout[-4000:]
