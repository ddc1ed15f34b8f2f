#!/usr/bin/env python3
"""Host-to-host throughput of the gather stage: `chans` producer threads push bursts from host memory into
BurstGatherer, `chans` consumer threads pull the indications / TRXD datagrams back (sigproc_selftest gather).
   python tools/bench_gather.py [chans] [max_batch] [timeout_us] [trxd_version] [repeat] [producers] [fifo_depth] [by_ref]
by_ref = 1: the capture is registered as the receive ring and the producers push addresses (BurstGathererConfig::by_reference)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from osmo_trx_amd import synth, build
build.build_all()
chans = int(sys.argv[1]) if len(sys.argv) > 1 else 16
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
timeout_us = int(sys.argv[3]) if len(sys.argv) > 3 else 200
version = int(sys.argv[4]) if len(sys.argv) > 4 else 1
repeat = int(sys.argv[5]) if len(sys.argv) > 5 else 64
producers = int(sys.argv[6]) if len(sys.argv) > 6 else 4
fifo_depth = int(sys.argv[7]) if len(sys.argv) > 7 else 32
by_ref = int(sys.argv[8]) if len(sys.argv) > 8 else 0
n = 16384
iq, params = synth.make_mixed_bursts(n, "cpu")
exe = os.path.join(ROOT, "oracle", "_ref", "sigproc_selftest_abi")
if not os.path.exists(exe):
    exe = os.path.join(ROOT, "osmo_trx_amd", "lib", "sigproc_selftest")
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "iq.s16"), "wb").write(iq.numpy().tobytes())
    open(os.path.join(d, "p.bin"), "wb").write(params.tobytes())
    out = subprocess.run([exe, "gather", os.path.join(d, "iq.s16"), os.path.join(d, "p.bin"), str(n), str(chans), str(max_batch),
                          str(timeout_us), str(version), os.path.join(d, "o.bin"), str(repeat), str(producers), str(fifo_depth), str(by_ref)], stdout=subprocess.PIPE, text=True)
    print(f"{os.path.basename(exe)} chans {chans} max_batch {max_batch} timeout_us {timeout_us} trxd v{version}: {out.stdout.strip()}")
