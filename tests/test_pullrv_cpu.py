"""trxPullRadioVector() -- Transceiver::pullRadioVector(chan, struct trx_ul_burst_ind *bi) over the gatherer -- without a GPU.

host/trxPullRadioVector.cpp + host/BurstGatherer.cpp are compiled (plain, and with -fsanitize=address,undefined) against the
CPU stand-in of the host pipeline (tests/gatherer_stub/hostpipe_stub.cpp), which echoes every burst's routing stamp: the
"GPU" reports rc = the slot's type, toa = fn, energy = a stamp the driver put into the samples.  What is tested is the host
logic around the DSP (Transceiver.cpp:693-815): struct bi initialisation, -ENOENT for OFF slots without touching power or
noise, idle indications for muted channels, the 20-entry noise ring fed by IDLE slots only (radioVector.cpp:79-108), rssi and
noise in double, modulation / nbits, per-channel order -- against the oracle's restatement of the same function
(orc_pull_radio_vector, oracle/trx_oracle.c) driven with the same schedule."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "gatherer_stub")
HOST = os.path.join(ROOT, "osmo_trx_amd", "host")
SRCS = [os.path.join(STUB, "pullrv_driver.cpp"), os.path.join(STUB, "hostpipe_stub.cpp"), os.path.join(HOST, "BurstGatherer.cpp"),
        os.path.join(HOST, "trxPullRadioVector.cpp")]
INC = ["-include", os.path.join(STUB, "stall_decl.h"), "-DTRX_GATHERER_TEST_STALL=gatherer_test_stall",
       "-I", os.path.join(HOST, "compat"), "-I", HOST, "-I", os.path.join(ROOT, "include")]
REC = np.dtype([("code", "<i4"), ("nbits", "<u4"), ("fn", "<u4"), ("tn", "<u4"), ("idle", "<u4"), ("modulation", "<u4"),
                ("tss", "<u4"), ("tsc", "<u4"), ("ci", "<f4"), ("rssi", "<f8"), ("toa", "<f8"), ("noise", "<f8"), ("rx0", "<f4"),
                ("rx_clipping", "<u4"), ("rx_no_burst_detected", "<u4")])


@pytest.mark.parametrize("san", [[], ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]])
def test_pull_radio_vector_host_logic(tmp_path, san):
    exe = str(tmp_path / "pullrv")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + san + INC + ["-o", exe] + SRCS,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    n, chans, muted = 4000, 4, 1
    out = tmp_path / "rv.bin"
    r = subprocess.run([exe, str(n), str(chans), str(muted), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), timeout=600)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
    g = np.fromfile(out, dtype=REC)
    assert len(g) == n

    L = O.lib()
    L.orc_pull_radio_vector.restype = C.c_int
    L.orc_pull_radio_vector.argtypes = [C.POINTER(O.RxState), C.c_int, C.c_uint32, C.c_uint8, C.c_float, C.c_int, C.POINTER(O.Ebp),
                                        C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(O.UlBurstInd)]
    L.orc_rx_state_init.argtypes = [C.POINTER(O.RxState)]
    states = [O.RxState() for _ in range(chans)]
    for c, st in enumerate(states):
        L.orc_rx_state_init(C.byref(st))
        st.muted = int(c == muted)
    soft = np.zeros(444, dtype=np.float32)
    n_idle = n_off = 0
    for i in range(n):
        typ = O.OFF if i % 11 == 3 else O.IDLE if i % 5 == 1 else O.RACH if i % 7 == 2 else O.TSC
        fn = i // chans
        energy = float(1 + (i * 7919) % 5003)
        ebp = O.Ebp()
        ebp.toa = float(fn)                                       # the stub's echo: toa = fn, tsc = the slot's, ci = 0
        ebp.tsc = i & 7
        # the stub's soft row is (fn + k) & 1 in 0..1 form; the oracle slices -1..+1 values: feed it the unsliced equivalent
        soft[:148] = 2.0 * ((fn + np.arange(148)) & 1) - 1.0
        rc = typ if typ not in (O.OFF, O.IDLE) else 0
        bi = O.UlBurstInd()
        st = states[i % chans]
        code = L.orc_pull_radio_vector(C.byref(st), typ, fn, i & 7, energy, rc, C.byref(ebp), soft.ctypes.data, 156, 32767.0, -3.5,
                                       C.byref(bi))
        r = g[i]
        assert r["code"] == code and r["fn"] == bi.fn and r["tn"] == bi.tn, i
        assert r["nbits"] == bi.nbits and r["idle"] == bi.idle and r["modulation"] == bi.modulation and r["tss"] == 0, i
        assert r["rssi"] == bi.rssi and r["toa"] == bi.toa and r["tsc"] == bi.tsc, (i, r["rssi"], bi.rssi)
        assert (np.isinf(r["noise"]) and np.isinf(bi.noise)) or r["noise"] == bi.noise, (i, r["noise"], bi.noise)
        if code == 0 and not bi.idle:
            assert r["rx0"] == np.float32(fn & 1)
        n_idle += typ == O.IDLE
        n_off += typ == O.OFF
    assert n_idle > 20 * chans and n_off > 100                     # the noise rings wrapped several times
    assert (g["code"][np.arange(n) % 11 == 3] == -2).all()
