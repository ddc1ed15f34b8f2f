"""GPU tests of the C++ host shim (osmo_trx_amd/host): the reference's sigProcLib.h signatures
(detectAnyBurst / demodAnyBurst / energyDetect / vectorSlicer) and the batched pullRadioVector core,
driven from C++ exactly as the reference's callers drive them, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "osmo_trx_amd", "lib", "sigproc_selftest")
# the same source compiled against the reference's unmodified headers + its signalVector.cpp, linked to the ABI-true
# libtrxsigproc.so (built in the container where /root/reference exists; travels prebuilt in oracle/_ref/)
ABI_EXE = os.path.join(ROOT, "oracle", "_ref", "sigproc_selftest_abi")


@pytest.fixture(scope="module", params=["standalone", "reference_abi"])
def exe(request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from osmo_trx_amd import build as trx_build
    trx_build.build_all()
    if request.param == "reference_abi":
        if not os.path.exists(ABI_EXE):
            pytest.skip("oracle/_ref/sigproc_selftest_abi was not built (no /root/reference at build time)")
        return ABI_EXE
    assert os.path.exists(EXE)
    return EXE


def test_captured_burst_through_sigproclib_api(exe, golden_dir, tmp_path):
    """burst-gen.cpp:274-290 with the shim's detectAnyBurst()/demodAnyBurst()."""
    out = tmp_path / "cap.txt"
    subprocess.check_call([exe, "capture", os.path.join(golden_dir, "nb_chunk_tsc7.cfile"), str(out)])
    kv = dict(line.split(" ", 1) for line in out.read_text().strip().splitlines())
    assert int(kv["rc"]) == O.TSC
    assert np.float32(float(kv["toa"])) == np.float32(12.535156)
    ar, ai = (float(v) for v in kv["amp"].split())
    assert abs(ar + 0.00112989) < 1e-8 and abs(ai - 0.00166411) < 1e-8
    assert abs(float(kv["ci"]) - 6.460016) < 2e-5
    assert int(kv["nsoft"]) == 156
    bits = np.fromfile(os.path.join(golden_dir, "demodbits_tsc7.s8"), dtype=np.int8)
    assert kv["bits"] == "".join("1" if b > 0 else "0" for b in bits)
    assert int(kv["demod_alone_identical"]) == 1          # demodAnyBurst() on its own == fused result
    x = np.fromfile(os.path.join(golden_dir, "nb_chunk_tsc7.cfile"), dtype=np.complex64)
    e = O.lib().orc_energy_detect(x.ctypes.data, len(x), 80)
    assert abs(float(kv["energy"]) - e) <= 1e-6 * e
    rc, ebp = O.detect_any_burst(x, 7, 4.0, 4, O.TSC, 40)
    soft = O.demod_any_burst(x, rc, 4, ebp)
    sl = np.zeros(148, dtype=np.float32)
    O.lib().orc_vector_slicer(sl.ctypes.data, soft.ctypes.data, 148)
    got = [np.float32(float(v)) for v in kv["sliced0"].split()]
    assert got == [sl[0], sl[73], sl[147]]


def test_pull_radio_vector_batch(exe, tmp_path):
    from osmo_trx_amd import synth
    n = 512
    iq, params = synth.make_mixed_bursts(n, "cpu")
    params = synth.make_idle_off_mix(params)
    (tmp_path / "iq.s16").write_bytes(iq.numpy().tobytes())
    (tmp_path / "p.bin").write_bytes(params.tobytes())
    subprocess.check_call([exe, "batch", str(tmp_path / "iq.s16"), str(tmp_path / "p.bin"), str(n), "4", "625",
                           str(tmp_path / "r.bin"), str(tmp_path / "s.bin")])
    rec = np.fromfile(tmp_path / "r.bin", dtype=np.float32).reshape(n, 6)
    soft = np.fromfile(tmp_path / "s.bin", dtype=np.float32).reshape(n, 148)
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    assert np.array_equal(rec[:, 0].astype(np.int32), o_res["rc"])
    assert np.array_equal(rec[:, 4].astype(np.uint8), o_res["idle"])
    det = o_res["idle"] == 0
    assert np.array_equal(rec[det, 1], o_res["toa"][det])
    assert np.array_equal(rec[det, 5].astype(np.uint8), o_res["tsc"][det])
    O.assert_fast_ci(rec[det, 2], o_res["ci"][det])              # the batched core runs the fused kernel (FAST detector)
    on = params["type"] != O.OFF
    fin = np.isfinite(o_res["rssi"]) & on
    np.testing.assert_allclose(rec[fin, 3], o_res["rssi"][fin], rtol=1e-5, atol=1e-4)
    # the batched core uses the fused demodulator: soft bits within 1e-5 of full scale, same hard decisions
    np.testing.assert_allclose(soft[det], o_soft[det], rtol=0, atol=1e-5)
    sure = np.abs(o_soft[det] - 0.5) > 1e-4
    assert np.array_equal((soft[det] > 0.5)[sure], (o_soft[det] > 0.5)[sure])


def test_multi_arfcn_rx_class(exe, tmp_path):
    """MultiArfcnRx = the Rx half of RadioInterfaceMulti (radioInterfaceMulti.cpp:237-314) driven from C++ in chunks of
    1, 2, 3, ... blocks: logical channels 0,1,2 <- filterbank channels 1,0,3 (getLogicalChan, :87-122), each equal to
    the oracle's Channelizer + Resampler(65,48) chain on the whole stream."""
    from osmo_trx_amd import synth
    n_blocks = 45
    wide = synth.make_wideband_stream(n_blocks, "cpu")
    (tmp_path / "w.s16").write_bytes(wide.numpy().tobytes())
    subprocess.check_call([exe, "multi", str(tmp_path / "w.s16"), str(n_blocks), "3", str(tmp_path / "ch")])
    L = O.lib()
    c = L.orc_channelizer_new(4, 192, 16)
    x = wide.numpy().astype(np.float32).view(np.complex64).reshape(n_blocks, 768)
    chan = np.zeros((4, n_blocks * 192), dtype=np.complex64)
    for b in range(n_blocks):
        out = np.zeros((4, 192), dtype=np.complex64)
        blk = np.ascontiguousarray(x[b])
        L.orc_channelizer_rotate(c, blk.ctypes.data, 768, out.ctypes.data)
        chan[:, b * 192:(b + 1) * 192] = out
    L.orc_channelizer_free(c)
    r = L.orc_resampler_new(65, 48, 16, 1.0)
    for lchan, pchan in ((0, 1), (1, 0), (2, 3)):
        padded = np.concatenate([np.zeros(16, dtype=np.complex64), chan[pchan]])
        ref = np.zeros(n_blocks * 260, dtype=np.complex64)
        L.orc_resampler_rotate(r, padded[16:].ctypes.data, n_blocks * 192, ref.ctypes.data, len(ref))
        got = np.fromfile(tmp_path / f"ch{lchan}.cf32", dtype=np.complex64)
        assert np.array_equal(got.view(np.float32), ref.view(np.float32)), lchan
    L.orc_resampler_free(r)


def test_sch_through_sigproclib_api(exe, tmp_path):
    """detectSCHBurst() + demodAnyBurst(SCH) of the shim (call pattern of ms_rx_lower.cpp:213-250) vs the oracle."""
    import sch_util
    rng = np.random.default_rng(21)
    y, bits = sch_util.sch_burst(rng, 700, 9)
    (tmp_path / "sch.cfile").write_bytes(y.tobytes())
    buf, _ = sch_util.sch_burst(rng, 60000, 4 * 7000 + 2, amp=2000.0, noise=120.0)
    (tmp_path / "buf.cfile").write_bytes(buf.tobytes())
    for path, state, x in (("sch.cfile", 0, y), ("sch.cfile", 1, y), ("buf.cfile", 2, buf)):
        out = tmp_path / f"sch{state}.txt"
        subprocess.check_call([exe, "sch", str(tmp_path / path), str(state), str(out)])
        kv = dict(line.split(" ", 1) for line in out.read_text().strip().splitlines())
        rc, e = O.detect_sch_burst(x, 4.0, 4, state)
        assert int(kv["rc"]) == rc
        assert np.float32(float(kv["toa"])) == np.float32(e.toa)
        ar, ai = (float(v) for v in kv["amp"].split())
        assert abs(complex(ar, ai) - complex(e.amp[0], e.amp[1])) <= 1e-6 * max(abs(complex(e.amp[0], e.amp[1])), 1e-30)
        if rc > 0:
            assert abs(float(kv["ci"]) - e.ci) <= 2e-5
        if state == 0:
            assert rc == 1
            soft = O.demod_any_burst(y[:625], 4, 4, e)
            assert kv["bits"] == "".join("1" if v > 0 else "0" for v in soft[:148])
            assert kv["bits"] == "".join("0" if b else "1" for b in bits)      # polarity: see tests/test_oracle.py


def test_delay_and_scale_vector_api(exe, tmp_path):
    """delayVector() + scaleVector() of the shim vs the oracle, fractional / integer / negative delays."""
    rng = np.random.default_rng(22)
    x = (rng.standard_normal(625) + 1j * rng.standard_normal(625)).astype(np.complex64) * 1000
    (tmp_path / "x.cfile").write_bytes(x.tobytes())
    sc = np.complex64(0.3 - 0.8j)
    for d in (0.0, 0.005, 2.37, -3.6, 17.999, -0.5, 700.0):
        out = tmp_path / "d.cf32"
        subprocess.check_call([exe, "delay", str(tmp_path / "x.cfile"), repr(d), repr(float(sc.real)), repr(float(sc.imag)),
                               str(out)])
        got = np.fromfile(out, dtype=np.complex64)
        ref = np.zeros(625, dtype=np.complex64)
        O.lib().orc_delay_vector(x.ctypes.data, 625, d, ref.ctypes.data)
        re = ref.real * sc.real - ref.imag * sc.imag                 # Complex.h:74 operand order, fp32
        im = ref.real * sc.imag + ref.imag * sc.real
        assert np.array_equal(got.real, re.astype(np.float32)) and np.array_equal(got.imag, im.astype(np.float32)), d


def test_va_through_shim(exe, tmp_path):
    """scaleVector() + demodAnyBurst_va() of the shim (Transceiver.cpp:782-784) vs the oracle."""
    from test_oracle import _va_burst
    rng = np.random.default_rng(23)
    for tsc in (0, 5):
        y, bits = _va_burst(rng, tsc, 7, snr_db=20.0)
        (tmp_path / "va.cfile").write_bytes(y.tobytes())
        out = tmp_path / "va.f32"
        subprocess.check_call([exe, "va", str(tmp_path / "va.cfile"), str(tsc), str(out)])
        got = np.fromfile(out, dtype=np.float32)
        _, ref = O.demod_any_burst_va(y, O.TSC, tsc, 3)
        assert len(got) == 156 and np.array_equal(got, ref)
        assert int(((got[:148] > 0).astype(np.uint8) != bits).sum()) <= 2


def test_pull_radio_vector_batch_va(exe, tmp_path):
    """pullRadioVectorBatchVA(): cfg->use_va flow of pullRadioVector (Transceiver.cpp:724-803) -- power from the burst
    as read (20 samples early), detection on the shifted copy, soft bits from the Viterbi receiver -- against the same
    composition of oracle calls."""
    from osmo_trx_amd import synth
    n = 192
    # place the bursts where both views work: VA start 0..20 samples <-> detector TOA -0.4 .. 4.6 symbols
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=91, max_toa=5, delay_sym=(-4.4, 0.2), p_clip=0.0)
    params = synth.make_idle_off_mix(params)
    (tmp_path / "iq.s16").write_bytes(iq.numpy().tobytes())
    (tmp_path / "p.bin").write_bytes(params.tobytes())
    subprocess.check_call([exe, "batchva", str(tmp_path / "iq.s16"), str(tmp_path / "p.bin"), str(n),
                           str(tmp_path / "r.bin"), str(tmp_path / "s.bin")])
    rec = np.fromfile(tmp_path / "r.bin", dtype=np.float32).reshape(n, 7)
    soft = np.fromfile(tmp_path / "s.bin", dtype=np.float32).reshape(n, 148)
    x = iq.numpy().astype(np.float32).view(np.complex64).reshape(n, 625)
    ndet = 0
    for i in range(n):
        t, tsc, mt = int(params["type"][i]), int(params["tsc"][i]), int(params["max_toa"][i])
        if t == O.OFF:
            assert rec[i, 0] == 0 and rec[i, 4] == 1 and rec[i, 3] == 0
            continue
        xi = np.ascontiguousarray(x[i])
        e = O.lib().orc_energy_detect(xi.ctypes.data, 625, 80)
        assert abs(rec[i, 6] - e) <= 3e-6 * e
        assert abs(rec[i, 3] - 20.0 * np.log10(32767.0 / np.sqrt(np.float32(e)))) <= 1e-4
        if t == O.IDLE:
            assert rec[i, 4] == 1
            continue
        sh = np.zeros(625, dtype=np.complex64)
        sh[:585] = xi[20:605]
        rc, ebp = O.detect_any_burst(sh, tsc, 4.0, 4, t, mt)
        assert int(rec[i, 0]) == rc
        if rc <= 0:
            assert rec[i, 4] == 1
            continue
        ndet += 1
        assert rec[i, 1] == np.float32(ebp.toa) and abs(rec[i, 2] - ebp.ci) <= float(O.fast_ci_bar(ebp.ci)) and int(rec[i, 5]) == tsc
        _, va = O.demod_any_burst_va(xi, rc, tsc, mt)
        sl = np.zeros(148, dtype=np.float32)
        O.lib().orc_vector_slicer(sl.ctypes.data, va.ctypes.data, 148)
        assert np.array_equal(soft[i], sl), i
    assert ndet > 80                                                  # (IDLE / OFF slots, noise-only bursts excluded)
