"""ctypes binding of oracle/liboracle.so -- the CPU checker (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The shipped package (osmo_trx_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

# CorrType (sigProcLib.h:30-38)
OFF, TSC, EXT_RACH, RACH, SCH, EDGE, IDLE = range(7)
SIGERR_NONE, SIGERR_BOUNDS, SIGERR_CLIP, SIGERR_UNSUPPORTED, SIGERR_INTERNAL = range(5)


class CorrSeq(C.Structure):
    _fields_ = [("n", C.c_int), ("seq", C.c_float * 128), ("gain", C.c_float * 2), ("toa", C.c_float)]


class Tables(C.Structure):
    _fields_ = [
        ("sinc_table", C.c_float * 1025),
        ("rot1", C.c_float * 314), ("rrot1", C.c_float * 314),
        ("rot4", C.c_float * 1250), ("rrot4", C.c_float * 1250),
        ("pulse1_c0", C.c_float * 4), ("pulse4_c0", C.c_float * 16), ("pulse4_c1", C.c_float * 8),
        ("c0_inv", C.c_float * 5),
        ("midamble", CorrSeq * 8), ("edge_midamble", CorrSeq * 8), ("rach", CorrSeq * 3),
        ("sch", CorrSeq), ("dummy", CorrSeq),
        ("delay_filt", (C.c_float * 20) * 64),
        ("dec_taps", C.c_float * 16),
    ]


class Ebp(C.Structure):
    _fields_ = [("amp", C.c_float * 2), ("toa", C.c_float), ("tsc", C.c_uint8), ("ci", C.c_float)]


PARAMS_DTYPE = np.dtype([("type", "u1"), ("tsc", "u1"), ("max_toa", "<u2"), ("reserved", "<u4")])
RESULT_DTYPE = np.dtype([
    ("rc", "<i4"), ("toa", "<f4"), ("amp_re", "<f4"), ("amp_im", "<f4"), ("ci", "<f4"),
    ("energy", "<f4"), ("rssi", "<f4"), ("tsc", "u1"), ("clip", "u1"), ("idle", "u1"), ("nbits_div4", "u1"),
])
assert PARAMS_DTYPE.itemsize == 8 and RESULT_DTYPE.itemsize == 32


def build(force=False):
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(ORACLE_DIR, "trx_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/Transceiver52M") and not os.path.exists(
            os.path.join(ORACLE_DIR, "_ref", "libref_generic.so")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


def header_constant(name):
    """A tolerance constant of include/trxhip.h (the statements live there and nowhere else)."""
    import re
    txt = open(os.path.join(ROOT, "include", "trxhip.h")).read()
    return float(re.search(r"#define\s+" + name + r"\s+([0-9.eE+-]+)f?\b", txt).group(1))


def fast_ci_bar(ci_ref):
    """TRXHIP_FAST_CI_ATOL_DB(ci) of include/trxhip.h -- C/I of the fused kernels' FAST detector against the reference's:
    S - C cancels by the factor 1 + C/I."""
    return 1e-4 + 1.4e-5 * (1.0 + np.power(10.0, np.asarray(ci_ref, dtype=np.float64) * 0.1))


def assert_fast_ci(ci, ci_ref):
    """C/I of a fused-kernel result against the oracle's: same NaN pattern (S < C on a noise slot), inside the bar elsewhere."""
    ci, ci_ref = np.asarray(ci, dtype=np.float64), np.asarray(ci_ref, dtype=np.float64)
    nan = np.isnan(ci_ref)
    assert np.array_equal(np.isnan(ci), nan)
    assert (np.abs(ci - ci_ref)[~nan] <= fast_ci_bar(ci_ref[~nan])).all(), float((np.abs(ci - ci_ref)[~nan] / fast_ci_bar(ci_ref[~nan])).max())


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.orc_setup.restype = C.c_int
    L.orc_get_tables.restype = C.POINTER(Tables)
    for f in (L.orc_convolve_real, L.orc_convolve_complex):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_convert_short_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_convert_float_short.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
    L.orc_energy_detect.restype = C.c_float
    L.orc_energy_detect.argtypes = [C.c_void_p, C.c_int, C.c_uint]
    L.orc_vector_slicer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_delay_vector.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.orc_detect_any_burst.restype = C.c_int
    L.orc_detect_any_burst.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_int, C.c_int, C.c_uint,
                                       C.POINTER(Ebp)]
    L.orc_detect_sch_burst.restype = C.c_int
    L.orc_detect_sch_burst.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(Ebp)]
    L.orc_demod_any_burst_va.restype = C.c_int
    L.orc_demod_any_burst_va.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.orc_va_viterbi.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p]
    L.orc_demod_any_burst.restype = C.c_int
    L.orc_demod_any_burst.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Ebp), C.c_void_p]
    L.orc_modulate_burst.restype = C.c_int
    L.orc_modulate_burst.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_modulate_edge_burst.restype = C.c_int
    L.orc_modulate_edge_burst.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_pull_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.orc_pull_batch_div.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_trxd_toa256.restype = C.c_int
    L.orc_trxd_toa256.argtypes = [C.c_double]
    L.orc_trxd_ci_cb.restype = C.c_int16
    L.orc_trxd_ci_cb.argtypes = [C.c_float]
    L.orc_trxd_soft_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.orc_trxd_pack.restype = C.c_int
    L.orc_trxd_pack.argtypes = [C.c_void_p, C.c_uint, C.c_uint32, C.c_uint8, C.c_double, C.c_double, C.c_int, C.c_int,
                                C.c_uint8, C.c_uint8, C.c_float, C.c_void_p, C.c_uint]
    L.orc_resampler_new.restype = C.c_void_p
    L.orc_resampler_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_resampler_free.argtypes = [C.c_void_p]
    L.orc_resampler_partition.restype = fp
    L.orc_resampler_partition.argtypes = [C.c_void_p, C.c_int]
    L.orc_resampler_rotate.restype = C.c_int
    L.orc_resampler_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_channelizer_new.restype = C.c_void_p
    L.orc_channelizer_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_channelizer_free.argtypes = [C.c_void_p]
    L.orc_channelizer_subfilter.restype = fp
    L.orc_channelizer_subfilter.argtypes = [C.c_void_p, C.c_int]
    L.orc_channelizer_rotate.restype = C.c_int
    L.orc_channelizer_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_setup()
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


REF_DIR = os.path.join(ORACLE_DIR, "_ref")
_ref_libs = {}


def ref_arch_available(kind):
    return os.path.exists(os.path.join(REF_DIR, f"libref_{kind}.so"))


def use_ref_arch(kind):
    """Route every FIR of the oracle's call graph -- table generation included -- through the reference's OWN arch kernels,
    compiled unmodified into oracle/_ref/libref_<kind>.so (kind = "generic" | "sse"; oracle/Makefile), or back through the
    restated loops (kind = None).  See orc_set_arch() in oracle/trx_oracle.c."""
    L = lib()
    L.orc_set_arch.argtypes = [C.c_void_p, C.c_void_p]
    if kind is None:
        L.orc_set_arch(None, None)
    else:
        if kind not in _ref_libs:
            R = C.CDLL(os.path.join(REF_DIR, f"libref_{kind}.so"))
            R.convolve_init()                                     # arch/x86/convolve.c:66-90: SSE3 function table when built with it
            R.convert_init()
            _ref_libs[kind] = R
        R = _ref_libs[kind]
        L.orc_set_arch(C.cast(R.convolve_real, C.c_void_p), C.cast(R.convolve_complex, C.c_void_p))
    L.orc_setup()


def tables():
    """Oracle tables as a dict of numpy arrays (complex tables as complex64)."""
    t = lib().orc_get_tables().contents

    def cx(a):
        return np.frombuffer(bytes(a), dtype=np.complex64).copy()

    def seq(s):
        return {"n": s.n, "seq": cx(s.seq)[: s.n], "gain": complex(s.gain[0], s.gain[1]), "toa": float(s.toa)}

    return {
        "sinc_table": np.array(t.sinc_table, dtype=np.float32),
        "rot1": cx(t.rot1), "rrot1": cx(t.rrot1), "rot4": cx(t.rot4), "rrot4": cx(t.rrot4),
        "pulse1_c0": np.array(t.pulse1_c0, dtype=np.float32),
        "pulse4_c0": np.array(t.pulse4_c0, dtype=np.float32),
        "pulse4_c1": np.array(t.pulse4_c1, dtype=np.float32),
        "c0_inv": np.array(t.c0_inv, dtype=np.float32),
        "midamble": [seq(s) for s in t.midamble],
        "edge_midamble": [seq(s) for s in t.edge_midamble],
        "rach": [seq(s) for s in t.rach],
        "sch": seq(t.sch), "dummy": seq(t.dummy),
        "delay_filt": np.array([list(r) for r in t.delay_filt], dtype=np.float32),
        "dec_taps": np.array(t.dec_taps, dtype=np.float32),
    }


def detect_any_burst(burst, tsc, threshold, sps, ctype, max_toa):
    """burst: complex64[n].  Returns (rc, dict(amp, toa, tsc, ci))."""
    b = np.ascontiguousarray(burst, dtype=np.complex64)
    e = Ebp()
    rc = lib().orc_detect_any_burst(_ptr(b), len(b), tsc, threshold, sps, ctype, max_toa, C.byref(e))
    return rc, e


SCH_DETECT_FULL, SCH_DETECT_NARROW, SCH_DETECT_BUFFER = 0, 1, 2


def detect_sch_burst(burst, threshold, sps, state):
    """detectSCHBurst (sigProcLib.cpp:1805-1861).  Returns (rc, Ebp)."""
    b = np.ascontiguousarray(burst, dtype=np.complex64)
    e = Ebp()
    rc = lib().orc_detect_sch_burst(_ptr(b), len(b), threshold, sps, state, C.byref(e))
    return rc, e


VA_SCALE = np.float32(1.0 / float((1 << 14) - 1))      # Transceiver.cpp:783


def demod_any_burst_va(burst, ctype, tsc, max_toa, scale=VA_SCALE):
    """scaleVector + demodAnyBurst_va (Transceiver.cpp:782-784, :620-645).  Returns (burst_start, soft[156])."""
    b = np.ascontiguousarray(burst, dtype=np.complex64)
    soft = np.zeros(156, dtype=np.float32)
    st = lib().orc_demod_any_burst_va(_ptr(b), len(b), ctype, tsc, max_toa, float(scale), _ptr(soft))
    return st, soft


def va_viterbi(x, rhh, start_state, stops=(4, 12)):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    r = np.ascontiguousarray(rhh, dtype=np.complex64)
    s = np.array(stops, dtype=np.uint32)
    out = np.zeros(len(x), dtype=np.float32)
    lib().orc_va_viterbi(_ptr(x), len(x), _ptr(r), start_state, _ptr(s), len(s), _ptr(out))
    return out


def demod_any_burst(burst, ctype, sps, ebp):
    b = np.ascontiguousarray(burst, dtype=np.complex64)
    soft = np.zeros(448, dtype=np.float32)
    n = lib().orc_demod_any_burst(_ptr(b), len(b), ctype, sps, C.byref(ebp), _ptr(soft))
    return soft[:max(n, 0)].copy()


def pull_batch(iq, sps, params, threshold=4.0, full_scale=32767.0, soft_stride=148, slice_bits=True):
    """iq: int16[n, burst_len, 2]; params: PARAMS_DTYPE[n]. Returns (RESULT_DTYPE[n], float32[n, soft_stride])."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    n, burst_len = iq.shape[0], iq.shape[1]
    params = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    soft = np.zeros((n, soft_stride), dtype=np.float32)
    lib().orc_pull_batch(_ptr(iq), n, burst_len, sps, _ptr(params), threshold, full_scale,
                         _ptr(res), _ptr(soft), soft_stride, 1 if slice_bits else 0)
    return res, soft


def pull_batch_div(iq_paths, sps, params, threshold=4.0, full_scale=32767.0, soft_stride=148, slice_bits=True):
    """iq_paths: int16[n, n_paths, burst_len, 2] (Transceiver.cpp:723-751). Returns (results, soft, path uint8[n])."""
    iq = np.ascontiguousarray(iq_paths, dtype=np.int16)
    n, n_paths, burst_len = iq.shape[0], iq.shape[1], iq.shape[2]
    params = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    soft = np.zeros((n, soft_stride), dtype=np.float32)
    path = np.zeros(n, dtype=np.uint8)
    lib().orc_pull_batch_div(_ptr(iq), n, n_paths, burst_len, sps, _ptr(params), C.c_float(threshold), C.c_double(full_scale),
                             _ptr(res), _ptr(soft), soft_stride, 1 if slice_bits else 0, _ptr(path))
    return res, soft, path


def modulate_burst(bits, guard, sps, empty=False):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(700, dtype=np.complex64)
    n = lib().orc_modulate_burst(_ptr(bits), len(bits), guard, sps, int(empty), _ptr(out))
    return out[:n].copy()


def convolve(x, h, start, length, complex_taps):
    """x: complex64 (may need history before `start`), h: complex64 taps."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    h = np.ascontiguousarray(h, dtype=np.complex64)
    y = np.zeros(length, dtype=np.complex64)
    f = lib().orc_convolve_complex if complex_taps else lib().orc_convolve_real
    f(_ptr(x), len(x), _ptr(h), len(h), _ptr(y), len(y), start, length)
    return y


class RxState(C.Structure):
    _fields_ = [("noises", C.c_float * 20), ("itr", C.c_size_t), ("noise_lev", C.c_float), ("muted", C.c_int),
                ("rx_empty_burst", C.c_uint), ("rx_clipping", C.c_uint), ("rx_no_burst_detected", C.c_uint)]


class UlBurstInd(C.Structure):
    _fields_ = [("rx_burst", C.c_float * 444), ("nbits", C.c_uint), ("fn", C.c_uint32), ("tn", C.c_uint8), ("rssi", C.c_double),
                ("toa", C.c_double), ("noise", C.c_double), ("idle", C.c_int), ("modulation", C.c_int), ("tss", C.c_uint8),
                ("tsc", C.c_uint8), ("ci", C.c_float)]


def pull_radio_vector_chain(iq, params, chans, muted=-1, full_scale=32767.0, rssi_offset=0.0, use_va=False):
    """Transceiver::pullRadioVector() for a schedule of slots (burst i on channel i % chans, fn = i // chans, tn = i & 7): the
    oracle's DSP per burst (energyDetect -> detectAnyBurst -> demodAnyBurst) inside the restated wrapper with one
    orc_rx_state per channel.  Returns a list of (code, UlBurstInd, rx_clipping, rx_no_burst_detected) in input order."""
    L = lib()
    L.orc_pull_radio_vector.restype = C.c_int
    L.orc_pull_radio_vector.argtypes = [C.POINTER(RxState), C.c_int, C.c_uint32, C.c_uint8, C.c_float, C.c_int, C.POINTER(Ebp),
                                        C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(UlBurstInd)]
    L.orc_rx_state_init.argtypes = [C.POINTER(RxState)]
    states = [RxState() for _ in range(chans)]
    for c, st in enumerate(states):
        L.orc_rx_state_init(C.byref(st))
        st.muted = int(c == muted)
    n = len(params)
    out = []
    soft = np.zeros(444, dtype=np.float32)
    for i in range(n):
        x = np.ascontiguousarray(iq[i].astype(np.float32))             # convert_short_float: no scaling
        typ, tsc, max_toa = int(params["type"][i]), int(params["tsc"][i]), int(params["max_toa"][i])
        ebp = Ebp()
        rc, nsoft, pw = 0, 0, 0.0
        if typ != OFF:
            pw = float(L.orc_energy_detect(x.ctypes.data, 625, 80))
            if typ != IDLE and use_va:
                # cfg->use_va (Transceiver.cpp:760-787): detection on the copy shifted by 20 samples (zeros behind), soft bits
                # from scaleVector(1 / 16383) + demodAnyBurst_va() on the burst as read
                xs = np.zeros_like(x)
                xs[:625 - 40] = x[20:625 - 20]
                rc = L.orc_detect_any_burst(xs.ctypes.data, 625, tsc, 4.0, 4, typ, max_toa, C.byref(ebp))
                if rc > 0:
                    xc = np.ascontiguousarray(x.view(np.complex64).reshape(625))
                    _, va = demod_any_burst_va(xc, rc, tsc, max_toa)
                    soft[:156] = va[:156]
                    nsoft = 156
            elif typ != IDLE:
                rc = L.orc_detect_any_burst(x.ctypes.data, 625, tsc, 4.0, 4, typ, max_toa, C.byref(ebp))
                if rc > 0:
                    nsoft = L.orc_demod_any_burst(x.ctypes.data, 625, rc, 4, C.byref(ebp), soft.ctypes.data)
        bi = UlBurstInd()
        st = states[i % chans]
        code = L.orc_pull_radio_vector(C.byref(st), typ, i // chans, i & 7, pw, rc, C.byref(ebp), soft.ctypes.data, nsoft,
                                       full_scale, rssi_offset, C.byref(bi))
        out.append((code, bi, st.rx_clipping, st.rx_no_burst_detected))
    return out


def trxd_pack_batch(res, params, soft, meta, rssi_offset=0.0, pkt_stride=160):
    """pullRadioVector()'s bi -> trxd_send_burst_ind_v0/v1 bytes for a batch (Transceiver.cpp:694-814, proto_trxd.c:68-117).
    res RESULT_DTYPE[n], params PARAMS_DTYPE[n], soft float32[n, stride] (sliced), meta {fn, tn, version, tss}[n].
    Returns (uint8[n, pkt_stride] zero padded, uint16[n] lengths); OFF slots give length 0 (-ENOENT: nothing sent)."""
    L = lib()
    n = len(res)
    pkt = np.zeros((n, pkt_stride), dtype=np.uint8)
    plen = np.zeros(n, dtype=np.uint16)
    buf = np.zeros(11 + 444 + 2, dtype=np.uint8)
    for i in range(n):
        if params["type"][i] == OFF:
            continue
        idle = bool(res["idle"][i])
        nbits = 0 if idle else 4 * int(res["nbits_div4"][i])
        row = np.ascontiguousarray(soft[i, :nbits], dtype=np.float32)
        rssi = float(np.float64(res["rssi"][i]) + np.float64(np.float32(rssi_offset)))
        # ret_idle leaves toa / tsc / ci at their initial zeros (Transceiver.cpp:694-704, :808-814)
        ln = L.orc_trxd_pack(buf.ctypes.data, int(meta["version"][i]), int(meta["fn"][i]), int(meta["tn"][i]), rssi,
                             0.0 if idle else float(res["toa"][i]), int(idle), int(nbits == 444), int(meta["tss"][i]),
                             0 if idle else int(res["tsc"][i]), 0.0 if idle else float(res["ci"][i]), row.ctypes.data, nbits)
        assert ln >= 0
        ln = min(ln, pkt_stride)
        pkt[i, :ln] = buf[:ln]
        plen[i] = ln
    return pkt, plen
