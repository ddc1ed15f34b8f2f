"""ctypes binding of libtrxhip.so (include/trxhip.h).  Plumbing only.

Device memory, streams and torch.distributed come from PyTorch; every compute call goes through the
C ABI with raw device pointers.  There is no CPU path here: if the library is not built, or no GPU is
present, construction of TrxHip raises TrxHipError.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))

# CorrType, sigProcLib.h:30-38
OFF, TSC, EXT_RACH, RACH, SCH, EDGE, IDLE = range(7)

PARAMS_DTYPE = np.dtype([("type", "u1"), ("tsc", "u1"), ("max_toa", "<u2"), ("reserved", "<u4")])
RESULT_DTYPE = np.dtype([
    ("rc", "<i4"), ("toa", "<f4"), ("amp_re", "<f4"), ("amp_im", "<f4"), ("ci", "<f4"),
    ("energy", "<f4"), ("rssi", "<f4"), ("tsc", "u1"), ("clip", "u1"), ("idle", "u1"), ("nbits_div4", "u1"),
])
TRXD_META_DTYPE = np.dtype([("fn", "<u4"), ("tn", "u1"), ("version", "u1"), ("tss", "u1"), ("reserved", "u1")])
assert PARAMS_DTYPE.itemsize == 8 and RESULT_DTYPE.itemsize == 32 and TRXD_META_DTYPE.itemsize == 8

TRXD_RECORD_BYTES = 156
FLAG_SLICE = 1          # TRXHIP_FLAG_SLICE
FLAG_EXACT_DEMOD = 2    # TRXHIP_FLAG_EXACT_DEMOD
FLAG_IDLE_DUMMY = 4     # TRXHIP_FLAG_IDLE_DUMMY
FLAG_FEW_NB_SLOTS = 16  # TRXHIP_FLAG_FEW_NB_SLOTS (a hint: see include/trxhip.h)
SCH_DETECT_FULL, SCH_DETECT_NARROW, SCH_DETECT_BUFFER = 0, 1, 2   # sch_detect_type (sigProcLib.h:139-143)


def few_nb_hint(host_params):
    """TRXHIP_FLAG_FEW_NB_SLOTS when more than 1/32 of the slots of `host_params` (PARAMS_DTYPE[n], or None: no hint) are NOT
    normal-burst slots the normal-burst kernel takes (type TSC = 1, tsc < 8, max_toa <= 32)."""
    if host_params is None or len(host_params) == 0:
        return 0
    nb = (host_params["type"] == 1) & (host_params["tsc"] < 8) & (host_params["max_toa"] <= 32)
    return FLAG_FEW_NB_SLOTS if 32 * (len(host_params) - int(nb.sum())) > len(host_params) else 0


class TrxHipError(RuntimeError):
    pass


def lib_path():
    # TRXHIP_LIB: profiling override (e.g. the -DTRX_DIAG build); default = the product library
    return os.environ.get("TRXHIP_LIB") or os.path.join(PKG, "lib", "libtrxhip.so")


_LIB = None

# every symbol include/trxhip.h declares: (name, restype, argtypes)
_VP, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "trxhip_abi_version": (_I, []),
    "trxhip_device_count": (_I, []),
    "trxhip_create": (_I, [C.POINTER(_VP), _I]),
    "trxhip_destroy": (None, [_VP]),
    "trxhip_strerror": (C.c_char_p, [_I]),
    "trxhip_set_work_pool": (_I, [_VP, _I]),
    "trxhip_set_nb_kernel": (_I, [_VP, _I]),
    "trxhip_fast_stats": (_I, [_VP, C.POINTER(C.c_uint64), _I]),
    "trxhip_tables_size": (_SZ, []),
    "trxhip_tables_generate_host": (_I, [_VP, _SZ]),
    "trxhip_create_from_tables": (_I, [C.POINTER(_VP), _I, _VP, _SZ]),
    "trxhip_tables_device_ptr": (_I, [_VP, C.POINTER(_VP)]),
    "trxhip_tables_checksum": (C.c_uint64, [_VP, _SZ]),
    "trxhip_detect_demod_batch": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _I, _I, _F, _F, _I, _I, _VP]),
    "trxhip_detect_demod_batch_cf32": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _I, _I, _F, _F, _I, _I, _VP]),
    "trxhip_demod_batch_cf32": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _SZ, _I, _I, _I, _I, _VP]),
    "trxhip_energy_detect_batch_cf32": (_I, [_VP, _VP, _SZ, _I, C.c_uint, _VP, _VP]),
    "trxhip_delay_vector_batch_cf32": (_I, [_VP, _VP, _VP, _VP, _SZ, _I, _VP]),
    "trxhip_scale_vector_cf32": (_I, [_VP, _VP, _SZ, C.c_float, C.c_float, _VP]),
    "trxhip_demod_va_batch_cf32": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _SZ, _I, C.c_float, _I, _I, _VP]),
    "trxhip_detect_sch_batch_cf32": (_I, [_VP, _VP, _VP, _SZ, _SZ, _I, _I, C.c_float, _VP]),
    "trxhip_vector_slicer": (_I, [_VP, _VP, _VP, _SZ, _VP]),
    "trxhip_pack_trxd_batch": (_I, [_VP, _VP, _VP, _I, _VP, _SZ, _F, _VP]),
    "trxhip_pack_trxd_wire_batch": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _SZ, _F, _VP]),
    "trxhip_select_diversity_batch": (_I, [_VP, _VP, _SZ, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "trxhip_apply_diversity_power": (_I, [_VP, _VP, _VP, _VP, _SZ, C.c_float, _VP]),
    "trxhip_hostpipe_create": (_I, [_VP, _VP, C.POINTER(_VP)]),
    "trxhip_hostpipe_destroy": (None, [_VP]),
    "trxhip_hostpipe_slot_buffers": (_I, [_VP, _I, _VP]),
    "trxhip_hostpipe_set_levels": (_I, [_VP, C.c_float, C.c_float, C.c_float]),
    "trxhip_hostpipe_submit": (_I, [_VP, _I, _SZ]),
    "trxhip_hostpipe_wait": (_I, [_VP, _I]),
    "trxhip_hostpipe_query": (_I, [_VP, _I]),
    "trxhip_hostpipe_run": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ]),
    "trxhip_hostpipe_register_host": (_I, [_VP, _VP, _SZ]),
    "trxhip_hostpipe_unregister_host": (_I, [_VP, _VP]),
    "trxhip_hostpipe_slot_sources": (_I, [_VP, _I, C.POINTER(_VP)]),
    "trxhip_hostpipe_submit_by_ref": (_I, [_VP, _I, _SZ]),
    "trxhip_convolve_real_batch": (_I, [_VP, _VP, _I, _VP, _I, _VP, _I, _I, _I, _SZ, _VP]),
    "trxhip_convolve_complex_batch": (_I, [_VP, _VP, _I, _VP, _I, _VP, _I, _I, _I, _SZ, _VP]),
    "trxhip_convert_short_float": (_I, [_VP, _VP, _VP, _SZ, _VP]),
    "trxhip_convert_float_short": (_I, [_VP, _VP, _VP, _F, _SZ, _VP]),
    "trxhip_dft_batch": (_I, [_VP, _VP, _VP, _I, _SZ, _SZ, _SZ, _I, _VP]),
    "trxhip_channelize_batch": (_I, [_VP, _VP, _VP, _SZ, _I, _I, _I, _VP]),
    "trxhip_resample_batch": (_I, [_VP, _VP, _VP, _SZ, _I, _I, _SZ, _SZ, _SZ, _VP]),
    "trxhip_rx_frontend_create": (_I, [_VP, _I, _I, _I, C.POINTER(_VP)]),
    "trxhip_rx_frontend_destroy": (None, [_VP]),
    "trxhip_rx_frontend_reset": (_I, [_VP, _VP]),
    "trxhip_rx_frontend_seed": (_I, [_VP, _VP, _SZ, _VP]),
    "trxhip_rx_frontend_pull": (_I, [_VP, _VP, _SZ, _VP, _SZ, _VP]),
}


def load_library():
    """dlopen libtrxhip.so and bind every symbol of include/trxhip.h.  Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise TrxHipError(f"{path} is not built (run `python -m osmo_trx_amd.build`); there is no CPU fallback")
    try:
        L = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise TrxHipError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        if (name == "trxhip_fast_stats" or name.startswith("trxhip_hostpipe_")) and os.environ.get("TRXHIP_LIB") and not hasattr(L, name):
            continue          # tools/ab_multi.sh against an older measurement build of the library (product: always bound)
        f = getattr(L, name)  # AttributeError if the export is missing
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _check(rc, what):
    if rc != 0:
        msg = load_library().trxhip_strerror(rc).decode()
        raise TrxHipError(f"{what} failed: {rc} ({msg})")


def generate_tables_host():
    """The table blob as bytes (host-only; works without a GPU)."""
    L = load_library()
    n = L.trxhip_tables_size()
    buf = (C.c_ubyte * n)()
    _check(L.trxhip_tables_generate_host(buf, n), "trxhip_tables_generate_host")
    return bytes(buf)


def tables_checksum(blob):
    L = load_library()
    b = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
    return int(L.trxhip_tables_checksum(b, len(blob)))


class TrxHip:
    """One context per GPU (owns the device-resident tables).  All tensors are torch CUDA tensors."""

    def __init__(self, device=0, tables_blob=None):
        import torch
        self.torch = torch
        self.L = load_library()
        if not torch.cuda.is_available():
            raise TrxHipError("no GPU visible: osmo_trx_amd has no CPU fallback")
        self.device = int(device)
        h = _VP()
        if tables_blob is None:
            _check(self.L.trxhip_create(C.byref(h), self.device), "trxhip_create")
        else:
            b = (C.c_ubyte * len(tables_blob)).from_buffer_copy(tables_blob)
            _check(self.L.trxhip_create_from_tables(C.byref(h), self.device, b, len(tables_blob)),
                   "trxhip_create_from_tables")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.trxhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -------------------------------------------------------------------------------
    def _stream(self, stream=None):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device)
        return _VP(stream.cuda_stream)

    def _dev(self, t, dtype=None):
        torch = self.torch
        assert t.is_cuda and t.device.index == self.device and t.is_contiguous(), "need a contiguous tensor on this GPU"
        if dtype is not None:
            assert t.dtype == dtype, (t.dtype, dtype)
        return _VP(t.data_ptr())

    def tables_device_tensor(self):
        """uint8 view-less copy target for an in-place RCCL broadcast: (ptr, nbytes)."""
        p = _VP()
        _check(self.L.trxhip_tables_device_ptr(self.h, C.byref(p)), "trxhip_tables_device_ptr")
        return p.value, int(self.L.trxhip_tables_size())

    def set_work_pool(self, enabled):
        """Cross-die work pool of the 4-SPS kernel on / off (results never depend on it; a measurement switch)."""
        _check(self.L.trxhip_set_work_pool(self.h, 1 if enabled else 0), "trxhip_set_work_pool")

    def set_nb_kernel(self, enabled):
        """Normal-burst kernel + leftover list (default) or the general kernel alone (results are bit-identical; a measurement switch)."""
        _check(self.L.trxhip_set_nb_kernel(self.h, 1 if enabled else 0), "trxhip_set_nb_kernel")

    def fast_stats(self, reset=False):
        """Counters of the fused kernels' FAST detector since the last reset: {"reruns": bursts whose TOA search was re-run in
        the reference's operand order; left_*: bursts the normal-burst kernel left to the general one (correlation guard, peak-ratio gate
        too close to call, TOA outside the straight-line demodulator's geometry)}.  Synchronises the device."""
        out = (C.c_uint64 * 4)()
        _check(self.L.trxhip_fast_stats(self.h, out, 1 if reset else 0), "trxhip_fast_stats")
        return {"reruns": int(out[0]), "left_guard": int(out[1]), "left_gate": int(out[2]), "left_geometry": int(out[3])}

    def params_tensor(self, params_np):
        """PARAMS_DTYPE[n] numpy -> uint8[n, 8] device tensor."""
        torch = self.torch
        a = np.ascontiguousarray(params_np, dtype=PARAMS_DTYPE).view(np.uint8).reshape(-1, 8)
        return torch.from_numpy(a.copy()).to(f"cuda:{self.device}")

    def select_diversity(self, iq_paths, sps=4, stream=None):
        """Transceiver.cpp:723-741.  iq_paths: int16[n, n_paths, burst_len, 2] -> (iq_sel int16[n, burst_len, 2],
        avg_energy float32[n], path uint8[n])."""
        torch = self.torch
        n, n_paths, burst_len = iq_paths.shape[0], iq_paths.shape[1], iq_paths.shape[2]
        dev = iq_paths.device
        sel = torch.empty((n, burst_len, 2), dtype=torch.int16, device=dev)
        avg = torch.empty(n, dtype=torch.float32, device=dev)
        path = torch.empty(n, dtype=torch.uint8, device=dev)
        _check(self.L.trxhip_select_diversity_batch(self.h, self._dev(iq_paths, torch.int16), n, n_paths, burst_len, sps,
                                                    self._dev(sel), self._dev(avg), self._dev(path), self._stream(stream)),
               "trxhip_select_diversity_batch")
        return sel, avg, path

    def apply_diversity_power(self, results, params, avg_energy, full_scale=32767.0, stream=None):
        _check(self.L.trxhip_apply_diversity_power(self.h, self._dev(results), self._dev(params), self._dev(avg_energy),
                                                   results.shape[0], full_scale, self._stream(stream)),
               "trxhip_apply_diversity_power")

    # ---- hot path ------------------------------------------------------------------------------
    def detect_demod(self, iq, params, sps=4, threshold=4.0, full_scale=32767.0, soft_stride=148, slice_bits=True,
                     results=None, soft=None, stream=None, want_soft=True, exact=False, idle_dummy=False, _diag_mask=0,
                     host_params=None, hint=None):
        """iq: int16[n, burst_len, 2] or complex64[n, burst_len] (device).  params: uint8[n, 8] (device).
        host_params: the caller's host copy of the parameters (PARAMS_DTYPE[n]), if it has one: the slot types decide the
        TRXHIP_FLAG_FEW_NB_SLOTS hint (a batch with few normal-burst slots runs the general kernel alone; results do not change).
        hint: that flag already worked out (few_nb_hint(host_params): a pass over the host array, a few milliseconds for a million
        slots -- a caller that launches the same slot table again and again computes it once).
        Returns (results uint8[n, 32], soft float32[n, soft_stride]) device tensors."""
        torch = self.torch
        n = iq.shape[0]
        burst_len = iq.shape[1]
        dev = f"cuda:{self.device}"
        if results is None:
            results = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        if soft is None and want_soft:
            soft = torch.empty((n, soft_stride), dtype=torch.float32, device=dev)
        assert params.shape == (n, 8) and params.dtype == torch.uint8
        sp = self._dev(soft, torch.float32) if soft is not None else _VP(0)
        if iq.dtype == torch.int16:
            assert iq.shape[2] == 2
            fn = self.L.trxhip_detect_demod_batch
            ip = self._dev(iq, torch.int16)
        elif iq.dtype == torch.complex64:
            fn = self.L.trxhip_detect_demod_batch_cf32
            ip = self._dev(iq)
        else:
            raise TrxHipError(f"unsupported IQ dtype {iq.dtype}")
        rc = fn(self.h, ip, self._dev(params), self._dev(results), sp, n, burst_len, sps,
                threshold, full_scale, soft_stride,
                (FLAG_SLICE if slice_bits else 0) | (FLAG_EXACT_DEMOD if exact else 0) |
                (FLAG_IDLE_DUMMY if idle_dummy else 0) | (int(_diag_mask) << 8) | (few_nb_hint(host_params) if hint is None else int(hint)),
                self._stream(stream))
        _check(rc, "trxhip_detect_demod_batch")
        return results, soft

    def demod_only(self, iq_cf32, params, ebp, sps=4, soft_stride=156, slice_bits=False, exact=False, stream=None):
        """demodAnyBurst() alone: iq complex64[n, L], params uint8[n, 8], ebp float32[n, 4] = {toa, amp_re, amp_im, 0}."""
        torch = self.torch
        n, burst_len = iq_cf32.shape
        results = torch.empty((n, 32), dtype=torch.uint8, device=iq_cf32.device)
        soft = torch.empty((n, soft_stride), dtype=torch.float32, device=iq_cf32.device)
        rc = self.L.trxhip_demod_batch_cf32(self.h, self._dev(iq_cf32), self._dev(params), self._dev(ebp, torch.float32),
                                            self._dev(results), self._dev(soft), n, burst_len, sps, soft_stride,
                                            (FLAG_SLICE if slice_bits else 0) | (FLAG_EXACT_DEMOD if exact else 0),
                                            self._stream(stream))
        _check(rc, "trxhip_demod_batch_cf32")
        return results, soft

    def delay_vector(self, x_cf32, delays, stream=None):
        """delayVector(): x complex64[n, len], delays float32[n] (samples).  Returns a new tensor."""
        torch = self.torch
        n, length = x_cf32.shape
        out = torch.empty_like(x_cf32)
        _check(self.L.trxhip_delay_vector_batch_cf32(self.h, self._dev(x_cf32), self._dev(out),
                                                     self._dev(delays, torch.float32), n, length, self._stream(stream)),
               "trxhip_delay_vector_batch_cf32")
        return out

    def scale_vector(self, x_cf32, scale, stream=None):
        """scaleVector(): in place x *= scale (complex)."""
        scale = complex(scale)
        _check(self.L.trxhip_scale_vector_cf32(self.h, self._dev(x_cf32), x_cf32.numel(), scale.real, scale.imag,
                                               self._stream(stream)), "trxhip_scale_vector_cf32")
        return x_cf32

    def demod_va(self, iq_cf32, params, scale=1.0 / 16383.0, soft_stride=156, slice_bits=False, detected=None, stream=None):
        """Viterbi alternative (cfg->use_va): scaleVector + demodAnyBurst_va.  iq complex64[n, L], params uint8[n, 8].
        detected: uint8[n, 32] result records of a detection launch (only bursts with rc > 0 are demodulated).
        Returns (soft float32[n, soft_stride], starts int32[n])."""
        torch = self.torch
        n, burst_len = iq_cf32.shape
        soft = torch.empty((n, soft_stride), dtype=torch.float32, device=iq_cf32.device)
        starts = torch.empty(n, dtype=torch.int32, device=iq_cf32.device)
        _check(self.L.trxhip_demod_va_batch_cf32(self.h, self._dev(iq_cf32), self._dev(params),
                                                 self._dev(detected) if detected is not None else _VP(0), self._dev(soft),
                                                 self._dev(starts), n, burst_len, scale, soft_stride,
                                                 FLAG_SLICE if slice_bits else 0, self._stream(stream)),
               "trxhip_demod_va_batch_cf32")
        return soft, starts

    def detect_sch(self, iq_cf32, state=0, sps=4, threshold=4.0, stream=None):
        """detectSCHBurst() for complex64[n_bufs, buf_len] buffers; state = SCH_DETECT_FULL / _NARROW / _BUFFER.
        Returns results uint8[n_bufs, 32] (rc, toa, amp, ci)."""
        torch = self.torch
        n, buf_len = iq_cf32.shape
        results = torch.empty((n, 32), dtype=torch.uint8, device=iq_cf32.device)
        _check(self.L.trxhip_detect_sch_batch_cf32(self.h, self._dev(iq_cf32), self._dev(results), n, buf_len, sps, state,
                                                   threshold, self._stream(stream)), "trxhip_detect_sch_batch_cf32")
        return results

    @staticmethod
    def results_to_numpy(results):
        return results.cpu().numpy().view(RESULT_DTYPE).reshape(-1)

    def pack_trxd(self, results, soft, rssi_offset=0.0, stream=None):
        torch = self.torch
        n = results.shape[0]
        pkt = torch.empty((n, TRXD_RECORD_BYTES), dtype=torch.uint8, device=results.device)
        rc = self.L.trxhip_pack_trxd_batch(self.h, self._dev(results), self._dev(soft, torch.float32), soft.shape[1],
                                           self._dev(pkt), n, rssi_offset, self._stream(stream))
        _check(rc, "trxhip_pack_trxd_batch")
        return pkt

    def pack_trxd_wire(self, results, params, soft, meta, pkt_stride=160, rssi_offset=0.0, stream=None):
        """TRXD v0/v1 datagrams (proto_trxd.c:68-117).  results uint8[n, 32], params uint8[n, 8], soft float32[n, stride]
        (sliced), meta uint8[n, 8] (TRXD_META_DTYPE).  Returns (pkt uint8[n, pkt_stride], pkt_len int16[n])."""
        torch = self.torch
        n = results.shape[0]
        pkt = torch.empty((n, pkt_stride), dtype=torch.uint8, device=results.device)
        plen = torch.empty(n, dtype=torch.int16, device=results.device)
        rc = self.L.trxhip_pack_trxd_wire_batch(self.h, self._dev(results), self._dev(params), self._dev(soft, torch.float32),
                                                soft.shape[1], self._dev(meta), self._dev(pkt), pkt_stride, self._dev(plen), n,
                                                rssi_offset, self._stream(stream))
        _check(rc, "trxhip_pack_trxd_wire_batch")
        return pkt, plen

    # ---- arch kernels ---------------------------------------------------------------------------
    def convolve(self, x, h, start, length, complex_taps, stream=None):
        """x: complex64[n_vec, x_len], h: complex64[h_len] -> complex64[n_vec, length]"""
        torch = self.torch
        n_vec, x_len = x.shape
        y = torch.empty((n_vec, length), dtype=torch.complex64, device=x.device)
        fn = self.L.trxhip_convolve_complex_batch if complex_taps else self.L.trxhip_convolve_real_batch
        rc = fn(self.h, self._dev(x), x_len, self._dev(h), h.shape[0], self._dev(y), length, start, length, n_vec,
                self._stream(stream))
        _check(rc, "trxhip_convolve_batch")
        return y

    def energy_detect(self, iq_cf32, window, stream=None):
        torch = self.torch
        n, burst_len = iq_cf32.shape
        out = torch.empty(n, dtype=torch.float32, device=iq_cf32.device)
        _check(self.L.trxhip_energy_detect_batch_cf32(self.h, self._dev(iq_cf32), n, burst_len, window, self._dev(out),
                                                      self._stream(stream)), "trxhip_energy_detect_batch_cf32")
        return out

    def vector_slicer(self, src, stream=None):
        torch = self.torch
        out = torch.empty_like(src)
        _check(self.L.trxhip_vector_slicer(self.h, self._dev(out), self._dev(src, torch.float32), src.numel(),
                                           self._stream(stream)), "trxhip_vector_slicer")
        return out

    def convert_short_float(self, s, stream=None):
        torch = self.torch
        out = torch.empty(s.shape, dtype=torch.float32, device=s.device)
        _check(self.L.trxhip_convert_short_float(self.h, self._dev(out), self._dev(s, torch.int16), s.numel(),
                                                 self._stream(stream)), "trxhip_convert_short_float")
        return out

    def channelize(self, wide_iq, n_blocks, m=4, block_len=192, h_len=16, stream=None):
        """wide_iq: int16[n_blocks*block_len*m, 2] -> complex64[m, n_blocks*block_len]"""
        torch = self.torch
        out = torch.empty((m, n_blocks * block_len), dtype=torch.complex64, device=wide_iq.device)
        _check(self.L.trxhip_channelize_batch(self.h, self._dev(wide_iq, torch.int16), self._dev(out), n_blocks, m,
                                              block_len, h_len, self._stream(stream)), "trxhip_channelize_batch")
        return out

    def resample(self, x, p, q, stream=None):
        """x: complex64[n_chan, n_in] -> complex64[n_chan, n_in*p/q]"""
        torch = self.torch
        n_chan, n_in = x.shape
        n_out = n_in // q * p
        out = torch.empty((n_chan, n_out), dtype=torch.complex64, device=x.device)
        _check(self.L.trxhip_resample_batch(self.h, self._dev(x), self._dev(out), n_in, p, q, n_chan, n_in, n_out,
                                            self._stream(stream)), "trxhip_resample_batch")
        return out


class RxFrontEnd:
    """Streaming Channelizer(4, block_len, 16) + Resampler(p, q, 16) with carried history (trxhip_rx_frontend_*)."""

    def __init__(self, trx, block_len=192, p=65, q=48):
        self.trx = trx
        self.block_len, self.p, self.q = block_len, p, q
        h = _VP()
        _check(trx.L.trxhip_rx_frontend_create(trx.h, block_len, p, q, C.byref(h)), "trxhip_rx_frontend_create")
        self.h = h

    def reset(self, stream=None):
        _check(self.trx.L.trxhip_rx_frontend_reset(self.h, self.trx._stream(stream)), "trxhip_rx_frontend_reset")

    def seed(self, wide_prev, n_blocks_prev, stream=None):
        """Start mid-stream: wide_prev = the n_blocks_prev >= 1 blocks preceding the shard (trxhip_rx_frontend_seed)."""
        ptr = self.trx._dev(wide_prev, self.trx.torch.int16) if n_blocks_prev else None
        _check(self.trx.L.trxhip_rx_frontend_seed(self.h, ptr, n_blocks_prev, self.trx._stream(stream)), "trxhip_rx_frontend_seed")

    def pull(self, wide_iq, n_blocks, stream=None, out=None):
        """wide_iq: int16[n_blocks*block_len*4, 2] -> complex64[4, n_blocks*block_len*p/q] (written into `out` when given)"""
        torch = self.trx.torch
        n_out = n_blocks * self.block_len // self.q * self.p
        if out is None:
            out = torch.empty((4, n_out), dtype=torch.complex64, device=wide_iq.device)
        elif tuple(out.shape) != (4, n_out) or out.dtype != torch.complex64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex64[4, n_out] tensor")
        _check(self.trx.L.trxhip_rx_frontend_pull(self.h, self.trx._dev(wide_iq, torch.int16), n_blocks, self.trx._dev(out),
                                                  n_out, self.trx._stream(stream)), "trxhip_rx_frontend_pull")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.trx.L.trxhip_rx_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _HostPipeCfg(C.Structure):
    _fields_ = [("max_bursts", C.c_uint32), ("depth", C.c_int32), ("burst_len", C.c_int32), ("sps", C.c_int32),
                ("soft_stride", C.c_int32), ("pkt_stride", C.c_int32), ("flags", C.c_int32), ("threshold", C.c_float),
                ("full_scale", C.c_float), ("rssi_offset", C.c_float), ("n_paths", C.c_int32)]


class _HostPipeSlot(C.Structure):
    _fields_ = [("iq", _VP), ("params", _VP), ("meta", _VP), ("results", _VP), ("soft", _VP), ("pkt", _VP), ("pkt_len", _VP)]


class HostPipe:
    """Host-fed, stream-pipelined hot path (trxhip_hostpipe_*): pinned staging slots, one stream each.

    slot(i) gives numpy views of slot i's pinned buffers; fill iq/params(/meta), submit(i, n), wait(i), read
    results/soft/pkt.  run() is the convenience form for ordinary numpy arrays."""

    def __init__(self, trx, max_bursts, depth=3, burst_len=625, sps=4, soft_stride=148, pkt_stride=0, flags=FLAG_SLICE,
                 threshold=4.0, full_scale=32767.0, rssi_offset=0.0, n_paths=1):
        self.trx = trx
        self.cfg = _HostPipeCfg(max_bursts, depth, burst_len, sps, soft_stride, pkt_stride, flags, threshold, full_scale,
                                rssi_offset, n_paths)
        h = _VP()
        _check(trx.L.trxhip_hostpipe_create(trx.h, C.byref(self.cfg), C.byref(h)), "trxhip_hostpipe_create")
        self.h = h
        self.depth = depth
        self._slots = [self._views(i) for i in range(depth)]

    def _views(self, i):
        s = _HostPipeSlot()
        _check(self.trx.L.trxhip_hostpipe_slot_buffers(self.h, i, C.byref(s)), "trxhip_hostpipe_slot_buffers")
        c = self.cfg
        n = c.max_bursts

        def view(ptr, nbytes, dtype, shape):
            if not ptr:
                return None
            buf = (C.c_ubyte * nbytes).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype).reshape(shape)
        return {
            "iq": (view(s.iq, n * c.burst_len * 4, np.int16, (n, c.burst_len, 2)) if c.n_paths <= 1 else
                   view(s.iq, n * c.n_paths * c.burst_len * 4, np.int16, (n, c.n_paths, c.burst_len, 2))),
            "params": view(s.params, n * 8, PARAMS_DTYPE, (n,)),
            "meta": view(s.meta, n * 8, TRXD_META_DTYPE, (n,)),
            "results": view(s.results, n * 32, RESULT_DTYPE, (n,)),
            "soft": view(s.soft, n * c.soft_stride * 4, np.float32, (n, c.soft_stride)) if c.soft_stride else None,
            "pkt": view(s.pkt, n * c.pkt_stride, np.uint8, (n, c.pkt_stride)) if c.pkt_stride else None,
            "pkt_len": view(s.pkt_len, n * 2, np.uint16, (n,)) if c.pkt_stride else None,
        }

    def slot(self, i):
        return self._slots[i]

    def submit(self, i, n):
        _check(self.trx.L.trxhip_hostpipe_submit(self.h, i, n), "trxhip_hostpipe_submit")

    # ---- bursts by reference: the samples stay where they are (a registered host range), the slot carries pointers ----
    def register_host(self, array):
        """Pin a numpy array (the radio's receive ring) and map it into the device: bursts inside it can be submitted by
        address.  The caller keeps the array alive until unregister_host() / close()."""
        _check(self.trx.L.trxhip_hostpipe_register_host(self.h, _VP(array.ctypes.data), array.nbytes), "trxhip_hostpipe_register_host")
        self._registered = getattr(self, "_registered", []) + [array]     # pinned pages must not be freed under the device

    def unregister_host(self, array):
        _check(self.trx.L.trxhip_hostpipe_unregister_host(self.h, _VP(array.ctypes.data)), "trxhip_hostpipe_unregister_host")
        self._registered = [a for a in getattr(self, "_registered", []) if a is not array]

    def sources(self, i):
        """uint64 view of slot i's pointer array (max_bursts host addresses)."""
        q = _VP()
        _check(self.trx.L.trxhip_hostpipe_slot_sources(self.h, i, C.byref(q)), "trxhip_hostpipe_slot_sources")
        buf = (C.c_ubyte * (self.cfg.max_bursts * 8)).from_address(q.value)
        return np.frombuffer(buf, dtype=np.uint64)

    def submit_by_ref(self, i, n):
        _check(self.trx.L.trxhip_hostpipe_submit_by_ref(self.h, i, n), "trxhip_hostpipe_submit_by_ref")

    def wait(self, i):
        _check(self.trx.L.trxhip_hostpipe_wait(self.h, i), "trxhip_hostpipe_wait")

    def query(self, i):
        return int(self.trx.L.trxhip_hostpipe_query(self.h, i))

    def run(self, iq, params, meta=None):
        """iq int16[n, burst_len, 2], params PARAMS_DTYPE[n], meta TRXD_META_DTYPE[n] (numpy, pageable).
        Returns dict(results, soft, pkt, pkt_len) of numpy arrays."""
        c = self.cfg
        n = iq.shape[0]
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        params = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
        res = np.empty(n, dtype=RESULT_DTYPE)
        soft = np.empty((n, c.soft_stride), dtype=np.float32) if c.soft_stride else None
        pkt = np.empty((n, c.pkt_stride), dtype=np.uint8) if c.pkt_stride else None
        plen = np.empty(n, dtype=np.uint16) if c.pkt_stride else None
        if meta is not None:
            meta = np.ascontiguousarray(meta, dtype=TRXD_META_DTYPE)

        def ptr(a):
            return _VP(a.ctypes.data) if a is not None else _VP(0)
        _check(self.trx.L.trxhip_hostpipe_run(self.h, ptr(iq), ptr(params), ptr(meta), ptr(res), ptr(soft), ptr(pkt), ptr(plen), n),
               "trxhip_hostpipe_run")
        return {"results": res, "soft": soft, "pkt": pkt, "pkt_len": plen}

    def close(self):
        if getattr(self, "h", None):
            self._slots = None
            self.trx.L.trxhip_hostpipe_destroy(self.h)
            self.h = None
            self._registered = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
