#!/usr/bin/env python3
"""tests/golden/ref_stage_vectors.npz -- the detect / demod call graph stage by stage on the reference's OWN arch objects.

Run in the build container only (needs /root/reference for oracle/_ref; the output is committed data).  For the captured
burst (utils/va-test/nb_chunk_tsc7.cfile) and 64 seeded synthetic bursts (32 normal, 32 access) every FIR stage of
detectAnyBurst / demodAnyBurst is executed by the reference's unmodified kernels (oracle/_ref/libref_generic.so: arch/common
+ arch/x86 + Resampler.cpp compiled where they lie, oracle/Makefile), driven with the arguments the reference's call sites
use:

  dec    downsampleBurst (sigProcLib.cpp:1587-1601): Resampler(1, 4, 16)::rotate over [16 zeros | 624 samples] -> 156
  corr   detectBurst's convolve(.., CUSTOM, start, len) (:1674) = convolve_complex over the decimated burst:
           N = 16 (TSC of the slot,   start 71, len 19: analyzeTrafficBurst with max_toa 3, :1887-1904)
           N = 40 (RACH TS0,          start 39, len 79: detectRACHBurst with max_toa 63, :1782-1803)
           N = 64 (SCH sequence,      start 63, len 93: a window inside the vector; the MS-side search, :1805-1861)
  delay  delayVector's convolve(in, h, NULL, NO_DELAY) (:1060) = convolve_real with one of the 64 fractional-delay
         filters over [10 zeros | 625 samples | 10 zeros], start 10 + 10: 625 outputs (three 64-sample windows are stored)

The taps (training sequences, delay filters) are the oracle's tables, which tests/test_oracle.py pins -- byte for byte -- to
the tables the same generator produces when it runs over these very kernels (orc_set_arch).  Stored: the inputs (int16
bursts, TSC, filter index) and the generic build's outputs; the SSE build is covered by tests/test_oracle.py where
oracle/_ref exists.  Consumers: tests/test_oracle.py (the oracle's stage functions, bit for bit) and
tests/test_gpu_aux_kernels.py (the HIP stand-alone entry points trxhip_resample_batch / trxhip_convolve_complex_batch /
trxhip_delay_vector_batch_cf32, bit for bit)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import oracle_lib as O                                   # noqa: E402
from make_golden import ref_lib, aligned                 # noqa: E402

DELAY_WIN = ((0, 64), (300, 364), (561, 625))


def main():
    from osmo_trx_amd import synth
    L = ref_lib("generic")
    T = O.tables()
    cap = np.fromfile(os.path.join(HERE, "nb_chunk_tsc7.cfile"), dtype=np.complex64)[:625]
    nb, p_nb, _ = synth.make_normal_bursts(32, "cpu", 4, seed=0x57A6E, delay_sym=(0.0, 3.0))
    ab, p_ab, _ = synth.make_access_bursts(32, "cpu", seed=0x57A6F)
    iq = np.concatenate([nb.numpy(), ab.numpy()])                       # int16 [64, 625, 2]
    tsc = np.concatenate([[7], p_nb["tsc"], np.zeros(32, dtype=np.uint8)]).astype(np.uint8)
    kind = np.concatenate([[0], np.zeros(32, dtype=np.uint8), np.ones(32, dtype=np.uint8)]).astype(np.uint8)   # 0 normal, 1 access
    bursts = [cap] + [b.astype(np.float32).view(np.complex64).reshape(625) for b in iq]
    rng = np.random.default_rng(0x57A70)
    filt = rng.integers(1, 64, len(bursts)).astype(np.int32)
    filt[0] = 33

    dec_h = L.ref_resampler_new(1, 4, 16, 1.0)
    out = {"iq": iq, "tsc": tsc, "kind": kind, "filt": filt, "delay_win": np.array(DELAY_WIN, dtype=np.int32)}
    decs, corrs, corr_sch, delays = [], [], [], []

    def conv(fn, x, h, start, ln):
        xb = aligned(2 * len(x)); xb[:] = x.view(np.float32)
        hb = aligned(2 * len(h)); hb[:] = np.ascontiguousarray(h, dtype=np.complex64).view(np.float32)
        yb = aligned(2 * ln); yb[:] = 0
        rc = fn(xb.ctypes.data, len(x), hb.ctypes.data, len(h), yb.ctypes.data, ln, start, ln)
        assert rc == ln
        return yb.copy().view(np.complex64)

    for k, x in enumerate(bursts):
        buf = aligned(2 * (16 + 624)); buf[:] = 0
        buf[32:] = x[:624].view(np.float32)
        y = aligned(2 * 156)
        assert L.ref_resampler_rotate(dec_h, buf[32:].ctypes.data, 624, y.ctypes.data, 156) == 156
        dec = y.copy().view(np.complex64)
        decs.append(dec)
        if kind[k] == 0:
            corrs.append(conv(L.convolve_complex, dec, T["midamble"][int(tsc[k])]["seq"], 71, 19))
        else:
            corrs.append(conv(L.convolve_complex, dec, T["rach"][0]["seq"], 39, 79))
        corr_sch.append(conv(L.convolve_complex, dec, T["sch"]["seq"], 63, 93))
        h = np.zeros(20, dtype=np.complex64)
        h.real = T["delay_filt"][int(filt[k])]
        xp = np.concatenate([np.zeros(10, np.complex64), x, np.zeros(10, np.complex64)])       # signalVector(*x, head, tail), NO_DELAY
        d = conv(L.convolve_real, xp, h, 10 + 10, 625)
        delays.append(np.concatenate([d[a:b] for a, b in DELAY_WIN]))
    L.ref_resampler_free(dec_h)
    out["dec"] = np.stack(decs).view(np.float32)
    out["corr_nb"] = np.stack([c for c, kd in zip(corrs, kind) if kd == 0]).view(np.float32)
    out["corr_ab"] = np.stack([c for c, kd in zip(corrs, kind) if kd == 1]).view(np.float32)
    out["corr_sch"] = np.stack(corr_sch).view(np.float32)
    out["delay"] = np.stack(delays).view(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_stage_vectors.npz"), **out)
    print("ref_stage_vectors.npz:", {k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "ref_stage_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
