"""Synchronisation-burst test signals (test infrastructure: built with the oracle's modulator).
SCH burst = 3 tail + 39 data + 64 extended training sequence + 39 data + 3 tail (3GPP TS 45.002 5.2.5)."""
import numpy as np
import oracle_lib as O

# 3GPP TS 45.002 5.2.5 extended training sequence (GSM/GSMCommon.cpp:83-84 in the reference)
SCH_TS = "1011100101100010000001000000111100101101010001010111011000011011"


def sch_bits(rng):
    ts = np.array([int(c) for c in SCH_TS], dtype=np.uint8)
    return np.concatenate([np.zeros(3, np.uint8), rng.integers(0, 2, 39, dtype=np.uint8), ts,
                           rng.integers(0, 2, 39, dtype=np.uint8), np.zeros(3, np.uint8)])


def sch_burst(rng, n_out, offset, amp=3000.0, noise=100.0, present=True):
    """complex64[n_out] at 4 SPS: one Laurent-GMSK SCH burst starting `offset` samples in, AWGN of std `noise`."""
    bits = sch_bits(rng)
    y = np.zeros(n_out, dtype=np.complex64)
    if present:
        x = O.modulate_burst(bits, 8, 4)
        m = min(len(x), n_out - offset)
        y[offset:offset + m] = x[:m] * np.complex64(amp * np.exp(1j * rng.uniform(0, 2 * np.pi)))
    y += ((rng.normal(size=n_out) + 1j * rng.normal(size=n_out)) * noise).astype(np.complex64)
    return y, bits
