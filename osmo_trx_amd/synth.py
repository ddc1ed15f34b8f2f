"""Synthetic receive-side workload for the burst DSP (BASELINE.json configs).

Generates int16 IQ bursts the way a radio would hand them to Transceiver::pullRadioVector():
GMSK-modulated GSM normal / access bursts (the modulation follows the structure of the reference's
own generators -- Laurent two-pulse GMSK at 4 SPS, sigProcLib.cpp:595-670, single-pulse at 1 SPS,
:938-967; burst layouts of genRandNormalBurst :768-806 and genRandAccessBurst :811-841), then a
channel: complex gain, fractional delay, AWGN, int16 quantisation with saturation.

This is the workload generator for tests and bench.py (torch, runs on CPU or GPU); it is neither the
product nor the oracle, and need not be bit-identical to anything.
"""
import math

import numpy as np
import torch

from .trxhip import (PARAMS_DTYPE, TSC as T_TSC, RACH as T_RACH, EXT_RACH as T_EXT_RACH, IDLE as T_IDLE, OFF as T_OFF,
                     EDGE as T_EDGE)

# 3GPP TS 45.002 training sequences / access-burst bits
TSC_BITS = [
    "00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110",
    "01000111101101000100011110", "00011010111001000001101011", "01001110101100000100111010",
    "10100111110110001010011111", "11101111000100101110111100",
]
RACH_SYNC = [
    "01001011011111111001100110101010001111000",
    "01010100111110001000011000101111001001101",
    "11101111001001110101011000001101101110111",
]
RACH_HEAD = "00111010"          # 8 extended tail bits
EDGE_TSC_BITS = [
    "111111001111111001111001001001111111111111001111111111001111111001111001001001",
    "111111001111001001111001001001111001001001001111111111001111001001111001001001",
    "111001111111111111001001001111001001001111001111111001111111111111001001001111",
    "111001111111111001001001001111001001111001111111111001111111111001001001001111",
    "111111111001001111001111001001001111111001111111111111111001001111001111001001",
    "111001111111001001001111001111001001111111111111111001111111001001001111001111",
    "001111001111111001001001001001111001001111111111001111001111111001001001001001",
    "001001001111001001001001111111111001111111001111001001001111001001001001111111",
]
# 8-PSK constellation indexed by b0 | b1<<1 | b2<<2 (3GPP TS 45.004)
PSK8 = [(-0.70710678, 0.70710678), (0.0, -1.0), (0.0, 1.0), (0.70710678, -0.70710678),
        (-1.0, 0.0), (-0.70710678, -0.70710678), (0.70710678, 0.70710678), (1.0, 0.0)]

# Laurent pulses at 4 SPS (BT = 0.3): C0 (16 taps), C1 (8 taps)
C0_4 = [0.0, 4.46348606e-03, 2.84385729e-02, 1.03184855e-01, 2.56065552e-01, 4.76375085e-01, 7.05961177e-01,
        8.71291644e-01, 9.29453645e-01, 8.71291644e-01, 7.05961177e-01, 4.76375085e-01, 2.56065552e-01,
        1.03184855e-01, 2.84385729e-02, 4.46348606e-03]
C1_4 = [0.0, 8.16373112e-03, 2.84385729e-02, 5.64158904e-02, 7.05463553e-02, 5.64158904e-02, 2.84385729e-02,
        8.16373112e-03]


def _pulse_1sps():
    a = np.arange(4, dtype=np.float64) - 1.5
    p = 0.96 * np.exp(-1.1380 * a * a - 0.527 * a ** 4)
    return (p / math.sqrt(float((p * p).sum()))).tolist()


C0_1 = _pulse_1sps()
SEED = 0x05D07A58
# TOA the detector reports for an undelayed burst out of these modulators (modulator padding + pulse
# and decimator group delay; the reference's own modulateBurst output measures the same): subtracted
# so that `delay_sym` below is the TOA the detector should report.
BASE_TOA = {4: 4.60, 1: 1.50}


def _bits(s, device):
    return torch.tensor([int(c) for c in s], dtype=torch.uint8, device=device)


def _fir_causal(x, taps):
    """y[i] = sum_k x[i-(H-1)+k] * taps[k]  (zero history) for complex x[N, L]."""
    H = len(taps)
    w = torch.tensor(taps, dtype=torch.float32, device=x.device).view(1, 1, H)
    xr = torch.view_as_real(x).permute(0, 2, 1).reshape(-1, 1, x.shape[1])
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(xr, (H - 1, 0)), w)
    y = y.reshape(x.shape[0], 2, x.shape[1]).permute(0, 2, 1).contiguous()
    return torch.view_as_complex(y)


def modulate_laurent_4sps(bits):
    """bits uint8[N, nb] (nb <= 148+..) -> complex64[N, 625]"""
    N, nb = bits.shape
    dev = bits.device
    L = 625
    pos = torch.arange(L, device=dev, dtype=torch.float32)
    rot = torch.polar(torch.ones(L, device=dev), pos * (math.pi / 8.0))
    sym = bits.to(torch.float32) * 2.0 - 1.0
    a = torch.full((N, nb + 2), -1.0, device=dev)
    a[:, 1:nb + 1] = sym                               # k = -1 .. nb at positions 4(k+1)
    c0 = torch.zeros((N, L), dtype=torch.complex64, device=dev)
    idx0 = 4 * torch.arange(nb + 2, device=dev)
    c0[:, idx0] = a.to(torch.complex64) * rot[idx0]
    # C1 symbols at positions 4(k+1), k = 1..nb:  j * phi_k * c0, phi_1 = -1, phi_k = 2*(b[k-1]^b[k-2])-1
    phi = torch.full((N, nb), -1.0, device=dev)
    if nb >= 2:
        x = (bits[:, 1:] ^ bits[:, :-1]).to(torch.float32) * 2.0 - 1.0     # b[k-1]^b[k-2], k = 2..nb
        phi[:, 1:] = x
    idx1 = 4 * (torch.arange(1, nb + 1, device=dev) + 1)
    c1 = torch.zeros((N, L), dtype=torch.complex64, device=dev)
    c1[:, idx1] = c0[:, idx1] * (1j * phi.to(torch.complex64))
    return _fir_causal(c0, C0_4) + _fir_causal(c1, C1_4)


def modulate_basic_1sps(bits, length):
    N, nb = bits.shape
    dev = bits.device
    pos = torch.arange(length, device=dev, dtype=torch.float32)
    rot = torch.polar(torch.ones(length, device=dev), pos * (math.pi / 2.0))
    x = torch.zeros((N, length), dtype=torch.complex64, device=dev)
    x[:, :nb] = (bits.to(torch.float32) * 2.0 - 1.0).to(torch.complex64) * rot[:nb]
    return _fir_causal(x, C0_1)


def modulate_edge_4sps(bits):
    """8-PSK: bits uint8[N, 444] -> complex64[N, 625]: 3pi/8 rotation per symbol, one symbol of delay, C0 pulse."""
    N = bits.shape[0]
    dev = bits.device
    b = bits.view(N, 148, 3).to(torch.int64)
    idx = b[:, :, 0] | (b[:, :, 1] << 1) | (b[:, :, 2] << 2)
    table = torch.tensor([complex(*p) for p in PSK8], dtype=torch.complex64, device=dev)
    sym = table[idx]
    k = torch.arange(148, device=dev, dtype=torch.float32)
    rot = torch.polar(torch.ones(148, device=dev), k * (3.0 * math.pi / 8.0))
    x = torch.zeros((N, 625), dtype=torch.complex64, device=dev)
    x[:, 4 + 4 * torch.arange(148, device=dev)] = sym * rot
    return _fir_causal(x, C0_4)


def edge_burst_bits(n, tsc, gen, device):
    """3 tail symbols (111) | 58 data | 26 training | 58 data | 3 tail: 444 bits"""
    bits = torch.randint(0, 2, (n, 444), generator=gen, device=device, dtype=torch.uint8)
    bits[:, :9] = 1
    bits[:, -9:] = 1
    tab = torch.stack([_bits(s, device) for s in EDGE_TSC_BITS])
    bits[:, 9 + 174:9 + 174 + 78] = tab[tsc.long()]
    return bits


def make_edge_bursts(n, device="cpu", seed=SEED + 4, max_toa=3, amp_range=(2000.0, 12000.0), snr_range=(18.0, 35.0),
                     delay_sym=(0.0, 3.0), chunk=65536):
    """EDGE (8-PSK) normal bursts, 4 SPS, burst i uses TSC i%8; slots marked EDGE."""
    device = torch.device(device)
    iq = torch.empty((n, 625, 2), dtype=torch.int16, device=device)
    params = np.zeros(n, dtype=PARAMS_DTYPE)
    params["type"] = T_EDGE
    params["max_toa"] = max_toa
    params["tsc"] = (np.arange(n) % 8).astype(np.uint8)
    all_bits = np.zeros((n, 444), dtype=np.uint8)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        gen = _gen(seed + 31337 * (c0 // chunk), device)
        t = torch.from_numpy(params["tsc"][c0:c1]).to(device)
        bits = edge_burst_bits(m, t, gen, device)
        wave = modulate_edge_4sps(bits)
        u = torch.rand((3, m), generator=gen, device=device)
        amp = amp_range[0] * torch.pow(torch.tensor(amp_range[1] / amp_range[0], device=device), u[0])
        snr = snr_range[0] + (snr_range[1] - snr_range[0]) * u[1]
        dly = delay_sym[0] + (delay_sym[1] - delay_sym[0]) * u[2]
        iq[c0:c1] = _channel(wave, amp, snr, (dly - BASE_TOA[4]) * 4.0, torch.zeros(m, dtype=torch.bool, device=device), gen)
        all_bits[c0:c1] = bits.cpu().numpy()
    return iq, params, all_bits


def _frac_delay(x, delay):
    """Delay complex x[N, L] by delay[N] samples (fractional; a small negative delay drops the ramp-up) via FFT."""
    N, L = x.shape
    nfft = 1 << int(math.ceil(math.log2(L + max(float(delay.max().item()), 0.0) + 64)))
    X = torch.fft.fft(x, n=nfft, dim=1)
    f = torch.fft.fftfreq(nfft, device=x.device).to(torch.float32)
    ph = torch.polar(torch.ones((N, nfft), device=x.device), -2.0 * math.pi * f[None, :] * delay[:, None])
    return torch.fft.ifft(X * ph, dim=1)[:, :L].contiguous()


def normal_burst_bits(n, tsc, gen, device):
    """genRandNormalBurst layout: 3 tail | 57 data | steal | 26 TSC | steal | 57 data | 3 tail"""
    bits = torch.zeros((n, 148), dtype=torch.uint8, device=device)
    bits[:, 3:60] = torch.randint(0, 2, (n, 57), generator=gen, device=device, dtype=torch.uint8)
    bits[:, 88:145] = torch.randint(0, 2, (n, 57), generator=gen, device=device, dtype=torch.uint8)
    tsc_tab = torch.stack([_bits(s, device) for s in TSC_BITS])
    bits[:, 61:87] = tsc_tab[tsc.long()]
    return bits


def access_burst_bits(n, ts, gen, device):
    """8 tail | 41 sync | 36 data | 3 tail"""
    bits = torch.zeros((n, 88), dtype=torch.uint8, device=device)
    bits[:, 0:8] = _bits(RACH_HEAD, device)
    sync = torch.stack([_bits(s, device) for s in RACH_SYNC])
    bits[:, 8:49] = sync[ts.long()]
    bits[:, 49:85] = torch.randint(0, 2, (n, 36), generator=gen, device=device, dtype=torch.uint8)
    return bits


def _channel(wave, amp, snr_db, delay_samples, noise_only, gen, clip_mask=None):
    """Apply delay, complex gain, AWGN; quantise to int16 with saturation.  Returns int16[N, L, 2]."""
    N, L = wave.shape
    dev = wave.device
    y = _frac_delay(wave, delay_samples)
    phase = torch.rand(N, generator=gen, device=dev) * (2 * math.pi)
    g = torch.polar(amp, phase)
    y = y * g[:, None]
    y = torch.where(noise_only[:, None], torch.zeros_like(y), y)
    sigma = amp * torch.pow(10.0, -snr_db / 20.0) / math.sqrt(2.0)
    noise = torch.randn((N, L, 2), generator=gen, device=dev) * sigma[:, None, None]
    out = torch.view_as_real(y) + noise
    if clip_mask is not None and bool(clip_mask.any()):
        peak = out.abs().amax(dim=(1, 2)).clamp_min(1.0)
        out = torch.where(clip_mask[:, None, None], out * (34000.0 / peak)[:, None, None], out)
    return torch.round(out).clamp_(-32768, 32767).to(torch.int16).contiguous()


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def make_normal_bursts(n, device="cpu", sps=4, seed=SEED, max_toa=3, tsc=None, amp_range=(500.0, 20000.0),
                       snr_range=(5.0, 30.0), delay_sym=(0.0, 4.0), p_noise=0.05, p_clip=0.01, burst_len=None,
                       chunk=65536, chunk0=0):
    """BASELINE.json configs[1] (sps=4: all 8 TSCs, burst i uses TSC i%8) / configs[0] (sps=1).
    Returns (iq int16[n, L, 2] on `device`, params PARAMS_DTYPE[n] numpy, truth dict of numpy arrays).
    The batch is generated in independently seeded chunks; chunk0 = index of the first one, so that a rank can generate
    bursts [chunk0 * chunk, chunk0 * chunk + n) of a larger batch without the rest (bench.py's strong-scaling leg)."""
    device = torch.device(device)
    L = burst_len or (625 if sps == 4 else 156)
    iq = torch.empty((n, L, 2), dtype=torch.int16, device=device)
    truth = {k: np.zeros(n, dtype=np.float32) for k in ("delay_sym", "amp", "snr_db")}
    truth["noise_only"] = np.zeros(n, dtype=bool)
    truth["clipped"] = np.zeros(n, dtype=bool)
    params = np.zeros(n, dtype=PARAMS_DTYPE)
    params["type"] = T_TSC
    params["max_toa"] = max_toa
    all_tsc = (np.arange(n) % 8).astype(np.uint8) if tsc is None else np.full(n, tsc, dtype=np.uint8)
    params["tsc"] = all_tsc
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        gen = _gen(seed + 7919 * (chunk0 + c0 // chunk), device)
        t = torch.from_numpy(all_tsc[c0:c1]).to(device)
        bits = normal_burst_bits(m, t, gen, device)
        wave = modulate_laurent_4sps(bits) if sps == 4 else modulate_basic_1sps(bits, L)
        u = torch.rand((4, m), generator=gen, device=device)
        amp = amp_range[0] * torch.pow(torch.tensor(amp_range[1] / amp_range[0], device=device), u[0])
        snr = snr_range[0] + (snr_range[1] - snr_range[0]) * u[1]
        dly = delay_sym[0] + (delay_sym[1] - delay_sym[0]) * u[2]
        noise_only = u[3] < p_noise
        clip = (u[3] >= p_noise) & (u[3] < p_noise + p_clip)
        iq[c0:c1] = _channel(wave, amp, snr, (dly - BASE_TOA[sps]) * sps, noise_only, gen, clip)
        truth["delay_sym"][c0:c1] = dly.cpu().numpy()
        truth["amp"][c0:c1] = amp.cpu().numpy()
        truth["snr_db"][c0:c1] = snr.cpu().numpy()
        truth["noise_only"][c0:c1] = noise_only.cpu().numpy()
        truth["clipped"][c0:c1] = clip.cpu().numpy()
    return iq, params, truth


def make_access_bursts(n, device="cpu", seed=SEED + 1, max_toa=63, ext=False, amp_range=(500.0, 20000.0),
                       snr_range=(5.0, 30.0), p_noise=0.05, chunk=65536, chunk0=0):
    """BASELINE.json configs[2]: access bursts, integer+fractional delay in [0, 63] symbols, 4 SPS."""
    device = torch.device(device)
    L = 625
    iq = torch.empty((n, L, 2), dtype=torch.int16, device=device)
    params = np.zeros(n, dtype=PARAMS_DTYPE)
    params["type"] = T_EXT_RACH if ext else T_RACH
    params["max_toa"] = max_toa
    truth = {"delay_sym": np.zeros(n, dtype=np.float32), "ts": np.zeros(n, dtype=np.uint8),
             "noise_only": np.zeros(n, dtype=bool)}
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        gen = _gen(seed + 104729 * (chunk0 + c0 // chunk), device)
        ts = (torch.randint(0, 3, (m,), generator=gen, device=device) if ext
              else torch.zeros(m, dtype=torch.int64, device=device))
        bits = access_burst_bits(m, ts, gen, device)
        wave = modulate_laurent_4sps(bits)
        u = torch.rand((4, m), generator=gen, device=device)
        amp = amp_range[0] * torch.pow(torch.tensor(amp_range[1] / amp_range[0], device=device), u[0])
        snr = snr_range[0] + (snr_range[1] - snr_range[0]) * u[1]
        dly = u[2] * float(min(max_toa, 63))
        noise_only = u[3] < p_noise
        iq[c0:c1] = _channel(wave, amp, snr, (dly - BASE_TOA[4]) * 4.0, noise_only, gen)
        truth["delay_sym"][c0:c1] = dly.cpu().numpy()
        truth["ts"][c0:c1] = ts.cpu().numpy().astype(np.uint8)
        truth["noise_only"][c0:c1] = noise_only.cpu().numpy()
    return iq, params, truth


def make_mixed_bursts(n, device="cpu", seed=SEED + 2, chunk=65536, offset=0):
    """BASELINE.json configs[4]: 7:1 NB:RACH interleaved (every 8th burst is an access burst).
    offset: generate bursts [offset, offset + n) of a larger batch (a multiple of 8 * chunk, so that both the normal-burst
    chunks and the access-burst chunks of the global batch start on a chunk boundary): concatenating the shards of all
    ranks gives exactly the batch one call with offset 0 would."""
    device = torch.device(device)
    if offset % (8 * chunk):
        raise ValueError("offset must be a multiple of 8 * chunk")
    iq_nb, p_nb, _ = make_normal_bursts(n, device, 4, seed, chunk=chunk, chunk0=offset // chunk)
    n_r = (n + 7) // 8
    iq_r, p_r, _ = make_access_bursts(n_r, device, seed + 1, chunk=chunk, chunk0=offset // (8 * chunk))
    sel = torch.arange(7, n, 8, device=device)
    iq_nb[sel] = iq_r[: len(sel)]
    p_nb[7::8] = p_r[: len(sel)]
    return iq_nb, p_nb


def make_idle_off_mix(params, every=16):
    """Mark some slots IDLE / OFF (exercises the early-outs of pullRadioVector, Transceiver.cpp:704-755)."""
    p = params.copy()
    p["type"][every - 1::every] = T_IDLE
    p["type"][every // 2 - 1::every * 4] = T_OFF
    return p


def make_wideband_stream(n_blocks, device="cpu", seed=SEED + 3, m=4, block_len=192):  # noqa: D401
    """BASELINE.json configs[3]: wideband int16 stream for the 4-path channelizer: three GMSK-like carriers
    at the filterbank centre frequencies k/4 (k = 0, 1, 3) + noise.  Returns int16[n_blocks*block_len*m, 2]."""
    device = torch.device(device)
    gen = _gen(seed, device)
    n = n_blocks * block_len * m
    t = torch.arange(n, device=device, dtype=torch.float32)
    out = torch.zeros(n, dtype=torch.complex64, device=device)
    for k, a in ((0, 3000.0), (1, 2000.0), (3, 1500.0)):
        nsym = n // (4 * m) + 2
        sym = torch.randint(0, 2, (nsym,), generator=gen, device=device).to(torch.float32) * 2 - 1
        ph = torch.cumsum(sym, 0) * (math.pi / 2)
        base = torch.polar(torch.full((nsym,), a, device=device), ph)
        base = base.repeat_interleave(4 * m)[:n]
        out = out + base * torch.polar(torch.ones(n, device=device), 2 * math.pi * (k / m) * t)
    noise = torch.randn((n, 2), generator=gen, device=device) * 50.0
    return torch.round(torch.view_as_real(out) + noise).clamp_(-32768, 32767).to(torch.int16).contiguous()


def make_multi_arfcn_wideband(n_slots, device="cpu", carriers=(0, 1, 3), seed=SEED + 5, amp=3000.0, noise=20.0, m=4):
    """BASELINE.json configs[3], end to end: a wideband int16 stream carrying one GSM carrier per filterbank
    channel in `carriers`, each a back-to-back sequence of 4-SPS normal bursts (625 samples per timeslot).
    Each carrier is generated at 4 SPS, resampled 48/65 to the channel rate (what Resampler(65,48) undoes),
    interpolated x4 and shifted to channel k's centre frequency k/m cycles per wideband sample.
    n_slots must be a multiple of 52 (so that every rate holds an integer number of blocks).
    Returns (wide int16[n_blocks*192*m, 2], n_blocks, bits uint8[len(carriers), n_slots, 148], tsc uint8[n_slots])."""
    assert n_slots % 52 == 0
    device = torch.device(device)
    gen = _gen(seed, device)
    n4 = n_slots * 625
    nc = n4 * 48 // 65
    nw = nc * m
    tsc = (torch.arange(n_slots, device=device) % 8).to(torch.uint8)
    W = torch.zeros(nw, dtype=torch.complex64, device=device)
    all_bits = []
    f = torch.fft.fftfreq(nc, device=device)                                # channel-rate bin -> signed index
    kbin = torch.round(f * nc).to(torch.int64)
    for k in carriers:
        bits = normal_burst_bits(n_slots, tsc, gen, device)
        all_bits.append(bits.cpu().numpy())
        wave = modulate_laurent_4sps(bits)                                  # [n_slots, 625]
        phase = torch.rand(n_slots, generator=gen, device=device) * (2 * math.pi)
        s = (wave * torch.polar(torch.full((n_slots,), float(amp), device=device), phase)[:, None]).reshape(-1)
        X = torch.fft.fft(s)
        # keep the nc lowest-frequency bins (48/65 resampling), place them around wideband bin k*nc (x m, shift)
        idx4 = kbin % n4
        Y = X[idx4] * (nc / n4)
        W[(kbin + k * nc) % nw] += Y * m
    w = torch.fft.ifft(W)
    out = torch.view_as_real(w) + torch.randn((nw, 2), generator=gen, device=device) * noise
    wide = torch.round(out).clamp_(-32768, 32767).to(torch.int16).contiguous()
    return wide, nc // 192, np.stack(all_bits), tsc.cpu().numpy()
