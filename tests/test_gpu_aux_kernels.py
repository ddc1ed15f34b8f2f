"""GPU parity tests of the kernels either side of the burst path, through the C ABI:
convert_short_float, convolve_real/complex (the reference's own known-answer vectors + the compiled
reference's outputs), Channelizer::rotate, Resampler::rotate, demod-only, TRXD packing."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_oracle import lcg_floats, ref_close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def trx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from osmo_trx_amd import TrxHip
    return TrxHip(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("h_len", [4, 8, 12, 16, 20, 24])
@pytest.mark.parametrize("kind", ["real", "complex"])
def test_convolve_reference_known_answer(trx, golden_dir, h_len, kind):
    """tests/Transceiver52M/convolve_test.c through trxhip_convolve_*_batch."""
    g = np.load(os.path.join(golden_dir, "convolve_golden.npz"))
    x, st = lcg_floats(200)
    h, _ = lcg_floats(50, st)
    start, ln = h_len - 1, 100 - (h_len - 1)
    y = trx.convolve(dev(x.view(np.complex64).reshape(1, 100)), dev(h.view(np.complex64)[:h_len]), start, ln,
                     kind == "complex")
    assert ref_close(g[f"y_ref_{kind}_base_{h_len}"], y.cpu().numpy().view(np.float32).ravel())


@pytest.mark.parametrize("h_len", [1, 5, 16, 20, 40, 64])
def test_convolve_batch_bit_exact_vs_compiled_reference(trx, golden_dir, h_len):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    x = v["x"].view(np.complex64)
    xb = np.stack([x, x[::-1].copy(), x * np.float32(0.5)])
    h = v[f"h_{h_len}"].view(np.complex64)
    start, ln = h_len - 1, 700 - (h_len - 1)
    for kind in ("real", "complex"):
        y = trx.convolve(dev(xb), dev(h), start, ln, kind == "complex").cpu().numpy()
        assert np.array_equal(y[0].view(np.float32), v[f"y_generic_{kind}_{h_len}"])
        for r in (1, 2):
            assert np.array_equal(y[r], O.convolve(xb[r], h, start, ln, kind == "complex"))


@pytest.mark.parametrize("x_len,h_len", [(7000, 16), (40, 5), (6200, 40)])
def test_convolve_long_and_short_vectors_take_the_direct_form(trx, x_len, h_len):
    """Windows that do not fit the LDS-staged kernel (> 48 KB) or are too short for it: the one-thread-per-output form."""
    rng = np.random.default_rng(77)
    xb = (rng.standard_normal((3, x_len)) + 1j * rng.standard_normal((3, x_len))).astype(np.complex64)
    h = (rng.standard_normal(h_len) + 1j * rng.standard_normal(h_len)).astype(np.complex64)
    start, ln = h_len - 1, x_len - (h_len - 1)
    for cplx in (False, True):
        y = trx.convolve(dev(xb), dev(h), start, ln, cplx).cpu().numpy()
        for r in range(3):
            assert np.array_equal(y[r], O.convolve(xb[r], h, start, ln, cplx))


def test_convolve_bounds_check(trx):
    from osmo_trx_amd import TrxHipError
    x = torch.zeros((1, 100), dtype=torch.complex64, device="cuda:0")
    h = torch.zeros(16, dtype=torch.complex64, device="cuda:0")
    with pytest.raises(TrxHipError):
        trx.convolve(x, h, 90, 20, True)          # start + len > x_len  (convolve_base.c:97-103)
    with pytest.raises(TrxHipError):
        trx.convolve(x, h, 3, 20, True)           # no head-room for the taps


def test_convert_short_float(trx, golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    out = trx.convert_short_float(dev(v["cvt_in"])).cpu().numpy()
    assert np.array_equal(out, v["cvt_generic"])
    odd = trx.convert_short_float(dev(v["cvt_in"])[1:1003].clone()).cpu().numpy()      # unaligned tail path
    assert np.array_equal(odd, v["cvt_generic"][1:1003])


def test_channelizer_matches_oracle_blockwise(trx):
    """BASELINE.json configs[3] (reduced): Channelizer(4, 192, 16) over a continuous wideband stream;
    the oracle runs the reference's block-by-block rotate() with carried history."""
    from osmo_trx_amd import synth
    n_blocks = 40
    wide = synth.make_wideband_stream(n_blocks, "cpu")
    got = trx.channelize(wide.to("cuda:0"), n_blocks).cpu().numpy()
    L = O.lib()
    c = L.orc_channelizer_new(4, 192, 16)
    x = wide.numpy().astype(np.float32).view(np.complex64).reshape(n_blocks, 768)
    ref = np.zeros((4, n_blocks * 192), dtype=np.complex64)
    for b in range(n_blocks):
        out = np.zeros((4, 192), dtype=np.complex64)
        blk = np.ascontiguousarray(x[b])
        assert L.orc_channelizer_rotate(c, blk.ctypes.data, 768, out.ctypes.data) == 0
        ref[:, b * 192:(b + 1) * 192] = out
    L.orc_channelizer_free(c)
    assert np.array_equal(got.view(np.float32), ref.view(np.float32))
    # the three carriers sit in channels 0, 1, 3 and channel 2 holds noise only
    p = (np.abs(got) ** 2).mean(axis=1)
    assert p[2] < 0.2 * min(p[0], p[1], p[3])     # rectangular-pulse test carriers leak a little


def test_resampler_65_48_and_decimator(trx, golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    # compiled reference Resampler(65,48): 16 samples of history then 192 in -> 260 out
    buf = v["rs6548_in"].view(np.complex64)
    x = np.zeros((1, 16 + 192), dtype=np.complex64)
    x[0] = buf
    # the C-ABI stream starts with zero history, so feed [history | block] and drop the outputs of the history part
    n_in = 16 + 192 + 32          # pad to a multiple of 48
    xs = np.zeros((2, n_in), dtype=np.complex64)
    xs[0, :208] = buf
    xs[1, :208] = buf[::-1]
    got = trx.resample(dev(xs), 65, 48).cpu().numpy()
    L = O.lib()
    r = L.orc_resampler_new(65, 48, 16, 1.0)
    for row in range(2):
        padded = np.concatenate([np.zeros(16, dtype=np.complex64), xs[row]])
        ref = np.zeros(n_in // 48 * 65, dtype=np.complex64)
        L.orc_resampler_rotate(r, padded[16:].ctypes.data, n_in, ref.ctypes.data, len(ref))
        assert np.array_equal(got[row].view(np.float32), ref.view(np.float32))
    L.orc_resampler_free(r)
    # /4 decimator == sigProcLib's downsampleBurst on the compiled reference's vector
    d = v["dec4_in"].view(np.complex64)[16:].reshape(1, 624)
    got = trx.resample(dev(d), 1, 4).cpu().numpy()
    assert np.array_equal(got[0].view(np.float32), v["dec4_generic"])


def test_standalone_entry_points_against_reference_stage_vectors(trx, golden_dir):
    """The HIP stand-alone entry points of the three FIR stages of detectAnyBurst / demodAnyBurst -- trxhip_resample_batch (1, 4)
    = downsampleBurst, trxhip_convolve_complex_batch = detectBurst's correlation at N = 16 / 40 / 64, and
    trxhip_delay_vector_batch_cf32 = delayVector's 20-tap filter -- against tests/golden/ref_stage_vectors.npz: the outputs of
    the reference's unmodified arch objects for the captured burst and 64 seeded bursts (tests/golden/make_stage_vectors.py).
    Bit for bit."""
    v = np.load(os.path.join(golden_dir, "ref_stage_vectors.npz"))
    cap = np.fromfile(os.path.join(golden_dir, "nb_chunk_tsc7.cfile"), dtype=np.complex64)[:625]
    xs = np.stack([cap] + [b.astype(np.float32).view(np.complex64).reshape(625) for b in v["iq"]])
    T = O.tables()
    # decimator: the C-ABI stream starts from zero history, exactly downsampleBurst's 16 zeros of head-room
    dec = trx.resample(dev(np.ascontiguousarray(xs[:, :624])), 1, 4)
    assert np.array_equal(dec.cpu().numpy().view(np.float32).reshape(65, -1), v["dec"])
    dec_ref = dev(np.ascontiguousarray(v["dec"].view(np.complex64)))
    nb = np.flatnonzero(v["kind"] == 0)
    ab = np.flatnonzero(v["kind"] == 1)
    for i, k in enumerate(nb):                                     # one launch per training sequence
        h = dev(np.ascontiguousarray(T["midamble"][int(v["tsc"][k])]["seq"]))
        c = trx.convolve(dec_ref[k:k + 1], h, 71, 19, True)
        assert np.array_equal(c.cpu().numpy().view(np.float32)[0], v["corr_nb"][i]), k
    c = trx.convolve(dec_ref[torch.from_numpy(ab).to("cuda:0")].contiguous(), dev(np.ascontiguousarray(T["rach"][0]["seq"])), 39, 79, True)
    assert np.array_equal(c.cpu().numpy().view(np.float32).reshape(32, -1), v["corr_ab"])
    c = trx.convolve(dec_ref, dev(np.ascontiguousarray(T["sch"]["seq"])), 63, 93, True)
    assert np.array_equal(c.cpu().numpy().view(np.float32).reshape(65, -1), v["corr_sch"])
    delays = ((v["filt"].astype(np.float32) + np.float32(0.5)) / np.float32(64.0)).astype(np.float32)
    d = trx.delay_vector(dev(xs), dev(delays)).cpu().numpy()
    got = np.concatenate([d[:, a:b] for a, b in v["delay_win"]], axis=1)
    assert np.array_equal(got.view(np.float32).reshape(65, -1), v["delay"])


def test_demod_only_matches_fused(trx):
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(512, "cpu", 4, seed=77)
    cf = torch.view_as_complex(iq.to(torch.float32)).contiguous().to("cuda:0")
    d_p = trx.params_tensor(params)
    res, soft = trx.detect_demod(cf, d_p, sps=4, soft_stride=156, slice_bits=False, exact=True)
    r = trx.results_to_numpy(res)
    ebp = np.stack([r["toa"], r["amp_re"], r["amp_im"], np.zeros(len(r), np.float32)], axis=1).astype(np.float32)
    det = r["rc"] > 0
    p2 = params.copy()
    p2["type"][~det] = O.OFF
    res2, soft2 = trx.demod_only(cf, trx.params_tensor(p2), dev(ebp), exact=True)
    torch.cuda.synchronize()
    assert det.sum() > 450
    assert torch.equal(soft[torch.from_numpy(det).to("cuda:0")], soft2[torch.from_numpy(det).to("cuda:0")])


def test_energy_and_slicer_entry_points(trx):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((7, 625)) + 1j * rng.standard_normal((7, 625))).astype(np.complex64) * 1000
    e = trx.energy_detect(dev(x), 80).cpu().numpy()
    for i in range(7):
        ref = O.lib().orc_energy_detect(np.ascontiguousarray(x[i]).ctypes.data, 625, 80)
        assert abs(e[i] - ref) <= 1e-6 * ref
    s = rng.standard_normal(1000).astype(np.float32) * 2
    out = trx.vector_slicer(dev(s)).cpu().numpy()
    ref = np.zeros_like(s)
    O.lib().orc_vector_slicer(ref.ctypes.data, s.ctypes.data, len(s))
    assert np.array_equal(out, ref)


def test_trxd_packing(trx):
    """proto_trxd.c:36-66 on the device: toa*256 be16, rssi u8, ci centi-bel be16, soft bits round(x*255)."""
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(256, "cpu", 4, seed=5)
    res, soft = trx.detect_demod(iq.to("cuda:0"), trx.params_tensor(params), sps=4)
    pkt = trx.pack_trxd(res, soft, rssi_offset=3.0).cpu().numpy()
    r = trx.results_to_numpy(res)
    s = soft.cpu().numpy()
    L = O.lib()
    for i in range(256):
        toa = L.orc_trxd_toa256(float(r["toa"][i])) & 0xFFFF
        assert (int(pkt[i, 0]) << 8 | int(pkt[i, 1])) == toa
        assert pkt[i, 2] == min(255, max(0, int(float(r["rssi"][i]) + 3.0)))
        ci = L.orc_trxd_ci_cb(float(r["ci"][i])) & 0xFFFF
        assert (int(pkt[i, 3]) << 8 | int(pkt[i, 4])) == ci
        assert pkt[i, 5] == r["tsc"][i] and pkt[i, 6] == r["idle"][i] and pkt[i, 7] == r["nbits_div4"][i]
        u8 = np.zeros(148, dtype=np.uint8)
        if not r["idle"][i]:
            L.orc_trxd_soft_u8(u8.ctypes.data, np.ascontiguousarray(s[i]).ctypes.data, 148)
        assert np.array_equal(pkt[i, 8:], u8)


def test_multi_arfcn_front_end_full_size(trx):
    """BASELINE.json configs[3]: 4-path channelizer over 256k blocks (one continuous stream) + 65/48 resampler
    + per-channel burst detection.  Size-independent properties: (a) overlap-save -- a segment processed alone
    with 16 samples of history per path reproduces the full run bit-for-bit; (b) the first blocks match the
    oracle's block-by-block Channelizer::rotate; (c) the resampled channel keeps feeding the detector."""
    from osmo_trx_amd import synth
    n_blocks = 1 << 18
    wide = synth.make_wideband_stream(n_blocks, "cuda:0")
    full = trx.channelize(wide, n_blocks)
    torch.cuda.synchronize()
    assert full.shape == (4, n_blocks * 192)
    # (a) segment starting at block s with 16 path-samples (= 64 wideband samples) of history
    s, nb = 100_003, 64
    seg = wide[(s * 192 - 16) * 4:((s + nb) * 192) * 4].contiguous()
    # pad the segment to whole blocks: (16 + nb*192) path samples -> use block_len = 16 + nb*192, one "block"
    part = trx.channelize(seg, 1, block_len=16 + nb * 192)
    torch.cuda.synchronize()
    assert torch.equal(part[:, 16 + 15:], full[:, s * 192 + 15:(s + nb) * 192])      # FIR memory is 15 samples
    # (b) oracle on the first 32 blocks
    L = O.lib()
    c = L.orc_channelizer_new(4, 192, 16)
    x = wide[:32 * 768].cpu().numpy().astype(np.float32).view(np.complex64).reshape(32, 768)
    for b in range(32):
        out = np.zeros((4, 192), dtype=np.complex64)
        blk = np.ascontiguousarray(x[b])
        L.orc_channelizer_rotate(c, blk.ctypes.data, 768, out.ctypes.data)
        assert np.array_equal(full[:, b * 192:(b + 1) * 192].cpu().numpy().view(np.float32), out.view(np.float32))
    L.orc_channelizer_free(c)
    # (c) 65/48 resampler on the three active paths: 192 -> 260 per block, continuous
    n_in = 48 * 4096
    rs = trx.resample(full[:, :n_in].contiguous(), 65, 48)
    torch.cuda.synchronize()
    assert rs.shape == (4, n_in // 48 * 65)
    r = L.orc_resampler_new(65, 48, 16, 1.0)
    xin = np.concatenate([np.zeros(16, dtype=np.complex64), full[1, :n_in].cpu().numpy()])
    ref = np.zeros(n_in // 48 * 65, dtype=np.complex64)
    L.orc_resampler_rotate(r, xin[16:].ctypes.data, n_in, ref.ctypes.data, len(ref))
    L.orc_resampler_free(r)
    assert np.array_equal(rs[1].cpu().numpy().view(np.float32), ref.view(np.float32))


def test_multi_arfcn_end_to_end(trx):
    """BASELINE.json configs[3] end to end on the device: wideband int16 -> Channelizer(4,192,16) -> Resampler(65,48)
    -> 625-sample timeslots -> detect + demod, for 3 carriers (filterbank channels 0, 1, 3).  The resampled channel
    stream IS the burst array (zero copy).  Checked: every burst of every carrier is found with a constant TOA
    (the front end's group delay) and its payload bits; the empty channel yields nothing; and the whole chain is
    bit-identical to the oracle's chain (block-by-block Channelizer::rotate, Resampler::rotate, detect/demod)."""
    from osmo_trx_amd import synth
    n_slots = 52 * 8
    wide, n_blocks, bits, tsc = synth.make_multi_arfcn_wideband(n_slots, "cuda:0")
    ch = trx.channelize(wide, n_blocks)                                       # [4, n_blocks*192]
    rs = trx.resample(ch, 65, 48)                                             # [4, n_slots*625] at 4 SPS
    assert rs.shape == (4, n_slots * 625)
    params = np.zeros(n_slots, dtype=O.PARAMS_DTYPE)
    params["type"], params["tsc"], params["max_toa"] = O.TSC, tsc, 20
    d_p = trx.params_tensor(params)
    toas = {}
    for k in range(4):
        bursts = rs[k].view(n_slots, 625)
        res, soft = trx.detect_demod(bursts, d_p, sps=4, full_scale=32767.0, exact=True)
        r = trx.results_to_numpy(res)
        if k == 2:
            assert (r["rc"] > 0).mean() < 0.03                                # no carrier in channel 2: false alarms only
            continue
        # the first slot starts inside the filters' start-up transient and the last one wraps (circular synthesis)
        body = slice(1, n_slots - 1)
        assert (r["rc"][body] == O.TSC).all()
        toas[k] = r["toa"][body]
        assert toas[k].std() < 0.05
        hard = (soft.cpu().numpy()[body] > 0.5).astype(np.uint8)
        # path reversal in the deinterleaver (Channelizer.cpp:43-44): the carrier at +1/4 cycle/sample comes out of
        # channel 3 and the one at -1/4 out of channel 1 (the reference maps them back in radioInterfaceMulti.cpp:92-124)
        ci = {0: 0, 3: 1, 1: 2}[k]
        ber = (hard[:, 3:145] != bits[ci][body][:, 3:145]).mean()
        assert ber < 1e-3, (k, ber)
    assert abs(toas[0].mean() - toas[1].mean()) < 0.05 and abs(toas[0].mean() - toas[3].mean()) < 0.05
    # oracle chain on the first 40 timeslots of channel 1
    L = O.lib()
    nb_o = 40 * 625 * 48 // 65 // 192 + 2
    c = L.orc_channelizer_new(4, 192, 16)
    x = wide[:nb_o * 768].cpu().numpy().astype(np.float32).view(np.complex64).reshape(nb_o, 768)
    chan1 = np.zeros(nb_o * 192, dtype=np.complex64)
    for b in range(nb_o):
        out = np.zeros((4, 192), dtype=np.complex64)
        blk = np.ascontiguousarray(x[b])
        L.orc_channelizer_rotate(c, blk.ctypes.data, 768, out.ctypes.data)
        chan1[b * 192:(b + 1) * 192] = out[1]
    L.orc_channelizer_free(c)
    rr = L.orc_resampler_new(65, 48, 16, 1.0)
    n_in = (len(chan1) // 48) * 48
    padded = np.concatenate([np.zeros(16, dtype=np.complex64), chan1[:n_in]])
    ref = np.zeros(n_in // 48 * 65, dtype=np.complex64)
    L.orc_resampler_rotate(rr, padded[16:].ctypes.data, n_in, ref.ctypes.data, len(ref))
    L.orc_resampler_free(rr)
    assert np.array_equal(rs[1, :40 * 625].cpu().numpy().view(np.float32), ref[:40 * 625].view(np.float32))
    res, soft = trx.detect_demod(rs[1].view(n_slots, 625)[:40].contiguous(), d_p[:40].contiguous(), sps=4,
                                 soft_stride=156, slice_bits=False, full_scale=32767.0, exact=True)
    r = trx.results_to_numpy(res)
    s = soft.cpu().numpy()
    for i in range(40):
        xb = ref[i * 625:(i + 1) * 625]
        rc, e = O.detect_any_burst(xb, int(tsc[i]), 4.0, 4, O.TSC, 20)
        assert rc == r["rc"][i]
        if rc > 0:
            assert e.toa == r["toa"][i] and e.amp[0] == r["amp_re"][i] and e.amp[1] == r["amp_im"][i]
            assert np.array_equal(O.demod_any_burst(xb, rc, 4, e), s[i])


@pytest.mark.parametrize("p,q,block_len", [(65, 48, 192), (65, 96, 192), (52, 75, 300)])
def test_streaming_front_end_any_chunking(trx, p, q, block_len):
    """trxhip_rx_frontend_*: the reference's per-call history carry (Channelizer.cpp:86-88, radioInterfaceMulti.cpp:
    283-300).  Any chunking of the stream must give, bit for bit, the one-piece result; the one-piece result is the
    oracle's (Channelizer blocks + Resampler(p,q) with the RadioInterfaceMulti / RadioInterfaceResamp ratios)."""
    from osmo_trx_amd import synth
    from osmo_trx_amd.trxhip import RxFrontEnd
    n_blocks = 60
    wide = synth.make_wideband_stream(n_blocks, "cuda:0", block_len=block_len)
    fe = RxFrontEnd(trx, block_len, p, q)
    whole = fe.pull(wide, n_blocks)
    fe.reset()
    pieces, pos = [], 0
    for nb in (1, 7, 2, 19, 1, 30):
        seg = wide[pos * block_len * 4:(pos + nb) * block_len * 4].contiguous()
        pieces.append(fe.pull(seg, nb))
        pos += nb
    assert pos == n_blocks
    torch.cuda.synchronize()
    chunked = torch.cat(pieces, dim=1)
    assert torch.equal(chunked, whole)
    # oracle: channelizer block by block, then Resampler(p, q) on the continuous channel stream
    L = O.lib()
    c = L.orc_channelizer_new(4, block_len, 16)
    x = wide.cpu().numpy().astype(np.float32).view(np.complex64).reshape(n_blocks, block_len * 4)
    chan = np.zeros((4, n_blocks * block_len), dtype=np.complex64)
    for b in range(n_blocks):
        out = np.zeros((4, block_len), dtype=np.complex64)
        blk = np.ascontiguousarray(x[b])
        assert L.orc_channelizer_rotate(c, blk.ctypes.data, block_len * 4, out.ctypes.data) == 0
        chan[:, b * block_len:(b + 1) * block_len] = out
    L.orc_channelizer_free(c)
    r = L.orc_resampler_new(p, q, 16, 1.0)
    n_in = n_blocks * block_len
    for k in range(4):
        padded = np.concatenate([np.zeros(16, dtype=np.complex64), chan[k]])
        ref = np.zeros(n_in // q * p, dtype=np.complex64)
        L.orc_resampler_rotate(r, padded[16:].ctypes.data, n_in, ref.ctypes.data, len(ref))
        assert np.array_equal(whole[k].cpu().numpy().view(np.float32), ref.view(np.float32)), k
    L.orc_resampler_free(r)
    fe.close()


@pytest.mark.parametrize("length", [625, 157, 5000])                    # 5000: longer than the LDS-staged form takes
def test_delay_vector_batch_bit_exact(trx, length):
    """trxhip_delay_vector_batch_cf32 == delayVector() (sigProcLib.cpp:1046-1098) per vector, every filter phase."""
    import torch
    rng = np.random.default_rng(31)
    n = 130
    x = (rng.standard_normal((n, length)) + 1j * rng.standard_normal((n, length))).astype(np.complex64) * 500
    delays = np.concatenate([np.arange(64) / 64.0 + 0.003, -np.arange(64) / 64.0 - 2.0, [0.0, 630.0]]).astype(np.float32)
    got = trx.delay_vector(torch.from_numpy(x).to("cuda:0"), torch.from_numpy(delays).to("cuda:0")).cpu().numpy()
    for v in range(n):
        ref = np.zeros(length, dtype=np.complex64)
        xv = np.ascontiguousarray(x[v])
        O.lib().orc_delay_vector(xv.ctypes.data, length, float(delays[v]), ref.ctypes.data)
        assert np.array_equal(got[v].view(np.float32), ref.view(np.float32)), (v, delays[v])
    y = trx.scale_vector(torch.from_numpy(x).to("cuda:0"), 0.25 - 2.0j).cpu().numpy()
    re = x.real * np.float32(0.25) - x.imag * np.float32(-2.0)
    im = x.real * np.float32(-2.0) + x.imag * np.float32(0.25)
    assert np.array_equal(y.real, re) and np.array_equal(y.imag, im)


def test_rx_frontend_time_shards_seeded_mid_stream(trx):
    """SURVEY.md 8e, the channelizer's shard edge: a stream cut into time shards, each on its own front end started
    mid-stream with trxhip_rx_frontend_seed() (the one block preceding the shard is enough: both filters are FIR with 15
    samples of memory, Channelizer.cpp:86-88, radioInterfaceMulti.cpp:283-300), gives -- concatenated -- exactly the
    output of one front end that saw the whole stream; an unseeded second shard does not.  All three resampling ratios."""
    from osmo_trx_amd import synth, trxhip
    for (p, q, bl) in ((65, 48, 192), (65, 96, 192), (52, 75, 300)):
        n_blocks = 4096
        wide = synth.make_wideband_stream(n_blocks, "cuda:0", seed=77 + p, block_len=bl)
        one = trxhip.RxFrontEnd(trx, block_len=bl, p=p, q=q)
        full = one.pull(wide, n_blocks)
        torch.cuda.synchronize()
        cuts = [0, 1000, 1001, 3000, n_blocks]                    # shard edges (one shard is a single block)
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            fe = trxhip.RxFrontEnd(trx, block_len=bl, p=p, q=q)
            if a:
                fe.seed(wide[(a - 1) * bl * 4:a * bl * 4].contiguous(), 1)
            else:
                fe.seed(None, 0)
            parts.append(fe.pull(wide[a * bl * 4:b * bl * 4].contiguous(), b - a))
            fe.close()
        torch.cuda.synchronize()
        cat = torch.cat(parts, dim=1)
        assert torch.equal(cat.view(torch.float32), full.view(torch.float32)), (p, q)
        cold = trxhip.RxFrontEnd(trx, block_len=bl, p=p, q=q)      # the same shard without the seed: differs at its start
        unseeded = cold.pull(wide[1000 * bl * 4:1001 * bl * 4].contiguous(), 1)
        torch.cuda.synchronize()
        assert not torch.equal(unseeded.view(torch.float32), parts[1].view(torch.float32))
        one.close(); cold.close()
