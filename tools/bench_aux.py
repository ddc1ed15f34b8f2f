#!/usr/bin/env python3
"""Every kernel of the library other than the timed configs[1] launch, on one representative size each, with the
algorithmic bytes (or work) of that launch stated -- the program profiled by tools/run_profiles_aux.sh so that each
DESIGN.md number has a rocprofv3 row behind it (profiles/README.md).  Prints one JSON line per kernel:
  {"kernel": <symbol prefix>, "what": ..., "units": N, "unit": ..., "algo_bytes": B, "event_ms": t}"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from osmo_trx_amd import TrxHip, synth, trxhip

REPS = int(os.environ.get("REPS", 5))
N = int(os.environ.get("N_BURSTS", 1 << 20))
trx = TrxHip(0)
dev = "cuda:0"


def timeit(fn):
    keep = [fn(), fn()]                                 # two live outputs: the timed calls allocate nothing new
    torch.cuda.synchronize()
    del keep
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS


def report(kernel, what, units, unit, algo_bytes, ms):
    print(json.dumps({"kernel": kernel, "what": what, "units": units, "unit": unit, "algo_bytes": algo_bytes,
                      "event_ms": round(ms, 4)}), flush=True)


# ---- hot kernel on the other workloads (burst_pull4_kernel<false, false, true>, the COMMON instantiation, unless stated)
def pull(name, iq, params, stride=148, exact=False, kernel="burst_pull4_kernel<false, false, true>"):
    n = iq.shape[0]
    dp = trx.params_tensor(params)
    res = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    soft = torch.empty((n, stride), dtype=torch.float32, device=dev)
    ms = timeit(lambda: trx.detect_demod(iq, dp, sps=4 if iq.shape[1] > 200 else 1, soft_stride=stride, results=res, soft=soft, exact=exact))
    report(kernel, name, n, "bursts", n * (iq.shape[1] * 4 + 8 + 32 + stride * 4), ms)
    return res, soft, dp


iq, p, _ = synth.make_normal_bursts(N, dev, 4)
res, soft, dp = pull("configs[1] NB max_toa 3, exact demodulator", iq, p, exact=True, kernel="burst_pull4_kernel<false, true, true>")

# ---- TRXD packers on that launch's output
meta = np.zeros(N, dtype=trxhip.TRXD_META_DTYPE)
meta["fn"] = np.arange(N) % 2715648
meta["tn"] = np.arange(N) & 7
meta["version"] = 1
d_meta = torch.from_numpy(meta.view(np.uint8).reshape(-1, 8).copy()).to(dev)
ms = timeit(lambda: trx.pack_trxd_wire(res, dp, soft, d_meta))
report("pack_trxd_wire16_kernel", "TRXD v1 datagrams, 160-byte rows", N, "bursts", N * (32 + 8 + 8 + 592 + 160 + 2), ms)
ms = timeit(lambda: trx.pack_trxd(res, soft))
report("pack_trxd_kernel", "156-byte device records", N, "bursts", N * (32 + 592 + 156), ms)

# ---- Viterbi alternative
x = torch.view_as_complex(iq.to(torch.float32).contiguous())
ms = timeit(lambda: trx.demod_va(x, dp))
report("va_demod_kernel", "use_va demodulation, NB", N, "bursts", N * (625 * 8 + 8 + 156 * 4 + 4), ms)
del x

iq, p, _ = synth.make_access_bursts(N, dev)
pull("configs[2] RACH max_toa 63", iq, p)
iq, p, _ = synth.make_access_bursts(N, dev, ext=True)
pull("configs[2] EXT_RACH (TS0/1/2) max_toa 63", iq, p)
iq, p = synth.make_mixed_bursts(N, dev)
pull("configs[4] share: 7:1 NB:RACH", iq, p)
iq, p, _ = synth.make_edge_bursts(N, dev)
pull("EDGE 8-PSK, 444 soft bits", iq, p, stride=444, kernel="burst_pull4_kernel<false, false, false>")
iq1, p1, _ = synth.make_normal_bursts(N, dev, 1, burst_len=156)
pull("configs[0] geometry: NB 1 SPS, 156 samples", iq1, p1, kernel="burst_pull_kernel<1, false, 3>")
del iq, iq1

# ---- front end (configs[3]): Channelizer(4, 192, 16) over 256k blocks, Resampler(65, 48) on the 4 channel streams
nb = 1 << 18
wide = synth.make_wideband_stream(nb, dev)
ms = timeit(lambda: trx.channelize(wide, nb))
report("channelize_kernel", "Channelizer::rotate, 768-sample blocks", nb, "blocks", nb * (768 * 4 + 4 * 192 * 8), ms)
ch = trx.channelize(wide, nb)
n_in = (ch.shape[1] // 48) * 48
xch = ch[:, :n_in].contiguous()
ms = timeit(lambda: trx.resample(xch, 65, 48))
report("resample_kernel", "Resampler(65,48)::rotate, 4 channels", nb, "blocks", 4 * n_in * 8 + 4 * (n_in // 48 * 65) * 8, ms)
fe = trxhip.RxFrontEnd(trx)
ms = timeit(lambda: fe.pull(wide, nb))
report("frontend_fused_kernel", "rx_frontend_pull: channelizer + resampler in one pass (streaming, carried history)", nb, "blocks",
       nb * 768 * 4 + 4 * (n_in // 48 * 65) * 8, ms)
del wide, ch, xch

# ---- arch kernels
nv = 1 << 16
xv = torch.view_as_complex(torch.randn((nv, 700, 2), device=dev))
h16 = torch.view_as_complex(torch.randn((16, 2), device=dev))
ms = timeit(lambda: trx.convolve(xv, h16, 40, 625, False))
report("convolve_lds_kernel<false>", "convolve_real, 16 taps, 625 outputs", nv, "vectors", nv * (700 * 8 + 625 * 8), ms)
ms = timeit(lambda: trx.convolve(xv, h16, 40, 625, True))
report("convolve_lds_kernel<true>", "convolve_complex, 16 taps, 625 outputs", nv, "vectors", nv * (700 * 8 + 625 * 8), ms)
s16 = torch.randint(-32768, 32767, (1 << 28,), dtype=torch.int16, device=dev)
ms = timeit(lambda: trx.convert_short_float(s16))
report("convert_short_float_kernel", "int16 -> fp32", s16.numel(), "values", s16.numel() * 6, ms)
del s16, xv

# ---- delayVector / energyDetect / vectorSlicer as stand-alone calls
nd = 1 << 16
xd = torch.view_as_complex(torch.randn((nd, 625, 2), device=dev))
dl = (torch.rand(nd, device=dev) * 20 - 10)
ms = timeit(lambda: trx.delay_vector(xd, dl))
report("delay_vector_kernel", "delayVector, 625 samples", nd, "vectors", nd * 625 * 16, ms)
ms = timeit(lambda: trx.energy_detect(xd, 80))
report("energy_detect_kernel", "energyDetect(burst, 80)", nd, "bursts", nd * 80 * 8, ms)
sf = torch.rand((nd, 148), device=dev) * 2 - 1
ms = timeit(lambda: trx.vector_slicer(sf))
report("vector_slicer_kernel", "vectorSlicer", sf.numel(), "values", sf.numel() * 8, ms)
del xd

# ---- SCH search (MS side): FULL window on 625-sample buffers, BUFFER search on 60000-sample buffers
ns = 1 << 14
xs = torch.view_as_complex(torch.randn((ns, 625, 2), device=dev))
ms = timeit(lambda: trx.detect_sch(xs, trxhip.SCH_DETECT_FULL))
report("sch_detect_kernel", "detectSCHBurst FULL, 625-sample buffers", ns, "buffers", ns * 625 * 8, ms)
nbuf = 256
xb = torch.view_as_complex(torch.randn((nbuf, 60000, 2), device=dev))
ms = timeit(lambda: trx.detect_sch(xb, trxhip.SCH_DETECT_BUFFER))
report("sch_detect_kernel", "detectSCHBurst BUFFER (12 frames), 60000-sample buffers", nbuf, "buffers", nbuf * 60000 * 8, ms)
