"""The zero-source-change link recipe: an osmo-trx binary keeps its own sigProcLib.o -- the Tx-side
modulators Transceiver.cpp:107-120,392-394 calls live in the same translation unit as detectAnyBurst, behind the tables
sigProcLibSetup() builds -- and the receive side is interposed at link time with GNU ld --wrap:

    g++ Transceiver.o sigProcLib.o ... -Wl,@osmo_trx_amd/lib/trxwrap.ldflags osmo_trx_amd/lib/libtrxwrap.a -ltrxsigproc -ltrxhip

Checked here without a GPU: a caller compiled against the reference's headers only, plus a stand-in for the reference's
sigProcLib.o that defines every function of sigProcLib.h the caller uses (each records that it ran), linked that way:
  * detectAnyBurst / demodAnyBurst / energyDetect / vectorSlicer / delayVector / scaleVector / detectSCHBurst calls reach
    the shim (no GPU here: its error conventions -SIGERR_INTERNAL / NULL come back, the stand-in's markers do not);
  * modulateBurst / generateDummyBurst still resolve to the binary's own object;
  * sigProcLibSetup() runs the binary's own set-up FIRST (Tx tables), then the GPU context creation, whose failure
    (no device) is what it returns; sigProcLibDestroy() tears both down.
Container only: needs /root/reference for the headers and signalVector.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIBDIR = os.path.join(ROOT, "osmo_trx_amd", "lib")

CALLER = r'''
#include <cstdio>
#include "sigProcLib.h"
static void dummy_free(void *) {}
static void *dummy_alloc(size_t) { return 0; }
int main()
{
	static complex buf[700];
	float out[148];
	const bool up = sigProcLibSetup();                                   /* Transceiver::init(), Transceiver.cpp:207 */
	printf("setup %d\n", (int)up);
	signalVector burst(buf + 41, 0, 625, dummy_alloc, dummy_free);        /* pullRadioVector(), :680-803 */
	struct estim_burst_params ebp;
	printf("energy %g\n", energyDetect(burst, 20 * 4));
	const int rc = detectAnyBurst(burst, 0, BURST_THRESH, 4, TSC, 3, &ebp);
	printf("detect %d\n", rc);
	SoftVector *rx = demodAnyBurst(burst, TSC, 4, &ebp);
	printf("demod %s\n", rx ? "vector" : "null");
	vectorSlicer(out, out, 0);
	signalVector tmp(16);
	printf("delay %s\n", delayVector(&tmp, &tmp, 0.5f) ? "vector" : "null");
	scaleVector(tmp, complex(1.0f, 0.0f));
	printf("sch %d\n", detectSCHBurst(tmp, 4.0f, 4, sch_detect_type::SCH_DETECT_FULL, &ebp));
	BitVector bits(148);
	signalVector *tx = modulateBurst(bits, 8, 4);                        /* Tx side: Transceiver.cpp:392-394 */
	printf("modulate %s\n", tx ? "vector" : "null");
	delete tx;
	delete generateDummyBurst(4, 0);                                     /* :107-120 */
	sigProcLibDestroy();
	return 0;
}
'''

# stands in for the reference's sigProcLib.o: same symbols (compiled from the reference's header), every function says so
STUB = r'''
#include <cstdio>
#include "sigProcLib.h"
bool sigProcLibSetup() { puts("ref:sigProcLibSetup"); return true; }
void sigProcLibDestroy(void) { puts("ref:sigProcLibDestroy"); }
void vectorSlicer(float *, const float *, size_t) { puts("ref:vectorSlicer"); }
float energyDetect(const signalVector &, unsigned) { puts("ref:energyDetect"); return 99.0f; }
int detectAnyBurst(const signalVector &, unsigned, float, int, CorrType, unsigned, struct estim_burst_params *) { puts("ref:detectAnyBurst"); return 99; }
SoftVector *demodAnyBurst(const signalVector &, CorrType, int, struct estim_burst_params *) { puts("ref:demodAnyBurst"); return new SoftVector(156); }
signalVector *delayVector(const signalVector *, signalVector *out, float) { puts("ref:delayVector"); return out; }
void scaleVector(signalVector &, complex) { puts("ref:scaleVector"); }
int detectSCHBurst(signalVector &, float, int, sch_detect_type, struct estim_burst_params *) { puts("ref:detectSCHBurst"); return 99; }
signalVector *modulateBurst(const BitVector &, int, int, bool) { puts("ref:modulateBurst"); return new signalVector(625); }
signalVector *generateDummyBurst(int, int) { puts("ref:generateDummyBurst"); return new signalVector(625); }
'''


def test_wrap_link_routes_rx_to_the_shim_and_tx_to_the_binary(tmp_path):
    need = [os.path.join(LIBDIR, f) for f in ("libtrxwrap.a", "trxwrap.ldflags", "libtrxsigproc.so", "libtrxhip.so")]
    if not os.path.isdir(REF):
        pytest.skip("needs /root/reference (container only)")
    from osmo_trx_amd import build as trx_build
    trx_build.build_all()
    assert all(os.path.exists(p) for p in need), need
    inc = ["-I", REF + "/Transceiver52M", "-I", REF + "/CommonLibs", "-I", REF + "/GSM"]
    objs = []
    for name, text in (("caller", CALLER), ("ref_sigproc_standin", STUB)):
        src = tmp_path / (name + ".cpp")
        src.write_text(text)
        obj = str(tmp_path / (name + ".o"))
        subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-c"] + inc + [str(src), "-o", obj])
        objs.append(obj)
    for ref_src in ("Transceiver52M/signalVector.cpp", "CommonLibs/BitVector.cpp"):     # the reference's own objects
        obj = str(tmp_path / (os.path.basename(ref_src) + ".o"))
        subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-c"] + inc + [os.path.join(REF, ref_src), "-o", obj])
        objs.append(obj)
    exe = str(tmp_path / "wrapped")
    link = ["g++", "-o", exe] + objs + ["-Wl,@" + need[1], need[0], "-L", LIBDIR, "-ltrxsigproc", "-ltrxhip",
                                      "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    lines = out.stdout.split("\n")
    # set-up: the binary's own first, then the GPU context (none here -> false); tear-down reaches the binary's too
    assert lines[0] == "ref:sigProcLibSetup"
    import torch
    if not torch.cuda.is_available():
        assert "setup 0" in lines
    assert "ref:sigProcLibDestroy" in lines
    # Tx side untouched
    assert "ref:modulateBurst" in lines and "modulate vector" in lines and "ref:generateDummyBurst" in lines
    # Rx side: never the binary's own functions
    for f in ("energyDetect", "detectAnyBurst", "demodAnyBurst", "vectorSlicer", "delayVector", "scaleVector", "detectSCHBurst"):
        assert "ref:" + f not in lines, out.stdout
    if not torch.cuda.is_available():                  # the shim's conventions without a context
        assert "detect -4" in lines and "demod null" in lines and "delay null" in lines
