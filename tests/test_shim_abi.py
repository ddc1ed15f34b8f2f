"""CPU tests of the drop-in boundary: the sigProcLib shim built against the reference's own headers is
layout- and symbol-compatible with reference-compiled callers.

`sigproc_selftest abi` prints sizeof/offsetof as its own compiler sees them next to what the library it links was
compiled with.  Two builds of that one source exist:
  osmo_trx_amd/lib/sigproc_selftest    stand-alone look-alike headers (host/compat, namespace trxhip_sa)
  oracle/_ref/sigproc_selftest_abi     the reference's unmodified sigProcLib.h / signalVector.h / Vector.h / Complex.h /
                                       BitVector.h + the reference's signalVector.cpp object, linked to libtrxsigproc.so
The second exists only where /root/reference was present at build time (this container); it travels prebuilt."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SA_EXE = os.path.join(ROOT, "osmo_trx_amd", "lib", "sigproc_selftest")
ABI_EXE = os.path.join(ROOT, "oracle", "_ref", "sigproc_selftest_abi")
SHIM = os.path.join(ROOT, "osmo_trx_amd", "lib", "libtrxsigproc.so")
REF = "/root/reference"


@pytest.fixture(scope="module", autouse=True)
def built():
    from osmo_trx_amd import build as trx_build
    trx_build.build_all()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "ref"], stdout=subprocess.DEVNULL)


def report(exe):
    out = subprocess.run([exe, "abi"], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    kv = {}
    for line in out.stdout.strip().splitlines():
        k, v = line.split(" ", 1)
        kv[k] = v
    return kv


def check_layout(kv):
    # CommonLibs/Vector.h:72-76 (three pointers + two function pointers) + signalVector.h:48-51 (two bools + enum)
    assert kv["sizeof_signalVector"] == "48 48"
    assert kv["sizeof_SoftVector"] == "40 40" and kv["sizeof_Vector_float"] == "40 40"
    assert kv["sizeof_complex"] == "8 8"
    # struct estim_burst_params, sigProcLib.h:113-118
    assert kv["sizeof_ebp"] == "20 20" and kv["offsetof_ebp_toa"] == "8 8"
    assert kv["offsetof_ebp_tsc"] == "12 12" and kv["offsetof_ebp_ci"] == "16 16"
    # the constructors pullRadioVector() (Transceiver.cpp:680) and radioVector (radioInterface.cpp:266,274) use
    assert kv["alias_size"].split() == ["625", "alias_start", "0", "headroom_size", "625", "headroom_start", "41"]
    # no GPU / no setup: the reference's error conventions, objects untouched
    assert kv["detect_without_setup"] == "-4"        # -SIGERR_INTERNAL
    assert kv["demod_without_setup"] == "null"


def test_standalone_build_layout():
    kv = report(SA_EXE)
    assert kv["abi"] == "standalone"
    check_layout(kv)


def test_reference_header_build_layout():
    """A TU that includes the reference's unmodified headers, linked against libtrxsigproc.so, agrees with the library
    on every size and offset (static_assert(sizeof(signalVector) == 48) inside, printed values compared here)."""
    if not os.path.exists(ABI_EXE):
        pytest.skip("oracle/_ref/sigproc_selftest_abi not built (no /root/reference at build time)")
    kv = report(ABI_EXE)
    assert kv["abi"] == "reference"
    check_layout(kv)


def nm(path, *flags):
    out = subprocess.run(["nm", "-C"] + list(flags) + [path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return [line.split(None, 2)[-1] if line[:1] != " " else line.split(None, 1)[-1] for line in out.splitlines() if line.strip()]


def test_shim_exports_the_references_symbols():
    """libtrxsigproc.so defines the very (mangled) functions a reference-compiled caller imports -- global-namespace
    signalVector / CorrType / estim_burst_params -- while the stand-alone library's live in trxhip_sa:: and cannot
    be bound by mistake."""
    if not os.path.exists(SHIM):
        pytest.skip("libtrxsigproc.so not built")
    defined = set(nm(SHIM, "-D", "--defined-only"))
    for sym in ("sigProcLibSetup()", "sigProcLibDestroy()",
                "vectorSlicer(float*, float const*, unsigned long)",
                "energyDetect(signalVector const&, unsigned int)",
                "detectAnyBurst(signalVector const&, unsigned int, float, int, CorrType, unsigned int, estim_burst_params*)",
                "demodAnyBurst(signalVector const&, CorrType, int, estim_burst_params*)",
                "detectSCHBurst(signalVector&, float, int, sch_detect_type, estim_burst_params*)",
                "delayVector(signalVector const*, signalVector*, float)",
                "scaleVector(signalVector&, Complex<float>)"):
        assert sym in defined, sym
    sa = set(nm(os.path.join(ROOT, "osmo_trx_amd", "lib", "libtrxsigproc_sa.so"), "-D", "--defined-only"))
    assert "trxhip_sa::detectAnyBurst(trxhip_sa::signalVector const&, unsigned int, float, int, trxhip_sa::CorrType, unsigned int, trxhip_sa::estim_burst_params*)" in sa
    assert not any(s.startswith("detectAnyBurst(") for s in sa)
    # what the shim needs from the caller's side: nothing but members of the reference's own signalVector.cpp
    undefined = [s for s in nm(SHIM, "-D", "--undefined-only") if "signalVector" in s or "Vector<" in s]
    assert undefined and all(s.startswith("signalVector::") for s in undefined), undefined


def test_reference_compiled_caller_links():
    """Compile a caller the way Transceiver.cpp is compiled (reference headers only, no shim headers) and check that
    every sigProcLib symbol it imports is one libtrxsigproc.so defines."""
    if not (os.path.isdir(REF) and os.path.exists(SHIM)):
        pytest.skip("needs /root/reference (container only)")
    import tempfile
    src = r'''
#include "sigProcLib.h"
static void dummy_free(void *) {}
static void *dummy_alloc(size_t) { return 0; }
int caller(complex *buf, float *out)            /* the calls of pullRadioVector(), Transceiver.cpp:680-803 */
{
	signalVector burst(buf, 0, 625, dummy_alloc, dummy_free);
	struct estim_burst_params ebp;
	float pow = energyDetect(burst, 20 * 4);
	int rc = detectAnyBurst(burst, 0, BURST_THRESH, 4, TSC, 3, &ebp);
	if (rc <= 0) return rc;
	SoftVector *rx = demodAnyBurst(burst, (CorrType)rc, 4, &ebp);
	vectorSlicer(out, rx->begin(), 148);
	delete rx;
	return pow > 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "caller.cpp")
        open(c, "w").write(src)
        o = os.path.join(d, "caller.o")
        subprocess.check_call(["g++", "-std=gnu++17", "-O2", "-c", "-I", REF + "/Transceiver52M", "-I", REF + "/CommonLibs",
                               c, "-o", o])
        wanted = [s for s in nm(o, "--undefined-only")
                  if s.split("(")[0] in ("energyDetect", "detectAnyBurst", "demodAnyBurst", "vectorSlicer")]
        assert len(wanted) == 4, wanted
        defined = set(nm(SHIM, "-D", "--defined-only"))
        for s in wanted:
            assert s in defined, s


def test_reference_vector_test_against_standalone_header(tmp_path):
    """The reference's own tests/CommonLibs/VectorTest.cpp, compiled unmodified from where it lies against the stand-alone
    look-alike Vector.h (host/compat), prints the reference's VectorTest.ok (kept as tests/golden/VectorTest.ok): the
    aliasing / ownership behaviour of the stand-alone container type is the reference's (row a2)."""
    src = os.path.join(REF, "tests", "CommonLibs", "VectorTest.cpp")
    if not os.path.exists(src):
        pytest.skip("needs /root/reference (container only)")
    exe = str(tmp_path / "VectorTest")
    subprocess.check_call(["g++", "-std=gnu++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "osmo_trx_amd", "host", "compat"),
                           "-o", exe, src])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "tests", "golden", "VectorTest.ok")).read()
