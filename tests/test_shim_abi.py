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
    # struct trx_ul_burst_ind (proto_trxd.h:24-37), the output type of pullRadioVector(): 444 floats, then nbits / fn / tn,
    # three doubles (rssi, toa, noise), bool idle, enum Modulation, tss, tsc, float ci -- same on both sides of the library
    for k, want in (("sizeof_trx_ul_burst_ind", 1832), ("offsetof_bi_rx_burst", 0), ("offsetof_bi_nbits", 1776), ("offsetof_bi_fn", 1780),
                    ("offsetof_bi_tn", 1784), ("offsetof_bi_rssi", 1792), ("offsetof_bi_toa", 1800), ("offsetof_bi_noise", 1808),
                    ("offsetof_bi_idle", 1816), ("offsetof_bi_modulation", 1820), ("offsetof_bi_tss", 1824), ("offsetof_bi_tsc", 1825),
                    ("offsetof_bi_ci", 1828), ("sizeof_enum_Modulation", 4)):
        assert kv[k] == f"{want} {want}", (k, kv[k])


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


def struct_fields(text, name):
    """[(type, member)] of `struct name { ... }` in a C header: declarations in order, comments stripped."""
    import re
    body = re.search(r"struct\s+" + name + r"\s*\{(.*?)\};", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if decl:
            t, m = decl.rsplit(" ", 1)
            out.append((t, m))
    return out


def test_trx_ul_burst_ind_declaration_matches_the_reference_header():
    """The adapter's fallback declaration of struct trx_ul_burst_ind / enum Modulation (host/trxPullRadioVector.h, used where
    libosmocore's <osmocom/core/endian.h> -- which the reference's proto_trxd.h includes -- is not installed) against the
    reference header where it lies: same members, same types, same order, same enumerators."""
    hdr = os.path.join(REF, "Transceiver52M", "proto_trxd.h")
    if not os.path.exists(hdr):
        pytest.skip("needs /root/reference (container only)")
    import re
    ref = open(hdr).read()
    mine = open(os.path.join(ROOT, "osmo_trx_amd", "host", "trxPullRadioVector.h")).read()
    assert struct_fields(mine, "trx_ul_burst_ind") == struct_fields(ref, "trx_ul_burst_ind")
    assert len(struct_fields(ref, "trx_ul_burst_ind")) == 12

    def enumerators(text):
        body = re.search(r"enum\s+Modulation\s*\{(.*?)\};", text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [e.strip() for e in body.split(",") if e.strip()]
    assert enumerators(mine) == enumerators(ref) == ["MODULATION_GMSK", "MODULATION_8PSK"]
    assert re.search(r"#define\s+MAX_RX_BURST_BUF_SIZE\s+444", ref) and re.search(r"#define\s+MAX_RX_BURST_BUF_SIZE\s+444", mine)


def test_pull_radio_vector_adapter_is_exported_with_the_reference_type():
    """libtrxsigproc.so exports trxPullRadioVector(BurstGatherer&, RxChanState&, unsigned long, trx_ul_burst_ind*) with the
    GLOBAL struct trx_ul_burst_ind (what a reference-compiled Transceiver.o passes)."""
    if not os.path.exists(SHIM):
        pytest.skip("libtrxsigproc.so not built")
    defined = set(nm(SHIM, "-D", "--defined-only"))
    assert "trxPullRadioVector(BurstGatherer&, RxChanState&, unsigned long, trx_ul_burst_ind*)" in defined
