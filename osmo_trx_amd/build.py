"""Build driver: compiles every HIP/C++ source of the product for gfx950, in-tree.

    python -m osmo_trx_amd.build            # libtrxhip.so + host shim + C++ test/demo programs

hipcc cross-compiles without a GPU.  Flags that matter:
  --offload-arch=gfx950   MI355X only, no other targets, no compatibility layers
  -ffp-contract=off       decision-parity with the reference's generic-C operand order
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libtrxhip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
# host-only translation units (they call the HIP runtime API, they hold no kernels): plain g++ against libamdhip64
HOSTCXX = [shutil.which("g++") or "g++", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include")]
HOSTLINK = ["-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result"]

LIB_SOURCES = ["trx_kernel4.hip", "trx_kernels.hip", "trx_aux_kernels.hip", "trx_sch.hip", "trx_va.hip", "trx_capi.cpp", "trx_hostpipe.cpp", "trx_tables.cpp"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    # every file of csrc/ (headers and the .hip files that are #included, e.g. trx_kernel_nb.hip) + the public header
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "trxhip.h")]
    if force or _stale(LIB, deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared"] + COMMON + ["-o", LIB] + srcs
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        out = _run(cmd)
        if verbose:
            print(out)
    return LIB


def build_diag(force=False):
    """Profiling-only variant with the phase-ablation hooks (-DTRX_DIAG); never loaded by the product."""
    out = os.path.join(LIBDIR, "libtrxhip_diag.so")
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    if force or _stale(out, [os.path.join(CSRC, f) for f in os.listdir(CSRC)]):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-DTRX_DIAG"] + COMMON + ["-o", out] + srcs)
    return out


def ref_include_dirs():
    """Include directories of an osmo-trx checkout (TRXHIP_REF_INCLUDE=<checkout>, default /root/reference), or None.
    The product shim is compiled against osmo-trx's own Vector.h / signalVector.h / Complex.h / BitVector.h /
    sigProcLib.h so that its objects have the reference's layout (ABI-true drop-in)."""
    ref = os.environ.get("TRXHIP_REF_INCLUDE", "/root/reference")
    t52, common = os.path.join(ref, "Transceiver52M"), os.path.join(ref, "CommonLibs")
    if all(os.path.exists(os.path.join(d, f)) for d, f in ((t52, "sigProcLib.h"), (t52, "signalVector.h"),
                                                            (t52, "Complex.h"), (common, "Vector.h"), (common, "BitVector.h"))):
        return [t52, common]
    return None


SHIM_SOURCES = ["sigProcLib.cpp", "MultiArfcnRx.cpp", "BurstGatherer.cpp", "trxPullRadioVector.cpp"]


def _build_shim(out, inc_dirs, force):
    srcs = [os.path.join(HOST, f) for f in SHIM_SOURCES]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")] + \
        [os.path.join(HOST, "compat", f) for f in os.listdir(os.path.join(HOST, "compat"))]
    if force or _stale(out, deps + [LIB]):
        inc = []
        for d in inc_dirs + [HOST, os.path.join(ROOT, "include")]:
            inc += ["-I", d]
        # -Wl,-z,undefs is the default for shared objects: the out-of-line members of the reference's signalVector
        # (signalVector.cpp) stay undefined here and are resolved by the process that loads the shim, as in osmo-trx
        # -Bsymbolic-functions: the library's own calls to detectAnyBurst() & co. bind to its own definitions even inside
        # an executable that defines functions of the same name (the --wrap recipe, build_wrap() below)
        _run(HOSTCXX + ["-shared"] + COMMON + ["-pthread"] + inc + ["-o", out] + srcs +
             ["-L", LIBDIR, "-ltrxhip", "-Wl,-rpath,$ORIGIN", "-Wl,-Bsymbolic-functions"] + HOSTLINK)
    return out


WRAPPED = ("sigProcLibSetup", "sigProcLibDestroy", "detectAnyBurst", "demodAnyBurst", "energyDetect", "vectorSlicer",
           "delayVector", "scaleVector", "detectSCHBurst")


def build_wrap(force=False):
    """libtrxwrap.a + trxwrap.ldflags: GNU ld --wrap interposition of the receive-side sigProcLib functions for an osmo-trx
    binary that keeps its own sigProcLib.o (Tx-side modulators + their tables): no source change in osmo-trx.
    trxwrap.cpp defines the nine functions under the reference's names; objcopy renames <mangled> -> __wrap_<mangled> and
    trxwrap_real_X -> __real_<mangled X>.  Needs an osmo-trx checkout (reference headers) like libtrxsigproc.so."""
    ref = ref_include_dirs()
    lib = os.path.join(LIBDIR, "libtrxwrap.a")
    flags = os.path.join(LIBDIR, "trxwrap.ldflags")
    if not ref:
        return [p for p in (lib, flags) if os.path.exists(p)]
    src = os.path.join(HOST, "trxwrap.cpp")
    if force or _stale(lib, [src, os.path.join(HOST, "trxWrap.h")]) or not os.path.exists(flags):
        obj = os.path.join(LIBDIR, "trxwrap.o")
        inc = []
        for d in ref + [HOST]:
            inc += ["-I", d]
        _run([HOSTCXX[0]] + COMMON + inc + ["-c", "-o", obj, src])
        syms = _run(["nm", obj]).splitlines()
        names = {}          # function name -> mangled
        for line in syms:
            parts = line.split()
            m = parts[-1]
            for w in WRAPPED:
                if parts[-2] == "T" and m.startswith("_Z%d%s" % (len(w), w)):
                    names[w] = m
        assert sorted(names) == sorted(WRAPPED), names
        redef = os.path.join(LIBDIR, "trxwrap.redef")
        with open(redef, "w") as f:
            for w, m in names.items():
                f.write("%s __wrap_%s\n" % (m, m))
            for w in ("sigProcLibSetup", "sigProcLibDestroy"):
                real = "_Z%d%s%s" % (len("trxwrap_real_" + w), "trxwrap_real_" + w, "v")
                f.write("%s __real_%s\n" % (real, names[w]))
        _run(["objcopy", "--redefine-syms=" + redef, obj])
        if os.path.exists(lib):
            os.remove(lib)
        _run(["ar", "rcs", lib, obj])
        with open(flags, "w") as f:       # for  g++ ... -Wl,@trxwrap.ldflags
            for w in WRAPPED:
                f.write("--wrap=%s\n" % names[w])
        os.remove(obj)
        os.remove(redef)
    return [lib, flags]


def build_host(force=False):
    """C++ host shim (sigProcLib.h API over the C ABI) in its two builds + the selftest executable.

    libtrxsigproc.so     against osmo-trx's own headers (only when a checkout is present: this container, or
                         TRXHIP_REF_INCLUDE); on a box without one the prebuilt library that travelled is kept.
    libtrxsigproc_sa.so  against host/compat (stand-alone look-alikes in namespace trxhip_sa) + sigproc_selftest.
    The ABI-true selftest (same source, reference headers, the reference's signalVector.cpp compiled where it lies)
    is built by oracle/Makefile into oracle/_ref/ -- reference objects only ever land there."""
    built = []
    if not os.path.exists(os.path.join(HOST, "sigProcLib.cpp")):
        return built
    ref = ref_include_dirs()
    shim = os.path.join(LIBDIR, "libtrxsigproc.so")
    if ref:
        built.append(_build_shim(shim, ref, force))
    elif os.path.exists(shim):
        built.append(shim)
    sa = _build_shim(os.path.join(LIBDIR, "libtrxsigproc_sa.so"), [os.path.join(HOST, "compat")], force)
    built.append(sa)
    src = os.path.join(HOST, "sigproc_selftest.cpp")
    exe = os.path.join(LIBDIR, "sigproc_selftest")
    if force or _stale(exe, [src, sa]):
        _run(HOSTCXX + COMMON + ["-pthread", "-I", os.path.join(HOST, "compat"), "-I", HOST, "-I", os.path.join(ROOT, "include"),
                       "-o", exe, src, "-L", LIBDIR, "-ltrxsigproc_sa", "-ltrxhip", "-Wl,-rpath,$ORIGIN"] + HOSTLINK)
    built.append(exe)
    return built


def build_arch(force=False):
    """libtrxarch.so: the reference's arch seam (convolve_real, convert_short_float, cxvec_fft, ...) under its own names,
    include/trxarch.h."""
    out = os.path.join(LIBDIR, "libtrxarch.so")
    src = os.path.join(HOST, "trxarch.cpp")
    if force or _stale(out, [src, os.path.join(ROOT, "include", "trxarch.h"), LIB]):
        _run(HOSTCXX + ["-shared"] + COMMON + ["-pthread", "-I", os.path.join(ROOT, "include"), "-o", out, src,
                        "-L", LIBDIR, "-ltrxhip", "-Wl,-rpath,$ORIGIN"] + HOSTLINK)
    return out


def build_all(force=False, verbose=False):
    out = [build_lib(force, verbose)]
    out += build_host(force)
    out += build_wrap(force)
    out.append(build_arch(force))
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv):
        print("built", os.path.relpath(p, ROOT))
