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
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result"]

LIB_SOURCES = ["trx_kernel4.hip", "trx_kernels.hip", "trx_aux_kernels.hip", "trx_sch.hip", "trx_va.hip", "trx_capi.cpp", "trx_tables.cpp"]


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
    deps = srcs + [os.path.join(CSRC, "trx_tables.h"), os.path.join(CSRC, "trx_device.h"),
                   os.path.join(ROOT, "include", "trxhip.h")]
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
    if force or _stale(out, srcs + [os.path.join(CSRC, "trx_tables.h")]):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-DTRX_DIAG"] + COMMON + ["-o", out] + srcs)
    return out


def build_host(force=False):
    """C++ host shim (sigProcLib.h-compatible API over the C ABI) + its test/demo executables."""
    built = []
    shim_src = os.path.join(HOST, "sigProcLib.cpp")
    if not os.path.exists(shim_src):
        return built
    shim_srcs = [shim_src, os.path.join(HOST, "MultiArfcnRx.cpp")]
    shim = os.path.join(LIBDIR, "libtrxsigproc.so")
    deps = shim_srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if force or _stale(shim, deps + [LIB]):
        _run([HIPCC, "-shared"] + COMMON + ["-I", HOST, "-I", os.path.join(ROOT, "include"), "-o", shim] + shim_srcs + [
              "-L", LIBDIR, "-ltrxhip", "-Wl,-rpath,$ORIGIN"])
    built.append(shim)
    for prog in ("sigproc_selftest",):
        src = os.path.join(HOST, prog + ".cpp")
        if not os.path.exists(src):
            continue
        exe = os.path.join(LIBDIR, prog)
        if force or _stale(exe, [src, shim]):
            _run([HIPCC] + COMMON + ["-I", HOST, "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                  "-L", LIBDIR, "-ltrxsigproc", "-ltrxhip", "-Wl,-rpath,$ORIGIN"])
        built.append(exe)
    return built


def build_all(force=False, verbose=False):
    out = [build_lib(force, verbose)]
    out += build_host(force)
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv):
        print("built", os.path.relpath(p, ROOT))
