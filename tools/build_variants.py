#!/usr/bin/env python3
"""Build experimental variants of the library for same-box A/B runs (tools/ab_multi.sh):
     python tools/build_variants.py name1:-DTRX_X1 name2:-DTRX_X2,-DTRX_Y ...
   -> osmo_trx_amd/lib/libtrxhip_<name>.so.  Only csrc/trx_kernel4.hip is recompiled per variant (the macros are read by
   the production kernel and the device header it includes); the other translation units are compiled once to
   build/variants/*.o.  Variants are measurement builds: nothing loads them except TRXHIP_LIB=<path>."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osmo_trx_amd import build as B

OBJ = os.path.join(ROOT, "build", "variants")
os.makedirs(OBJ, exist_ok=True)
FLAGS = ["--offload-arch=gfx950"] + B.COMMON


def cc(src, obj, extra=()):
    B._run([B.HIPCC] + FLAGS + list(extra) + ["-c", "-o", obj, src])
    return obj


def main():
    others = [s for s in B.LIB_SOURCES if s != "trx_kernel4.hip"]
    deps = [os.path.join(B.CSRC, f) for f in ("trx_tables.h", "trx_device.h")] + [os.path.join(ROOT, "include", "trxhip.h")]
    todo = []
    for s in others:
        src, obj = os.path.join(B.CSRC, s), os.path.join(OBJ, s + ".o")
        if B._stale(obj, [src] + deps):
            todo.append((src, obj, ()))
    variants = []
    for a in sys.argv[1:]:
        name, _, fl = a.partition(":")
        # "name:all:-DX,..." -- the flags change a shared header (e.g. the table layout): every translation unit is recompiled
        every = fl.startswith("all:")
        flags = [f for f in fl[4 if every else 0:].split(",") if f]
        obj = os.path.join(OBJ, f"k4_{name}.o")
        todo.append((os.path.join(B.CSRC, "trx_kernel4.hip"), obj, flags))
        rest = []
        for s in others:
            if every:
                o = os.path.join(OBJ, f"{name}_{s}.o")
                todo.append((os.path.join(B.CSRC, s), o, flags))
            else:
                o = os.path.join(OBJ, s + ".o")
            rest.append(o)
        variants.append((name, obj, rest))
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(lambda t: cc(*t), todo))
    for name, obj, rest in variants:
        out = os.path.join(B.LIBDIR, f"libtrxhip_{name}.so")
        B._run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + rest)
        print("built", os.path.relpath(out, ROOT))


main()
