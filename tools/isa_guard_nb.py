#!/usr/bin/env python3
"""What the compiler made of the normal-burst kernel's burst loop (csrc/trx_kernel_nb.hip), checked on its assembly:

    python tools/isa_guard_nb.py [--keep out.s]

The loop is C++ around hand-placed blocks, and twice in round 6 a harmless-looking source change made hipcc put
`s_waitcnt vmcnt(0)` between the next burst's prefetch loads (an exec-masked tenth load; a 64-bit vector address built in
one of the destination registers) -- 3-4 % of the headline each time, invisible to every test.  Checked here (and by
tests/test_isa_guard_cpu.py on every CPU run):
  * the ten prefetch loads of the loop are issued back to back: no s_waitcnt between the first and the last of them;
  * the work ticket is taken with a partial wait (lgkmcnt(5)), not a drain of the converted samples' LDS writes;
  * no spills, no scratch, 128 VGPRs (4 waves per SIMD), and the code size.
Prints a JSON summary; exit status 1 on a violation."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osmo_trx_amd import build as B   # noqa: E402

KERNEL = "_Z15nb_pull4_kernel"


def assembly(keep=None):
    out = keep or os.path.join(tempfile.mkdtemp(prefix="isa_guard_"), "k4.s")
    flags = [f for f in B.COMMON if not f.startswith("-W")]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950"] + flags + ["-w", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    "-o", out, os.path.join(B.CSRC, "trx_kernel4.hip")], check=True, capture_output=True)
    return open(out).read()


def check(text):
    lines = text.splitlines()
    start = next(i for i, t in enumerate(lines) if re.match(KERNEL + r"\w*:", t))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    errs = []
    # ---- the prefetch of the loop: the LAST run of non-temporal dword loads (the first one is in front of the loop)
    nt = [i for i, t in enumerate(body) if re.match(r"\s*global_load_dword v\d+, .* nt\s*$", t)]
    groups, cur = [], []
    for i in nt:
        if cur and i - cur[-1] > 8:
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    if not groups or len(groups[-1]) != 10:
        errs.append(f"expected a last group of 10 non-temporal prefetch loads, found groups of {[len(g) for g in groups]}")
    else:
        g = groups[-1]
        between = [body[i].strip() for i in range(g[0], g[-1]) if "s_waitcnt" in body[i]]
        # and in front of the group, behind the address arithmetic: a wait for vmcnt(0) there stalls on the stores just issued
        before = [body[i].strip() for i in range(max(0, g[0] - 12), g[0]) if re.search(r"s_waitcnt.*vmcnt\(0\)", body[i])]
        if between:
            errs.append(f"s_waitcnt between the loop's prefetch loads: {between}")
        if before:
            errs.append(f"s_waitcnt vmcnt(0) right in front of the loop's prefetch loads: {before}")
        if any(re.search(r"v\[\d+:\d+\], off", body[i]) for i in g):
            errs.append("prefetch loads with a 64-bit vector address (expected the scalar-base form)")
    if not any(re.search(r"s_waitcnt lgkmcnt\(5\)", t) for t in body):
        errs.append("no `s_waitcnt lgkmcnt(5)`: the work ticket is taken behind a full drain of the LDS")
    # ---- resources (the kernel's metadata record)
    md = text[text.index(".name:           " + KERNEL):]
    md = md[:md.index("  - .agpr_count") if "  - .agpr_count" in md else len(md)]
    res = {k: int(re.search(rf"\.{k}:\s+(\d+)", md).group(1)) for k in
           ("private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count")}
    if res["sgpr_spill_count"] or res["vgpr_spill_count"] or res["private_segment_fixed_size"]:
        errs.append(f"spills / scratch: {res}")
    if res["vgpr_count"] > 128:
        errs.append(f"{res['vgpr_count']} VGPRs: fewer than 4 waves per SIMD")
    n_ins = sum(1 for t in body if re.match(r"\s+[a-z_0-9]+(\s|$)", t) and not t.strip().startswith((".", ";")))
    return errs, {"kernel": "nb_pull4_kernel", **res, "instructions": n_ins,
                  "prefetch_groups": [len(g) for g in groups]}


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    errs, info = check(assembly(keep))
    info["violations"] = errs
    print(json.dumps(info, indent=1))
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
