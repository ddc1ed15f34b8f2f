#!/usr/bin/env python3
"""Static instruction accounting of one kernel from `hipcc -S -gline-tables-only` output: per basic block, the
instruction mix (VALU / SALU / LDS / VMEM / other) and the source lines it came from.
   hipcc --offload-arch=gfx950 --cuda-device-only -S -gline-tables-only -O3 -std=c++17 -ffp-contract=off -Iinclude \
         -o /tmp/k4.s osmo_trx_amd/csrc/trx_kernel4.hip
   python tools/isa_blocks.py /tmp/k4.s 'burst_pull4_kernelILb0ELb0E' [--lines]"""
import re, sys, collections

def classify(op):
    if op.startswith(("v_", )):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"

def main():
    path, pat = sys.argv[1], sys.argv[2]
    by_line = "--lines" in sys.argv
    files = {}
    inside = False
    blocks = []          # (label, Counter, Counter(lines))
    cur = None
    loc = ("?", 0)
    for ln in open(path):
        s = ln.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if not inside:
            if re.match(r"^_Z\w*:", ln) and pat in ln:
                inside = True
                cur = ["entry", collections.Counter(), collections.Counter()]
                blocks.append(cur)
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur = [m.group(1), collections.Counter(), collections.Counter()]
            blocks.append(cur)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[1][c] += 1
        cur[2][(loc[0], loc[1], c)] += 1
    if by_line:
        tot = collections.Counter()
        for b in blocks:
            for (f, l, c), n in b[2].items():
                tot[(f, l, c)] += n
        rows = collections.defaultdict(collections.Counter)
        for (f, l, c), n in tot.items():
            rows[(f, l)][c] += n
        for (f, l) in sorted(rows):
            r = rows[(f, l)]
            print(f"{f}:{l:5d}  valu {r['valu']:5d} salu {r['salu']:5d} lds {r['lds']:4d} vmem {r['vmem']:3d} wait {r['wait']:4d}")
        return
    for lab, cnt, lines in blocks:
        if sum(cnt.values()) == 0:
            continue
        ls = collections.Counter()
        for (f, l, c), n in lines.items():
            ls[(f, l)] += n
        top = ", ".join(f"{f.replace('trx_', '').replace('.hip', '').replace('.h', '')}:{l}x{n}" for (f, l), n in ls.most_common(4))
        print(f"{lab:12s} valu {cnt['valu']:4d} salu {cnt['salu']:4d} lds {cnt['lds']:3d} vmem {cnt['vmem']:2d} wait {cnt['wait']:3d} | {top}")

main()
