#!/usr/bin/env python3
"""Bank conflicts of ds_read_b64 on the 4-SPS kernel's polyphase LDS layout (trx_kernel4.hip) when lane l reads sample
stride * l + c: entry ((s & 3) * PH_A + PH_M0 + (s >> 2)) of 8 bytes, banks (address / 4) mod 64, lane groups {0-31} and
{32-63} (MI355X_MICROARCH.md, LDS).  Answers "why does the exact demodulator's delay filter run 12 outputs per lane on 52
lanes and not 10 on 63" (VERDICT r3 item 4b): only strides that are a multiple of 4 keep every lane of an instruction in one
phase array and are conflict-free; every other stride is 2-way conflicted (or worse) for every PH_A, which doubles the LDS
time of the filter's 25-27 reads per lane and more than eats the 17 % fewer multiply-adds.
   python tools/lds_stride_conflicts.py"""


def conflicts(PH_A, stride, PH_M0=12):
    worst, tot, cnt = 0, 0, 0
    for c in range(4):
        for j in range(4):
            for g in range(2):
                banks = {}
                for l in range(32 * g, 32 * g + 32):
                    s = stride * l + c + j + 48
                    a = (s & 3) * PH_A + PH_M0 + (s >> 2)
                    for d in (2 * a, 2 * a + 1):
                        banks.setdefault(d % 64, set()).add(d)
                m = max(len(v) for v in banks.values())
                worst = max(worst, m)
                tot += m
                cnt += 1
    return worst, tot / cnt


print("stride  PH_A=180 (worst, mean LDS cycles per group)   best PH_A in 164..229")
for stride in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16):
    best = min(((pa,) + conflicts(pa, stride) for pa in range(164, 230)), key=lambda t: (t[1], t[2]))
    print(f"{stride:6d}  {conflicts(180, stride)}   {best}")
