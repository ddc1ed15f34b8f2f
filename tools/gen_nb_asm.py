#!/usr/bin/env python3
"""Generator of the hand-placed instruction blocks of the normal-burst kernel (osmo_trx_amd/csrc/trx_kernel_nb.hip).

    python tools/gen_nb_asm.py            # writes osmo_trx_amd/csrc/trx_nb_asm.inc (committed next to this script)

Why generated: hipcc treats an `asm` statement as one opaque instruction -- it pads neither the hazards nor the memory
waits of what is inside, and it pads every boundary between two dependent asm statements with an s_nop of its own.  The
blocks below are therefore whole phases of a burst, and this script CHECKS what hipcc does not, on the sequence it emits:
  * wait states (gfx950 rules as the compiler's own code shows them): packed-fp32 result -> VALU reader 1; VALU-written
    VGPR -> DPP source 2; VALU-written SGPR / VCC -> VALU reader 2; transcendental result -> VALU reader 1; VALU-written
    VGPR -> v_readlane 1; VALU-written SGPR -> v_readlane lane select 4, -> VMEM address 5; x3 / x4 store data 2;
  * every register loaded from LDS is covered by an s_waitcnt lgkmcnt(n) before its first use (LDS returns in order).
Registers: operands the compiler allocates are %[name]; the blocks' temporaries are the fixed VGPRs v64..v127 and SGPRs
s87..s99, declared as clobbers of every statement (NB_ASM_CLOBBERS)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "osmo_trx_amd", "csrc", "trx_nb_asm.inc")

PH_A, PH_M0 = 180, 12
TRANS = ("v_rcp_", "v_sqrt_", "v_log_", "v_exp_", "v_rsq_", "v_sin_", "v_cos_")
NEG_MASKS = [0x447b, 0xc5bb, 0x7488, 0x7709, 0xa75c, 0xf60d, 0x4eb9, 0xdc21]   # TRX_UNIT_NEG_TSC0..7 (trx_device.h)


def regs_of(tok):
    """set of register names an operand token touches: v5, v[4:7], s3, s[2:3], vcc, exec, %[x] (symbolic, one unit)"""
    tok = tok.strip()
    tok = re.sub(r"^-", "", tok)
    tok = re.sub(r"^\|(.*)\|$", r"\1", tok)
    tok = re.sub(r"^abs\((.*)\)$", r"\1", tok)
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    if re.match(r"^[vs]\d+$", tok):
        return {tok}
    if tok in ("vcc", "exec"):
        return {tok}
    if tok.startswith("%["):
        return {tok}
    return set()


class Block:
    def __init__(self, name, sgpr_ops=()):
        self.name = name
        self.ins = []
        self.sgpr_ops = {f"%[{n}]" for n in sgpr_ops}      # operands the compiler allocates in SGPRs (the rest are VGPRs)

    def raw(self, text):
        self.ins.append(text)

    def __call__(self, text):
        self.ins.append(text)

    def label(self, l):
        self.ins.append(l + ":")

    # ---- checks -------------------------------------------------------------------------------------------------
    @staticmethod
    def parse(text):
        t = text.split("//")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            return None
        parts = t.split(None, 1)
        op = parts[0]
        ops = []
        if len(parts) > 1:
            rest = parts[1]
            # operands are separated by commas outside brackets; modifiers (op_sel:.., quad_perm:.., offset:..) follow by spaces
            depth = 0
            cur = ""
            for ch in rest:
                if ch == "[":
                    depth += 1
                if ch == "]":
                    depth -= 1
                if ch == "," and depth == 0:
                    ops.append(cur.strip())
                    cur = ""
                else:
                    cur += ch
            ops.append(cur.strip())
            # strip trailing modifiers from the last operand
            last = ops[-1].split()
            mods = " ".join(last[1:]) if len(last) > 1 else ""
            ops[-1] = last[0] if last else ""
        else:
            mods = ""
        return op, ops, mods

    def check(self):
        """linear hazard / waitcnt check of the emitted sequence (branches: the fall-through order is checked; code after a
        label is checked with the state that reaches it in program order, which is the worst case for the blocks here)"""
        last_w = {}            # reg -> (index in wait states, kind of writer)
        exec_tok = "full"      # symbolic value of EXEC: "full" or the text of the instruction that narrowed it
        w_exec = {}            # VGPR -> EXEC token it was last written under
        pos = 0                # position in wait states
        lds_q = []             # outstanding LDS ops in order: set of dst regs (empty set for writes)
        pending = {}           # reg -> True while its LDS load is outstanding
        errs = []
        for text in self.ins:
            p = self.parse(text)
            if p is None:
                continue
            op, ops, mods = p
            if op == "s_nop":
                pos += int(ops[0]) + 1
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", text)
                if m:
                    n = int(m.group(1))
                    while len(lds_q) > n:
                        for r in lds_q.pop(0):
                            pending.pop(r, None)
                pos += 1
                continue
            is_valu = op.startswith("v_")
            is_dpp = "_dpp" in op or "quad_perm" in mods or "row_" in mods or "wave_" in mods
            is_pk = op.startswith("v_pk_")
            is_trans = op.startswith(TRANS)
            is_lds = op.startswith("ds_")
            is_vmem = op.startswith(("global_", "buffer_", "flat_"))
            is_cmp = op.startswith("v_cmp")
            is_readlane = op.startswith(("v_readlane", "v_readfirstlane"))
            is_store = is_vmem and "store" in op
            # destination / sources
            if is_lds and ("write" in op or op.startswith("ds_add_u32") and "rtn" not in op):
                dsts, srcs = [], ops
            elif is_store:
                dsts, srcs = [], ops
            elif op.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_barrier", "s_sleep")):
                dsts, srcs = [], ops
            else:
                dsts, srcs = ops[:1], ops[1:]
            if is_cmp and op.endswith("_e32"):
                dsts, srcs = ["vcc"], ops[1:]
            if op.startswith(("v_cndmask_b32_e32", "v_addc", "v_subb")) or (op.startswith("v_cndmask_b32") and len(ops) == 3):
                srcs = srcs + ["vcc"]
            if op in ("v_fmac_f32_e32", "v_fmac_f32_dpp", "v_pk_fma_f32") or op.startswith(("v_fmac", "v_mac")) or \
               op.startswith("v_writelane"):
                srcs = srcs + dsts                                   # read-modify-write
            if op.startswith("v_max3") or "_dpp" in op:
                srcs = srcs + []                                     # (dst listed again in the operands when it is also a source)
            rd = set()
            for s in srcs:
                rd |= regs_of(s)
            wr = set()
            for d in dsts:
                wr |= regs_of(d)
            # ---- EXEC: a value written under a narrowed mask is only defined in that mask's lanes
            if (is_valu or is_lds or is_vmem) and not is_readlane:
                for r in rd:
                    if r in w_exec and w_exec[r] != "full" and w_exec[r] != exec_tok and r not in getattr(self, "exec_ok", ()):
                        errs.append(f"{self.name}: `{text.strip()}` under EXEC {exec_tok!r} reads {r}, written under {w_exec[r]!r}")
            # ---- LDS data must have arrived
            for r in rd:
                if r in pending:
                    errs.append(f"{self.name}: `{text.strip()}` reads {r} before its LDS load is waited for")
            # ---- wait states
            for k, s in enumerate(srcs):
                for r in regs_of(s):
                    if r not in last_w:
                        continue
                    r_is_s = (r.startswith("s") and not r.startswith("%")) or r == "vcc" or r in self.sgpr_ops
                    wpos, wkind = last_w[r]
                    gap = pos - wpos - 1
                    need = 0
                    if is_valu and wkind["pk"]:
                        need = max(need, 1)
                    if is_valu and wkind["trans"]:
                        need = max(need, 1)
                    if is_dpp and k == 0 and wkind["valu"] and not r_is_s:
                        need = max(need, 2)
                    if is_valu and wkind["valu"] and r_is_s:
                        need = max(need, 2)
                    if is_readlane and k == 0 and wkind["valu"]:
                        need = max(need, 1)
                    if is_readlane and k == 1 and wkind["valu"]:
                        need = max(need, 4)
                    if is_vmem and wkind["valu"] and r_is_s:
                        need = max(need, 5)
                    if gap < need:
                        errs.append(f"{self.name}: `{text.strip()}` reads {r} {gap} wait state(s) after its writer, needs {need}")
            if is_store and ("dwordx3" in op or "dwordx4" in op):
                self._wide_store = (pos, rd)
            kind = {"pk": is_pk, "trans": is_trans, "valu": is_valu}
            for r in wr:
                last_w[r] = (pos, kind)
                if r.startswith(("v", "%")) and r not in self.sgpr_ops:
                    # a full overwrite under the full mask defines every lane; a write under a narrowed mask leaves the other
                    # lanes as they were: defined only if they were defined before (tracked as the narrower of the two)
                    if exec_tok == "full" or w_exec.get(r) != "full":
                        w_exec[r] = exec_tok
            if "exec" in wr or any(d.strip().startswith("exec") for d in dsts):
                exec_tok = "full" if re.search(r"exec(_lo|_hi)?, -1$", text.strip()) and "exec_" not in text else text.strip()
            if is_lds:
                is_load = "read" in op or "rtn" in op
                lds_q.append(set(wr) if is_load else set())
                if is_load:
                    for r in wr:
                        pending[r] = True
            pos += 1
        if pending:
            errs.append(f"{self.name}: LDS loads still outstanding at the end of the block: {sorted(pending)[:6]}")
        return errs

    def text(self):
        lines = []
        for t in self.ins:
            lines.append(t)
        return lines


def vreg(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def sreg(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


# ------------------------------------------------------------------------------------------------------------------
# block DEC: the /4 decimator of the detection window (downsampleBurst restricted to what the correlation and computeCI
# read, sigProcLib.cpp:1587-1601) + the addition-only correlation's guard (unit_unsafe, trx_device.h).
#   %[pd]   VGPR  LDS byte address of P + PH_M0 + (56 + lane) - 4     (polyphase burst, trx_k4_common.h)
#   %[vd]   VGPR  LDS byte address of D[lane]
#   %[zero] VGPR  0 (address of the wave-uniform tap reads)
#   %[nact] SGPR  15 + len: active lanes
#   %[bad]  SGPR pair out: lanes whose decimated sample fails the guard
# y = sum_k x[4i-15+k] * g[k], product then sum, k ascending, first sum = first product (decimate16_sym<true>); taps 0..7
# only (bitwise symmetric filter).  The sixteen samples arrive in order: four waits, each covering the next four.
# ------------------------------------------------------------------------------------------------------------------
def block_dec(gdec_off):
    b = Block("DEC", ("nact", "bad"))
    X = lambda k: vreg(88 + 2 * k, 2)
    GA, GB = 120, 124
    b(f"s_bfm_b64 exec, %[nact], 0")
    b(f"ds_read_b128 {vreg(GA, 4)}, %[zero] offset:{gdec_off}")
    b(f"ds_read_b128 {vreg(GB, 4)}, %[zero] offset:{gdec_off + 16}")
    for k in range(16):
        off = (((k + 1) & 3) * PH_A + ((k + 1) >> 2)) * 8
        b(f"ds_read_b64 {X(k)}, %[pd] offset:{off}")

    def mul(k):
        kk = k if k < 8 else 15 - k
        base = (GB if (kk >> 2) else GA) + (2 if (kk & 2) else 0)
        sel = "op_sel:[0,1] op_sel_hi:[1,1]" if (kk & 1) else "op_sel:[0,0] op_sel_hi:[1,0]"
        b(f"v_pk_mul_f32 {X(k)}, {X(k)}, {vreg(base, 2)} {sel}")

    def add(k):            # y (in X(0)) += product k
        b(f"v_pk_add_f32 {X(0)}, {X(0)}, {X(k)}")

    b("s_waitcnt lgkmcnt(12)")
    mul(0); mul(1); mul(2); add(1); mul(3); add(2)
    b("s_waitcnt lgkmcnt(8)")
    mul(4); add(3); mul(5); add(4); mul(6); add(5); mul(7); add(6)
    b("s_waitcnt lgkmcnt(4)")
    mul(8); add(7); mul(9); add(8); mul(10); add(9); mul(11); add(10)
    b("s_waitcnt lgkmcnt(0)")
    mul(12); add(11); mul(13); add(12); mul(14); add(13); mul(15); add(14)
    b("s_nop 0")
    add(15)
    b(f"ds_write_b64 %[vd], {X(0)}")
    b("v_min_f32_e64 v90, |v88|, |v89|")
    b("v_max_f32_e64 v91, |v88|, |v89|")
    b("v_ldexp_f32 v90, v90, 17")
    b("v_cmp_lt_f32_e32 vcc, v90, v91")
    b("s_mov_b64 exec, -1")
    b("s_mov_b64 %[bad], vcc")
    b("s_waitcnt lgkmcnt(0)")          # (the store: nothing of this block may be outstanding for the checker; costs nothing behind the guard)
    return b


# ------------------------------------------------------------------------------------------------------------------
# block CORR: correlation of the window against the slot's training sequence without multiplications (corr_unit,
# trx_device.h: every tap is +-1 rotated by k pi/2 under the guard), lane = lag; arg-max input.
#   %[vd]  VGPR  LDS byte address of D[lane]      %[vcz] VGPR  LDS byte address of cz[lane]
#   %[len] SGPR  window length                    %[tsc] SGPR  training sequence 0..7
#   %[nrm] VGPR out: |corr|^2 (0 for lanes >= len)
# The sum is one chain by definition (the reference's order) and a dependent v_pk_add_f32 needs a wait state: the waits
# for the next sample ARE those wait states.  Eight variants (sign / swap patterns are instruction modifiers), entered by
# a computed jump.
# ------------------------------------------------------------------------------------------------------------------
def block_corr(lseq_off):
    b = Block("CORR", ("len", "tsc", "bad"))
    X = lambda k: vreg(88 + 2 * k, 2)
    ACC = vreg(120, 2)
    b(f"v_mov_b64_e32 {ACC}, 0")
    b("s_bfm_b64 exec, %[len], 0")
    for k in range(16):
        b(f"ds_read_b64 {X(k)}, %[vd] offset:{8 * k}")
    b("s_cmp_lg_u64 %[bad], 0")                                    # a sample failed the guard: the multiplying form (below)
    b("s_cbranch_scc1 .Lnb_corr_mul")
    b("s_getpc_b64 s[88:89]")
    b(".Lnb_corr_pc:")
    b("s_mul_i32 s90, %[tsc], .Lnb_corr_v1-.Lnb_corr_v0")
    b("s_add_u32 s88, s88, s90")
    b("s_addc_u32 s89, s89, 0")
    b("s_add_u32 s88, s88, .Lnb_corr_v0-.Lnb_corr_pc")
    b("s_addc_u32 s89, s89, 0")
    b("s_setpc_b64 s[88:89]")
    for t in range(8):
        b(f".Lnb_corr_v{t}:")
        for k in range(16):
            odd, neg = (k & 1), (NEG_MASKS[t] >> k) & 1
            if not odd and not neg:
                mod = ""
            elif not odd and neg:
                mod = " neg_lo:[0,1] neg_hi:[0,1]"
            elif odd and not neg:
                mod = " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
            else:
                mod = " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]"
            b(f"s_waitcnt lgkmcnt({15 - k})")
            b(f"v_pk_add_f32 {ACC}, {ACC}, {X(k)}{mod}")
        b("s_branch .Lnb_corr_join")
    b(".Lnb_corr_join:")
    b("s_add_u32 s90, %[len], 12")
    b("s_bfm_b64 exec, s90, 0")
    b(f"ds_write_b64 %[vcz], {ACC}")
    b("s_mov_b64 exec, -1")                         # (lanes >= len hold the 0 they were initialised with: |corr|^2 = 0 there)
    b(f"v_pk_mul_f32 v[122:123], {ACC}, {ACC}")
    b("s_waitcnt lgkmcnt(0)")
    b("v_add_f32_e32 %[nrm], v123, v122")
    b("s_branch .Lnb_corr_end")
    # ---- cold (about one burst in 3000): convolve_complex() as written (convolve_base.c:28-40, :72-85), taps from LDS:
    # (yr, yi) += (xr hr - xi hi, xr hi + xi hr) as two packed multiplies, one packed add with the low half negated, one accumulate
    b(".Lnb_corr_mul:")
    b("s_lshl_b32 s90, %[tsc], 7")
    b("v_mov_b32_e32 v64, s90")
    H = lambda k: vreg(66 + 2 * (k & 7), 2)
    for half in range(2):
        for q in range(4):
            b(f"ds_read_b128 {vreg(66 + 4 * q, 4)}, v64 offset:{lseq_off + 64 * half + 16 * q}")
        b("s_waitcnt lgkmcnt(0)")
        for k in range(8 * half, 8 * half + 8):
            b(f"v_pk_mul_f32 v[82:83], {X(k)}, {H(k)} op_sel:[0,0] op_sel_hi:[0,1]")
            b(f"v_pk_mul_f32 v[84:85], {X(k)}, {H(k)} op_sel:[1,1] op_sel_hi:[1,0]")
            b("s_nop 0")
            b("v_pk_add_f32 v[86:87], v[82:83], v[84:85] neg_lo:[0,1] neg_hi:[0,0]")
            b("s_nop 0")
            b(f"v_pk_add_f32 {ACC}, {ACC}, v[86:87]")
    b("s_branch .Lnb_corr_join")
    b(".Lnb_corr_end:")
    return b


def check_paths(b):
    """blocks with forward branches: the fall-through order and, for every label, the order that starts at the block's head and
    continues at the label from each branch to it (state at the branch)"""
    errs = b.check()
    lines = b.ins
    for i, t in enumerate(lines):
        m = re.match(r"^s_cbranch_\w+ (\S+)$|^s_branch (\S+)$", t.strip())
        if not m:
            continue
        lab = (m.group(1) or m.group(2)) + ":"
        if lab not in lines:
            continue
        j = lines.index(lab)
        if j < i:
            continue                                                # (a backward branch: the join re-enters checked code)
        v = Block(f"{b.name}[{t.strip()}]", [n[2:-1] for n in b.sgpr_ops])
        v.ins = lines[:i] + lines[j:]
        for e in v.check():
            if e.split(": ", 1)[1] not in [x.split(": ", 1)[1] for x in errs]:
                errs.append(e)
    return errs


def check_variants(b):
    """CORR: check each variant as its own straight line (prefix + variant t + suffix)"""
    errs = []
    lines = b.ins
    # the multiplying form: prefix up to the branch, the cold block, then the join
    im, ie = lines.index(".Lnb_corr_mul:"), lines.index(".Lnb_corr_end:")
    ibr = lines.index("s_cbranch_scc1 .Lnb_corr_mul")
    jj = lines.index(".Lnb_corr_join:")
    v = Block("CORR[mul]", ("len", "tsc", "bad"))
    v.ins = lines[:ibr] + lines[im:ie - 0] + lines[jj:im - 0]
    errs += v.check()
    lines = lines[:im] + lines[ie:]
    i0 = lines.index(".Lnb_corr_v0:")
    j = lines.index(".Lnb_corr_join:")
    per = (j - i0) // 8
    for t in range(8):
        v = Block(f"CORR[tsc {t}]", ("len", "tsc"))
        v.ins = lines[:i0] + lines[i0 + t * per: i0 + (t + 1) * per] + lines[j:]
        errs += v.check()
    return errs


# ------------------------------------------------------------------------------------------------------------------
# block AMAX: wave maximum of |corr|^2 (fastPeakDetect, :1120-1139) and the energyDetect sum (:1573-1585: the lanes = 0
# mod 4 hold the partial sums), two DPP chains interleaved with each other and with the lane-constant / header loads the
# following phases need -- a DPP source written by the instruction in front needs two wait states, and every one of them
# here is an instruction that had to be issued anyway.
#   %[nrm] VGPR |corr|^2   %[ep] VGPR energy partial sums
#   %[m] SGPR out: max     %[es] SGPR out: energy sum      %[bidx] SGPR out: first lane holding the maximum (-1: none)
#   fillers: eight independent single instructions handed in by the caller (strings)
# ------------------------------------------------------------------------------------------------------------------
def block_amax(fill):
    assert len(fill) == 8
    b = Block("AMAX", ("m", "es", "bidx"))
    D = "row_mask:0xf bank_mask:0xf"
    b("v_mov_b32_e32 v88, %[nrm]")
    b(f"v_mov_b32_dpp v89, %[ep] quad_perm:[0,0,0,0] {D}")
    b(fill[0])
    b(f"v_max_f32_dpp v88, v88, v88 quad_perm:[1,0,3,2] {D}")
    b(f"v_add_f32_dpp v89, v89, v89 row_half_mirror {D}")
    b(fill[1])
    b(f"v_max_f32_dpp v88, v88, v88 quad_perm:[2,3,0,1] {D}")
    b(f"v_add_f32_dpp v89, v89, v89 row_mirror {D}")
    b(fill[2])
    b(f"v_max_f32_dpp v88, v88, v88 row_half_mirror {D}")
    b("v_add_f32_dpp v89, v89, v89 row_bcast:15 row_mask:0xa bank_mask:0xf")
    b(fill[3])
    b(f"v_max_f32_dpp v88, v88, v88 row_mirror {D}")
    b("v_add_f32_dpp v89, v89, v89 row_bcast:31 row_mask:0xc bank_mask:0xf")
    b(fill[4])
    b("v_max_f32_dpp v88, v88, v88 row_bcast:15 row_mask:0xa bank_mask:0xf")
    b(fill[5])
    b("v_readlane_b32 %[es], v89, 63")
    b("v_max_f32_dpp v88, v88, v88 row_bcast:31 row_mask:0xc bank_mask:0xf")
    b(fill[6])
    b(fill[7])
    b("v_readlane_b32 %[m], v88, 63")
    b("s_nop 1")
    b("v_cmp_eq_f32_e32 vcc, %[m], %[nrm]")
    b("s_ff1_i32_b64 %[bidx], vcc")
    return b



def fhex(x):
    import struct
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


KAPPA = 6e-6                     # TRX_FAST_KAPPA (trx_device.h): certified-comparison margin of the FAST detector
D_ALL = "row_mask:0xf bank_mask:0xf"


def fast_walk(b, c, p, pos, levels, w="b64"):
    """three scalar instructions per level of the bisection tree: bit position of node (L, p), its "early < late" bit, p = 2p + bit"""
    b(f"s_mov_b32 {p}, 0")
    for lw in range(levels):
        b(f"s_lshl1_add_u32 {pos}, {p}, {2 * ((1 << lw) - 1)}")
        b(f"s_bitcmp1_{w} {c}, {pos}")
        b(f"s_addc_u32 {p}, {p}, {p}")


def careful_walk(b, c, cd, p, pos, levels, unsure_label, w="b64"):
    """the same walk, checking that every node ON THE PATH has a certified decision (cd = c | c >> 1)"""
    b(f"s_mov_b32 {p}, 0")
    for lw in range(levels):
        b(f"s_lshl1_add_u32 {pos}, {p}, {2 * ((1 << lw) - 1)}")
        b(f"s_bitcmp0_{w} {cd}, {pos}")
        b(f"s_cbranch_scc1 {unsure_label}")
        b(f"s_bitcmp1_{w} {c}, {pos}")
        b(f"s_addc_u32 {p}, {p}, {p}")


# ------------------------------------------------------------------------------------------------------------------
# block DETA: detectBurst() behind fastPeakDetect (sigProcLib.cpp:1683-1695): edge gate, computePeakRatio gate (:1541-1571)
# on an estimate with a proven margin, round A of peakDetect()'s bisection (levels 0..4, trx_device.h peak_detect_fast) and
# its certified tree walk.
#   in : %[bidx] %[len] %[czb] SGPR (czb = LDS byte address of cz[0]); %[kr] %[ka] VGPR lane constants (byte offsets of the
#        lane's peak-ratio term and of round A's first tap, relative to &cz[bidx]); %[l16] VGPR 16 * lane; %[k5]..%[k8] SGPR
#        thresh^2 / n; %[c0] VGPR thresh^2 * 1.0001e-5; %[nodes] SGPR pair 0x1555555555555555
#   out: %[st] SGPR 0 miss / 1 found / 2 gate too close to call / 3 an uncertified decision on the path; %[e] SGPR earlyIndex
#        * 512 after round A; %[km] VGPR KAPPA * |corr[bidx]|^2 (round B's margin term)
# The gate: reference |amp| / (sqrtf(avg / num) + 1e-5) < thresh.  Estimate t2 = avg * thresh^2 / num (avg tree-summed: within
# 6e-7; t2 within 1.2e-6 of the reference's squared threshold without its 1e-5).  (rms + 1e-5)^2 >= rms^2, so amp2 < t2 (1 -
# 1.2e-5) is a certain miss; (rms + 1e-5)^2 <= rms^2 (1 + 1e-5) + 1.0001e-5 (2ab <= a^2 + b^2), so amp2 > t2 * 1.000024 + c0 is
# a certain pass; in between (one burst in ~5e4) the burst is left to the general kernel, which rounds as the reference does.
# ------------------------------------------------------------------------------------------------------------------
def block_deta(wa4_off):
    b = Block("DETA", ("bidx", "len", "czb", "k5", "k6", "k7", "k8", "st", "e"))
    X = lambda k: vreg(88 + 2 * k, 2)            # round A samples 0..7
    Y = lambda k: vreg(112 + 2 * k, 2)           # samples 8..15
    b("s_mov_b32 %[st], 0")
    b("s_cmp_lt_i32 %[bidx], 3")                                   # :1683
    b("s_cbranch_scc1 .Lnb_da_end")
    b("s_sub_u32 s88, %[len], 3")
    b("s_cmp_gt_i32 %[bidx], s88")
    b("s_cbranch_scc1 .Lnb_da_end")
    b("s_lshl3_add_u32 s89, %[bidx], %[czb]")                      # &cz[bidx]
    b("s_add_u32 s90, %[len], -1")
    b("v_mov_b32_e32 v64, s89")
    b("v_add_u32_e32 v65, s89, %[kr]")
    b("s_lshl3_add_u32 s90, s90, %[czb]")                          # &cz[len - 1]
    b("v_add_u32_e32 v66, s89, %[ka]")
    b("ds_read_b64 v[68:69], v64")                                 # corr[bidx]
    b("v_mov_b32_e32 v67, s90")
    b("v_mov_b64_e32 v[72:73], 0")
    # (the chip runs at its power limit: what only a few lanes need runs on those lanes -- the eight terms of the gate on lanes 0..7)
    b("s_bfm_b64 exec, 8, 0")
    b("ds_read_b64 v[70:71], v65")                                 # this lane's peak-ratio term (read before cz[len - 1] is zeroed)
    b("s_mov_b64 exec, 1")
    b("ds_write_b64 v67, v[72:73]")                                # interpolatePoint() never reads the last correlation sample (:1105)
    b("s_mov_b64 exec, -1")
    b(f"ds_read_b128 v[76:79], %[l16] offset:{wa4_off}")
    b(f"ds_read_b128 v[80:83], %[l16] offset:{wa4_off + 1024}")
    for u in range(8):
        b(f"ds_read_b64 {X(u)}, v66 offset:{8 * u}")
    b("v_mov_b64_e32 v[84:85], 0")                                 # p0, p1 of the two-chain sums
    b("v_mov_b64_e32 v[86:87], 0")
    # number of in-range terms (:1555-1562) -> thresh^2 / num: scalar instructions, placed in the wait states of the vector ones
    sc = ["s_add_u32 s91, %[bidx], -1", "s_sub_u32 s92, %[len], %[bidx]", "s_min_u32 s91, s91, 4", "s_add_u32 s92, s92, -2",
          "s_min_u32 s92, s92, 4", "s_add_u32 s91, s91, s92", "s_cmp_eq_u32 s91, 8", "s_cselect_b32 s92, %[k8], %[k7]",
          "s_cmp_eq_u32 s91, 6", "s_cselect_b32 s93, %[k6], %[k5]", "s_cmp_ge_u32 s91, 7", "s_cselect_b32 s92, s92, s93"]
    for t in sc[:3]:
        b(t)
    b("s_waitcnt lgkmcnt(11)")
    b("v_pk_mul_f32 v[68:69], v[68:69], v[68:69]")
    b(sc[3])
    b("v_add_f32_e32 v74, v69, v68")                               # m = |corr[bidx]|^2
    b(f"v_mul_f32_e32 %[km], {fhex(KAPPA)}, v74")
    # (the chip runs at its power limit: what only a few lanes need runs on those lanes -- the eight terms of the gate on lanes 0..7)
    b("s_bfm_b64 exec, 8, 0")
    b("v_pk_mul_f32 v[70:71], v[70:71], v[70:71]")
    b(sc[4])
    b("v_add_f32_e32 v75, v71, v70")
    b(sc[5])
    b(sc[6])
    b(f"v_add_f32_dpp v75, v75, v75 quad_perm:[1,0,3,2] {D_ALL}")
    b(sc[7])
    b(sc[8])
    b(f"v_add_f32_dpp v75, v75, v75 quad_perm:[2,3,0,1] {D_ALL}")
    b(sc[9])
    b(sc[10])
    b(f"v_add_f32_dpp v75, v75, v75 row_half_mirror {D_ALL}")     # lanes 0..7: sum of the eight terms
    b(sc[11])
    b("v_mul_f32_e32 v112, s92, v75")                              # t2
    b(f"v_mul_f32_e32 v113, {fhex(1.0 - 1.2e-5)}, v112")
    b(f"v_fmamk_f32 v114, v112, {fhex(1.000024)}, %[c0]")
    b("v_cmp_lt_f32_e32 vcc, v74, v113")
    b("s_cbranch_vccnz .Lnb_da_wait_end")                          # certain miss
    b("v_cmp_gt_f32_e32 vcc, v74, v114")
    b("s_mov_b32 %[st], 2")
    b("s_cbranch_vccz .Lnb_da_wait_end")                           # too close to call
    b("s_mov_b64 exec, -1")
    # ---- round A: interp over taps fl-7 .. fl+8 with this lane's weights, two FMA chains (even / odd taps)
    b(f"ds_read_b128 v[104:107], %[l16] offset:{wa4_off + 2048}")
    b(f"ds_read_b128 v[108:111], %[l16] offset:{wa4_off + 3072}")
    for u in range(8):
        b(f"ds_read_b64 {Y(u)}, v66 offset:{8 * (8 + u)}")
    b("s_waitcnt lgkmcnt(10)")

    def fmas(xs, qa, qb):
        for u in range(0, 8, 2):
            q = qb if (u & 4) else qa
            hp = vreg(q + (2 if (u & 2) else 0), 2)
            b(f"v_pk_fma_f32 v[84:85], {xs(u)}, {hp}, v[84:85] op_sel:[0,0,0] op_sel_hi:[1,0,1]")
            b(f"v_pk_fma_f32 v[86:87], {xs(u + 1)}, {hp}, v[86:87] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    fmas(X, 76, 80)
    b("s_waitcnt lgkmcnt(0)")
    fmas(Y, 104, 108)
    b("s_nop 0")
    b("v_pk_add_f32 v[84:85], v[84:85], v[86:87]")
    b("s_nop 0")
    b("v_pk_mul_f32 v[84:85], v[84:85], v[84:85]")
    b("s_nop 0")
    b("v_add_f32_e32 v88, v85, v84")                               # nv = |interp|^2
    # certified comparison: r = KAPPA (nv + m); sure "early < late" when hi_E < lo_L, sure "early > late" when hi_L < lo_E
    b(f"v_fmamk_f32 v89, v88, {fhex(KAPPA)}, %[km]")
    b("v_sub_f32_e32 v90, v88, v89")
    b("v_add_f32_e32 v91, v88, v89")
    b("s_lshl_b32 s94, %[bidx], 9")
    b(f"v_mov_b32_dpp v90, v90 quad_perm:[1,0,3,2] {D_ALL}")
    b("v_cmp_lt_f32_e64 s[88:89], v91, v90")
    b("s_mov_b32 %[st], 1")
    b("s_lshr_b64 s[90:91], s[88:89], 1")
    b("s_or_b64 s[90:91], s[90:91], s[88:89]")
    b("s_mov_b32 s96, 0x55555555")
    b("s_mov_b32 s97, 0x15555555")
    b("s_and_b64 s[92:93], s[90:91], s[96:97]")
    b("s_cmp_eq_u64 s[92:93], s[96:97]")
    b("s_cbranch_scc0 .Lnb_da_careful")
    fast_walk(b, "s[88:89]", "s95", "s96", 5)
    b(".Lnb_da_walked:")
    b("s_lshl_b32 s95, s95, 5")                                    # off = 32 p - 496; E = (bidx - 1) * 512 + off
    b("s_add_u32 s94, s94, s95")
    b("s_add_u32 %[e], s94, -1008")
    b("s_branch .Lnb_da_end")
    b(".Lnb_da_careful:")
    careful_walk(b, "s[88:89]", "s[90:91]", "s95", "s96", 5, ".Lnb_da_unsure")
    b("s_branch .Lnb_da_walked")
    b(".Lnb_da_unsure:")
    b("s_mov_b32 %[st], 3")
    b("s_branch .Lnb_da_end")
    b(".Lnb_da_wait_end:")
    b("s_mov_b64 exec, -1")
    b("s_waitcnt lgkmcnt(0)")
    b(".Lnb_da_end:")
    return b


# ------------------------------------------------------------------------------------------------------------------
# block DETB: round B of the bisection (levels 5..8 on lanes 0..29, the sixteen possible final positions on lanes 32..47),
# its certified walk, and the interpolated peak value at the final position (peakDetect(), :1141-1186).
#   in : %[e] SGPR earlyIndex * 512; %[kb] VGPR lane constant offB; %[czb] SGPR; %[km] VGPR
#   out: %[st] SGPR 1 / 3; %[toa] SGPR final position * 512; %[xr] %[xi] SGPR interpolated value
# sinc LUT index of a fraction f: taps i <= fl use q = 512 k + f, taps i > fl use q = 512 k + (512 - f), XOR-swizzled
# (trx_tables.h: q ^ ((q >> 4) & 31); swz(512) = 512, so f = 0 needs no special case).  sincv[] sits at LDS offset 0.
# ------------------------------------------------------------------------------------------------------------------
def block_detb():
    b = Block("DETB", ("e", "czb", "st", "toa", "xr", "xi"))
    X = lambda k: vreg(80 + 2 * k, 2)
    W = lambda k: 96 + k
    Y = lambda k: vreg(104 + 2 * k, 2)
    V = lambda k: 120 + k
    b("v_mov_b64_e32 v[64:65], 0")                                 # p0, p1 (lanes outside the round keep 0: |interp|^2 = 0)
    b("v_mov_b64_e32 v[66:67], 0")
    b("v_add_u32_e32 v68, %[e], %[kb]")                            # position of this lane (1/512 symbol)
    b("s_add_u32 s88, %[czb], -56")
    b("s_mov_b32 exec_lo, 0x3fffffff")                             # lanes 0..29: 15 nodes x {early, late}; 32..47: final positions
    b("s_mov_b32 exec_hi, 0xffff")
    b("v_ashrrev_i32_e32 v69, 9, v68")
    b("v_and_b32_e32 v70, 0x1ff, v68")
    b("v_lshl_add_u32 v69, v69, 3, s88")                           # &cz[fl - 7]
    b("v_bfe_u32 v71, v70, 4, 5")
    b("v_sub_u32_e32 v72, 0x200, v70")
    b("v_xor_b32_e32 v71, v71, v70")
    b("v_bfe_u32 v73, v72, 4, 5")
    b("v_lshlrev_b32_e32 v71, 2, v71")                             # taps fl-7 .. fl: sincv[swz(f) + 512 (7 - u)]
    b("v_xor_b32_e32 v72, v73, v72")
    b("v_lshlrev_b32_e32 v72, 2, v72")                             # taps fl+1 .. fl+8: sincv[swz(512 - f) + 512 u]
    for u in range(8):
        b(f"ds_read_b64 {X(u)}, v69 offset:{8 * u}")
        b(f"ds_read_b32 {vreg(W(u))}, v71 offset:{2048 * (7 - u)}")

    def fma2(xs, ws, u):
        b(f"v_pk_fma_f32 v[64:65], {xs(u)}, {vreg(ws(u), 2)}, v[64:65] op_sel:[0,0,0] op_sel_hi:[1,0,1]")
        b(f"v_pk_fma_f32 v[66:67], {xs(u + 1)}, {vreg(ws(u), 2)}, v[66:67] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    # software pipeline: 12..16 loads in flight; each step consumes two taps of the first half and issues two of the second
    for k in range(4):
        b("s_waitcnt lgkmcnt(12)")
        fma2(X, W, 2 * k)
        for u in (2 * k, 2 * k + 1):
            b(f"ds_read_b64 {Y(u)}, v69 offset:{8 * (8 + u)}")
            b(f"ds_read_b32 {vreg(V(u))}, v72 offset:{2048 * u}")
    for k in range(4):
        b(f"s_waitcnt lgkmcnt({12 - 4 * k})")
        fma2(Y, V, 2 * k)
    b("s_mov_b64 exec, -1")
    b("v_pk_add_f32 v[74:75], v[64:65], v[66:67]")                 # interpolated value at this lane's position
    b("s_add_u32 s92, %[e], 497")
    b("v_pk_mul_f32 v[76:77], v[74:75], v[74:75]")
    b("s_mov_b32 %[st], 1")
    b("v_add_f32_e32 v78, v77, v76")
    b(f"v_fmamk_f32 v79, v78, {fhex(KAPPA)}, %[km]")
    b("v_sub_f32_e32 v96, v78, v79")
    b("v_add_f32_e32 v97, v78, v79")
    b("s_nop 0")
    b(f"v_mov_b32_dpp v96, v96 quad_perm:[1,0,3,2] {D_ALL}")
    b("v_cmp_lt_f32_e64 s[88:89], v97, v96")
    b("s_lshr_b32 s90, s88, 1")
    b("s_or_b32 s90, s90, s88")
    b("s_and_b32 s91, s90, 0x15555555")
    b("s_cmp_eq_u32 s91, 0x15555555")
    b("s_cbranch_scc0 .Lnb_db_careful")
    fast_walk(b, "s88", "s94", "s95", 4, "b32")
    b(".Lnb_db_walked:")
    b("s_add_u32 s93, s94, 32")                                    # lane that evaluated the final position
    b("s_lshl1_add_u32 %[toa], s94, s92")                          # E + offB + 512, offB = 2 p - 15
    b("v_readlane_b32 %[xr], v74, s93")
    b("v_readlane_b32 %[xi], v75, s93")
    b("s_branch .Lnb_db_end")
    b(".Lnb_db_careful:")
    careful_walk(b, "s88", "s90", "s94", "s95", 4, ".Lnb_db_unsure", "b32")
    b("s_branch .Lnb_db_walked")
    b(".Lnb_db_unsure:")
    b("s_mov_b32 %[st], 3")
    b(".Lnb_db_end:")
    return b

# ------------------------------------------------------------------------------------------------------------------
# block TAIL: what detectBurst() does behind peakDetect (:1695-1708) for a found burst, the demodulator's set-up and the
# burst's result record.
#   * TOA in 1/512 symbol -> shift and delay-filter row of the straight-line demodulator; fetch of the burst's low-edge
#     tap rows (trx_tables.edge8: 768 bytes, 16 per lane) from L2, as early as possible
#   * computeCI (:1608-1639): S = mean power of the 16 samples at the rounded TOA (tree sum), C = |peak|^2 / ci_den
#   * amp = peak / gain (:1701), toa = position - sync->toa - head (:1704, :1768)
#   * 1 / amp -> the per-lane multiplier VP of the output stage (block DEMOD); RSSI (Transceiver.cpp:741,751)
#   * the result record: field k in lane k of one register (rc, toa, amp.re, amp.im, ci, energy, rssi, flags)
#   in : SGPR toa (position * 512), xr / xi (peak value), t5 ((int)(sync->toa * 512)), hdrb (LDS byte address of the
#        sequence header), e8lo / e8hi (address of tab->edge8), es (energy sum), fsdb (20 log10 full_scale), flags;
#        VGPR l16 (16 * lane), vd (LDS byte address of D[lane]); SGPR pairs modd = 0xaaaa.., m23 = 0xcccc..
#   out: SGPR ok (0: TOA outside the straight-line geometry), nk; VGPR rows (4 registers), vp, rec
# ------------------------------------------------------------------------------------------------------------------
def block_tail():
    b = Block("TAIL", ("toa", "xr", "xi", "t5", "hdrb", "e8lo", "e8hi", "ok", "ssum", "pb", "cb", "db"))
    b("v_mov_b32_e32 v65, %[hdrb]")
    b("ds_read_b128 v[68:71], v65")                                # gain, 1 / gain
    b("s_sub_u32 s88, %[t5], %[toa]")
    b("s_add_u32 s87, s88, 5120")                                  # nk = -(toa512 - t5 - 10 * 512)
    b("s_mov_b32 %[ok], 0")
    b("s_and_b32 s91, s87, 127")
    b("s_lshr_b32 s92, s91, 1")
    b("s_cmp_ge_u32 s91, 2")
    b("s_cselect_b32 s92, s92, 64")                                # delay filter row (64 = none)
    b("s_mul_i32 s93, s92, 768")
    b("s_add_u32 s94, %[e8lo], s93")
    b("s_addc_u32 s95, %[e8hi], 0")
    b("s_abs_i32 s96, %[toa]")                                     # roundf(toa): half away from zero on k / 512
    b("s_add_u32 s96, s96, 256")
    b("global_load_dwordx4 v[120:123], %[l16], s[94:95]")
    b("s_lshr_b32 s96, s96, 9")
    b("s_sub_u32 s97, 0, s96")
    b("s_cmp_lt_i32 %[toa], 0")
    b("s_cselect_b32 s96, s97, s96")
    b("s_lshl_b32 s96, s96, 3")
    b("v_add_u32_e32 v64, s96, %[vd]")                             # &D[rt + lane]: sample ps + lane, ps = start + 1 - N + rt
    b("s_bfm_b64 exec, 16, 0")
    b("ds_read_b64 v[66:67], v64")                                 # computeCI's sixteen samples (:1608-1639)
    b("s_mov_b64 exec, -1")
    b("s_waitcnt lgkmcnt(1)")
    # ---- every lane: amp = peak / gain (:1701), 1 / amp, the output stage's multiplier VP
    b("v_mul_f32_e32 v82, %[xr], v70")                             # peak * (1 / gain): Complex.h:74
    b("v_mul_f32_e32 v83, %[xr], v71")
    b("v_mul_f32_e32 v84, %[xi], v71")
    b("v_mul_f32_e32 v85, %[xi], v70")
    b("v_sub_f32_e32 v86, v82, v84")                               # amp.re
    b("v_add_f32_e32 v87, v83, v85")                               # amp.im
    b("v_mul_f32_e32 v90, v86, v86")
    b("v_mul_f32_e32 v89, v87, v87")
    b("v_add_f32_e32 v89, v89, v90")                               # |amp|^2
    b("v_rcp_f32_e32 v89, v89")
    b("s_mov_b32 s88, 0xaaaaaaaa")
    b("s_mov_b32 s89, 0xaaaaaaaa")
    b("v_mul_f32_e32 v91, v86, v89")                               # 1 / amp = conj(amp) / |amp|^2 (Complex.h:75,144-150)
    b("v_mul_f32_e64 v92, -v87, v89")
    b("v_cndmask_b32_e64 v119, v91, v92, s[88:89]")               # VP[lane & 3] = sx, sy, -sx, -sy (odd lanes: sy)
    b("s_mov_b32 s88, 0xcccccccc")
    b("s_mov_b32 s89, 0xcccccccc")
    b("v_cndmask_b32_e64 v119, v119, -v119, s[88:89]")            # (lanes 2, 3 mod 4: negated)
    # (the burst's record -- C/I, RSSI, amp, toa -- is not made here: its inputs go into lane `slot` of seven registers and 64
    # records are made at once, trx_kernel_nb.hip flush_records; what is per burst is S, computeCI's mean sample power)
    block_demod(b)
    b("s_branch .Lnb_tl_end")
    b(".Lnb_tl_wait_end:")
    b("s_waitcnt vmcnt(0)")                                        # (the tap rows were requested: their registers are free again behind this)
    b(".Lnb_tl_end:")
    return b


# ------------------------------------------------------------------------------------------------------------------
# block DEMOD: demodGmskBurst (:2055-2072) of the usual geometry, fused: delay o decimate as one 24-tap filter at the
# symbol instants (fir24x3, trx_k4_common.h -- the same sums in the same order), rotation / 1/amp / slicer input in
# registers.
#   lanes 0..47: symbols 4 + 3 lane + j with the burst's composite row; lanes 52..55: symbol e = (-lane) & 3, main part of
#   its truncated row (parked in D); lanes 56..59: the same symbols' taps u < 8 (window two symbols earlier)
#   in : SGPR nk, pb (LDS byte address of P[0] of this wave), cb (LDS byte address of comp + 32: tap U0 of row 0), db (LDS
#        byte address of D); VGPR rows (4), vp, l16, kic (lane constant: 8 * first symbol), ktp (lane constant: byte offset
#        of the lane's tap row inside D, -1 for the lanes on the composite row)
#   out: VGPR d0 d1 d2: real((-j)^i z / amp) of the lane's three symbols (the slicer's input)
# ------------------------------------------------------------------------------------------------------------------
def block_demod(b):
    # the straight-line geometry: shift w = nk >> 7 in -36 .. 0 (0 <= TOA <= 9 symbols); other bursts leave with ok = 0 -- 1 / amp
    # and S are done -- and take the general form of the demodulator (trx_kernel_nb.hip, cold)
    ACC = [vreg(64 + 2 * j, 2) for j in range(3)]
    RING = lambda v: vreg(72 + 2 * (v & 15), 2)
    CQ = [104, 108]
    P = [112, 113, 114, 115]
    # S = mean |sample|^2 of computeCI's sixteen samples (lanes 0..15: a tree sum), in the wait states of the address arithmetic
    b("s_bfm_b64 exec, 16, 0")
    b("s_waitcnt lgkmcnt(0)")
    b("v_pk_mul_f32 v[66:67], v[66:67], v[66:67]")
    b("s_ashr_i32 s88, s87, 7")                                    # w
    b("v_add_f32_e32 v81, v67, v66")                               # |sample|^2
    b("s_and_b32 s89, s87, 127")
    b("s_lshr_b32 s90, s89, 1")
    b(f"v_add_f32_dpp v81, v81, v81 quad_perm:[1,0,3,2] {D_ALL}")
    b("s_cmp_ge_u32 s89, 2")
    b("s_cselect_b32 s90, s90, 64")                                # fidx
    b(f"v_add_f32_dpp v81, v81, v81 quad_perm:[2,3,0,1] {D_ALL}")
    b("s_sub_u32 s91, -18, s88")                                   # c = -24 - w + U0: tap U0 of symbol i reads sample 4 i + c
    b("s_and_b32 s92, s91, 3")                                     # ph0
    b(f"v_add_f32_dpp v81, v81, v81 row_half_mirror {D_ALL}")
    b("s_ashr_i32 s93, s91, 2")
    b("s_mul_i32 s94, s92, 180")
    b(f"v_add_f32_dpp v81, v81, v81 row_mirror {D_ALL}")
    b("s_add_u32 s94, s94, s93")
    b("s_lshl3_add_u32 s94, s94, %[pb]")                           # &P[ph0][m = c >> 2] - 12 entries
    b("v_mul_f32_e32 v81, 0x3d800000, v81")                        # S = sum / 16 (row 0)
    b("s_mul_i32 s95, s90, 144")
    b("s_add_u32 s95, s95, %[cb]")                                 # composite row of the burst's delay filter, from tap U0
    b("v_readlane_b32 %[ssum], v81, 0")
    b("s_mov_b64 exec, -1")
    b("s_sub_u32 s96, 0, s88")
    b("s_cmp_gt_u32 s96, 36")
    b("s_cbranch_scc1 .Lnb_tl_wait_end")
    b("s_mov_b32 %[ok], 1")
    b(f"v_add_u32_e32 {vreg(P[0])}, s94, %[kic]")
    for k in range(1, 4):
        # p[k] = p[0] + (k * PH_A + ((ph0 + k) >> 2) * (1 - 4 * PH_A)) entries
        b(f"s_add_u32 s96, s92, {k}")
        b("s_lshr_b32 s96, s96, 2")
        b(f"s_mul_i32 s96, s96, {(1 - 4 * PH_A) * 8}")
        b(f"s_add_u32 s96, s96, {k * PH_A * 8}")
        b(f"v_add_u32_e32 {vreg(P[k])}, s96, {vreg(P[0])}")
    D, NT, NV = 4, 24, 32
    PH0 = 96                                                       # byte offset of PH_M0 entries
    b("s_mov_b32 exec_hi, 0x0ff0ffff")                             # lanes 48..51 and 60..63 have no symbol: off for the filter
    # the first twelve samples do not depend on the tap rows: requested in front of the wait for those (L2 latency)
    for v in range(8 + D):
        b(f"ds_read_b64 {RING(v)}, {vreg(P[v & 3])} offset:{PH0 + 8 * (v >> 2)}")
    b("v_add_u32_e32 v116, %[db], %[l16]")
    b("v_cmp_lt_i32_e32 vcc, -1, %[ktp]")
    b("v_add_u32_e32 v117, %[db], %[ktp]")
    b("v_mov_b32_e32 v118, s95")
    for j_ in range(3):
        b(f"v_mov_b64_e32 {ACC[j_]}, 0")
    b("v_cndmask_b32_e32 v117, v118, v117, vcc")                   # this lane's tap row
    b("s_waitcnt vmcnt(0)")
    b("ds_write_b128 v116, v[120:123]")                            # park the low-edge rows: lane l < 48 holds floats 4l .. 4l+3 of the 8 x 24 block
    fir_core(b, ACC, RING, CQ, P, preload=False)
    # output stage: low-edge symbols = main part (lane 52 + c) + taps u < 8 (lane 56 + c); symbol i wants real((-j)^i z / amp) =
    # z.x * VP[k] + z.y * VP[k - 1], k = i & 3 = (j - lane) & 3: a quad permutation of VP, applied by the DPP operand
    b("s_nop 0")
    b("v_add_f32_dpp v64, v64, v64 row_shl:4 row_mask:0x8 bank_mask:0xf")
    b("v_add_f32_dpp v65, v65, v65 row_shl:4 row_mask:0x8 bank_mask:0xf")
    b(f"v_mul_f32_dpp %[d1], v119, v66 quad_perm:[1,0,3,2] {D_ALL}")
    b(f"v_mul_f32_dpp %[d2], v119, v68 quad_perm:[2,1,0,3] {D_ALL}")
    b(f"v_mul_f32_dpp %[d0], v119, v64 quad_perm:[0,3,2,1] {D_ALL}")
    b(f"v_fmac_f32_dpp %[d1], v119, v67 quad_perm:[0,3,2,1] {D_ALL}")
    b(f"v_fmac_f32_dpp %[d2], v119, v69 quad_perm:[1,0,3,2] {D_ALL}")
    b(f"v_fmac_f32_dpp %[d0], v119, v65 quad_perm:[3,2,1,0] {D_ALL}")
    b("s_mov_b64 exec, -1")


# ------------------------------------------------------------------------------------------------------------------
# block FIRG: the composite filter alone, for the general form of the demodulator (cold: trx_kernel_nb.hip demod_general) --
# lane l < 52 computes the raw sums of symbols ic .. ic + 2 over taps U0 .. U0 + 23 of the burst's composite row.
#   in : SGPR %[nk], %[pb], %[cb] (as DEMOD); VGPR %[kic] 8 * ic of this lane (clamped by the caller)
#   out: VGPR pairs %[a0] %[a1] %[a2]
# ------------------------------------------------------------------------------------------------------------------
def block_firg():
    b = Block("FIRG", ("nk", "pb", "cb"))
    ACC = ["%[a0]", "%[a1]", "%[a2]"]
    RING = lambda v: vreg(72 + 2 * (v & 15), 2)
    CQ = [104, 108]
    P = [112, 113, 114, 115]
    b("s_ashr_i32 s88, %[nk], 7")                                  # w
    b("s_and_b32 s89, %[nk], 127")
    b("s_lshr_b32 s90, s89, 1")
    b("s_cmp_ge_u32 s89, 2")
    b("s_cselect_b32 s90, s90, 64")                                # fidx
    b("s_sub_u32 s91, -18, s88")                                   # c = -24 - w + U0
    b("s_and_b32 s92, s91, 3")
    b("s_ashr_i32 s93, s91, 2")
    b("s_mul_i32 s94, s92, 180")
    b("s_add_u32 s94, s94, s93")
    b("s_lshl3_add_u32 s94, s94, %[pb]")
    b("s_mul_i32 s95, s90, 144")
    b("s_add_u32 s95, s95, %[cb]")
    b(f"v_add_u32_e32 {vreg(P[0])}, s94, %[kic]")
    for k in range(1, 4):
        b(f"s_add_u32 s96, s92, {k}")
        b("s_lshr_b32 s96, s96, 2")
        b(f"s_mul_i32 s96, s96, {(1 - 4 * PH_A) * 8}")
        b(f"s_add_u32 s96, s96, {k * PH_A * 8}")
        b(f"v_add_u32_e32 {vreg(P[k])}, s96, {vreg(P[0])}")
    for j_ in range(3):
        b(f"v_mov_b64_e32 {ACC[j_]}, 0")
    b("v_mov_b32_e32 v117, s95")
    b("s_bfm_b64 exec, 52, 0")
    fir_core(b, ACC, RING, CQ, P, preload=True)
    b("s_mov_b64 exec, -1")
    return b


def fir_core(b, ACC, RING, CQ, P, preload):
    """fir24x3 (trx_k4_common.h): taps outer, a ring of sixteen samples running four ahead, taps four at a time; v117 = the lane's
    tap row.  preload: the first twelve samples are requested here (DEMOD requests them earlier)"""
    D, NT, NV = 4, 24, 32
    PH0 = 96                                                       # byte offset of PH_M0 entries
    if preload:
        for v in range(8 + D):
            b(f"ds_read_b64 {RING(v)}, {vreg(P[v & 3])} offset:{PH0 + 8 * (v >> 2)}")
    b(f"ds_read_b128 {vreg(CQ[0], 4)}, v117")
    for u in range(NT):
        if (u & 3) == 0 and u + 4 < NT:
            b(f"ds_read_b128 {vreg(CQ[((u >> 2) + 1) & 1], 4)}, v117 offset:{16 * ((u >> 2) + 1)}")
        if (u & 3) == 0:
            n_new = 0
            for k in range(4):
                v = u + 8 + D + k
                if v < NV:
                    b(f"ds_read_b64 {RING(v)}, {vreg(P[v & 3])} offset:{PH0 + 8 * (v >> 2)}")
                    n_new += 1
            # everything older than this group's own loads (taps of the next group + four samples) must have arrived
            b(f"s_waitcnt lgkmcnt({n_new + (1 if u + 4 < NT else 0)})")
        ca = CQ[(u >> 2) & 1]
        hp = vreg(ca + (2 if (u & 2) else 0), 2)
        sel = "op_sel:[0,1,0] op_sel_hi:[1,1,1]" if (u & 1) else "op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        for j in range(3):
            b(f"v_pk_fma_f32 {ACC[j]}, {RING(u + 4 * j)}, {hp}, {ACC[j]} {sel}")


def c_string(lines):
    out = []
    for t in lines:
        out.append('\t"' + t.replace("\\", "\\\\").replace('"', '\\"') + '\\n\\t"')
    return "\n".join(out)


def main():
    sys.path.insert(0, ROOT)
    # LDS layout of the kernel's tables (must match trx_kernel_nb.hip; the kernel static_asserts these numbers)
    SINCV_LDS = 4096 + 32
    gdec_off = (SINCV_LDS + 16 * 64 + 65 * 36) * 4
    NB_TABLES_END_OLD = SINCV_LDS + 16 * 64 + 65 * 36 + 16 + 64 + 5 * 64 + 5 * 64      # floats in front of the training-sequence taps
    blocks = {}
    blocks["DEC"] = block_dec(gdec_off)
    lseq_off = NB_TABLES_END_OLD * 4
    blocks["CORR"] = block_corr(lseq_off)
    wa4_off = SINCV_LDS * 4
    blocks["DETA"] = block_deta(wa4_off)
    blocks["DETB"] = block_detb()
    blocks["TAIL"] = block_tail()
    blocks["FIRG"] = block_firg()
    errs = blocks["DEC"].check() + check_variants(blocks["CORR"]) + check_paths(blocks["DETA"]) + check_paths(blocks["DETB"]) + \
        check_paths(blocks["TAIL"]) + check_paths(blocks["FIRG"])
    hdr = ["// trx_nb_asm.inc -- GENERATED by tools/gen_nb_asm.py (hazards and LDS waits checked there); do not edit.",
           f"#define NB_ASM_GDEC_OFF {gdec_off}",
           f"#define NB_ASM_LSEQ_OFF {lseq_off}",
           "#define NB_ASM_CLOBBERS " + ", ".join(f'"v{i}"' for i in range(64, 128)) + ", " +
           ", ".join(f'"s{i}"' for i in range(87, 100)) + ', "vcc", "scc", "memory"']
    for name, b in blocks.items():
        hdr.append(f"#define NB_ASM_{name} \\")
        lines = b.text()
        hdr.append(" \\\n".join('\t"' + t + '\\n\\t"' for t in lines))
    # AMAX is parameterised by the caller's fillers: emitted as a function-like macro
    fill = [f"FILL{i}" for i in range(8)]
    am = block_amax([f"@@{i}@@" for i in range(8)])
    chk = block_amax(["s_nop 0"] * 8)
    errs += chk.check()
    hdr.append("#define NB_ASM_AMAX(" + ", ".join(fill) + ") \\")
    body = []
    for t in am.text():
        m = re.match(r"^@@(\d)@@$", t)
        if m:
            body.append(f"\tFILL{m.group(1)} \"\\n\\t\"")
        else:
            body.append('\t"' + t + '\\n\\t"')
    hdr.append(" \\\n".join(body))
    if errs:
        print("\n".join(errs))
        sys.exit(1)
    open(OUT, "w").write("\n".join(hdr) + "\n")
    print("wrote", os.path.relpath(OUT, ROOT), {k: len([t for t in b.text() if Block.parse(t)]) for k, b in blocks.items()})


if __name__ == "__main__":
    main()
