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
Registers: operands the compiler allocates are %[name]; the blocks' temporaries are the fixed VGPRs v88..v127 and SGPRs
s88..s99, declared as clobbers of every statement (NB_ASM_CLOBBERS)."""
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
                dsts, srcs = ["vcc"], ops
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
                    if exec_tok == "full" or r not in w_exec:
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
def block_corr():
    b = Block("CORR", ("len", "tsc"))
    X = lambda k: vreg(88 + 2 * k, 2)
    ACC = vreg(120, 2)
    b(f"v_mov_b64_e32 {ACC}, 0")
    b("s_bfm_b64 exec, %[len], 0")
    for k in range(16):
        b(f"ds_read_b64 {X(k)}, %[vd] offset:{8 * k}")
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
    return b


def check_variants(b):
    """CORR: check each variant as its own straight line (prefix + variant t + suffix)"""
    errs = []
    lines = b.ins
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
    blocks = {}
    blocks["DEC"] = block_dec(gdec_off)
    blocks["CORR"] = block_corr()
    errs = blocks["DEC"].check() + check_variants(blocks["CORR"])
    hdr = ["// trx_nb_asm.inc -- GENERATED by tools/gen_nb_asm.py (hazards and LDS waits checked there); do not edit.",
           f"#define NB_ASM_GDEC_OFF {gdec_off}",
           "#define NB_ASM_CLOBBERS " + ", ".join(f'"v{i}"' for i in range(88, 128)) + ", " +
           ", ".join(f'"s{i}"' for i in range(88, 100)) + ', "vcc", "scc", "memory"']
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
