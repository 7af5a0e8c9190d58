#!/usr/bin/env python3
"""Emits anonymous-credit-tokens_amd/csrc/fe25519_gen.inc: fe_mul / fe_sq bodies for the 9-limb representation of GF(2^255-19)
(radix 2^(255/9): limb i sits at bit ceil(85 i / 3), widths 29 28 28 29 28 28 29 28 28).
  * a_i b_j lands in column i + j, times 2 when the two limbs' half-open positions add up past the column's (i = 1 mod 3 with
    j != 0 mod 3, or j = 1 mod 3 with i = 2 mod 3): three pre-doubled limbs per operand instead of the 19x / 2x preparation
    of the 10-limb form;
  * the product is left UNREDUCED over 17 columns; the high half (columns 9..16, weight 2^255 = 19) is summed first, each column
    handing bits 32.. of its sum to the next as (upper register) * 2^(32 - width) -- a multiply-accumulate with an inline
    constant, after which the lower register is the column's residue with no shift or mask --, and low column k then starts from
    its carry-in, adds 19 * residue_k as one more multiply-accumulate and its own products: 81 + 7 + 9 multiply-accumulates per
    multiplication (10-limb: 100 + 9 preparations), 45 + 7 + 9 per squaring;
  * every column is ONE asm statement of v_mad_u64_u32 whose first addend is the carry of the column below (fe25519.h explains why).
Operand budget (fe25519.h): every term is at most phi gamma 2^58 with phi, gamma the operands' limb sizes relative to 2^width, a
column has at most nine term-equivalents, so phi gamma <= 7 keeps every column below 2^64."""
POS = [-(-85 * i // 3) for i in range(10)]
W = [POS[i + 1] - POS[i] for i in range(9)]

def needs2(i, j):
    return (i % 3 == 1 and j % 3 != 0) or (j % 3 == 1 and i % 3 == 2)

def mul_terms(k):
    out = []
    for i in range(9):
        j = k - i
        if 0 <= j < 9:
            a, b = f"f{i}", f"g{j}"
            if needs2(i, j):
                if i % 3 == 1: a = f"f{i}_2"
                else: b = f"g{j}_2"
            out.append((a, b))
    return out

def sq_terms(k):
    """c f_i f_j with c = (1 or 2 for i != j) * (2 if needs2): 1 -> f_i f_j, 2 -> (2 f_i) f_j, 4 -> (2 f_i)(2 f_j)"""
    out = []
    for i in range(9):
        j = k - i
        if i <= j < 9:
            c = (1 if i == j else 2) * (2 if needs2(i, j) else 1)
            a = f"f{i}_2" if c >= 2 else f"f{i}"
            b = f"f{j}_2" if c == 4 else f"f{j}"
            out.append((a, b))
    return out

def sq_terms_dbl(k):
    """the columns of 2 f^2: c f_i f_j with c = 2 * (1 or 2 for i != j) * (2 if needs2): 2 -> (2 f_i) f_i, 4 -> (2 f_i)(2 f_j), 8 -> (4 f_x)(2 f_y) with
    x the one of i, j that is 1 mod 3 (it is whenever the radix doubles a squaring's cross term)"""
    out = []
    for i in range(9):
        j = k - i
        if i <= j < 9:
            c = 2 * (1 if i == j else 2) * (2 if needs2(i, j) else 1)
            if c == 2: out.append((f"f{i}_2", f"f{i}"))
            elif c == 4: out.append((f"f{i}_2", f"f{j}_2"))
            else:
                x, y = (i, j) if i % 3 == 1 else (j, i)
                assert x % 3 == 1
                out.append((f"f{x}_4", f"f{y}_2"))
    return out

helpers = {}
def blk(terms, first):
    """One column as ONE asm statement.  A term is (x, y): y a register name, or an int: -16..64 is an inline constant of the
    instruction, anything else sits in an SGPR (one constant-bus operand per VOP3 instruction on gfx9)."""
    kinds = tuple("v" if isinstance(y, str) else ("i%d" % y if -16 <= y <= 64 else "s") for x, y in terms)
    name = f"act_c{'z' if first else ''}{len(terms)}" + "".join("_" + k + "at%d" % t for t, k in enumerate(kinds) if k != "v")
    if name not in helpers:
        ops = '"=&v"(acc)' if first else '"+v"(acc)'
        ins, lines, params, hostsum, n_in = [], [], [], [], 1
        for t, k in enumerate(kinds):
            addend = "0" if first and t == 0 else "%0"
            params.append(f"uint32_t x{t}")
            ins.append(f'"v"(x{t})'); xa = f"%{n_in}"; n_in += 1
            if k == "v":
                params.append(f"uint32_t y{t}"); ins.append(f'"v"(y{t})'); ya = f"%{n_in}"; n_in += 1; hostsum.append(f"(uint64_t)x{t} * (uint64_t)y{t}")
            elif k == "s":
                params.append(f"uint32_t y{t}"); ins.append(f'"s"(y{t})'); ya = f"%{n_in}"; n_in += 1; hostsum.append(f"(uint64_t)x{t} * (uint64_t)y{t}")
            else:
                ya = k[1:]; hostsum.append(f"(uint64_t)x{t} * (uint64_t){k[1:]}u")
            lines.append(f"v_mad_u64_u32 %0, vcc, {xa}, {ya}, {addend}")
        dev = f'static __device__ __forceinline__ void {name}(uint64_t& acc, {", ".join(params)}) {{ asm("' + "\\n\\t".join(lines) + f'" : {ops} : {", ".join(ins)} : "vcc"); }}'
        host = f'static inline void {name}(uint64_t& acc, {", ".join(params)}) {{ acc = ' + ("" if first else "acc + ") + " + ".join(hostsum) + "; }"
        helpers[name] = (dev, host)
    args = ", ".join(x if not isinstance(y, str) and -16 <= y <= 64 else f"{x}, {y if isinstance(y, str) else str(y) + 'u'}" for x, y in terms)
    return name, args

def body(terms_of, prep):
    """High half first.  Its columns pass on only bits 32.. of their sums -- the upper register, times 2^(32 - width) as one
    more multiply-accumulate (an inline constant) -- so that the lower register IS the column's residue, no shift and no mask;
    low column k then starts from its carry-in and adds 19 * that residue and its own products."""
    m = [prep]
    for k in range(9, 17):
        t = terms_of(k)
        if k == 9: n, a = blk(t, True); m.append(f"  uint64_t h9; {n}(h9, {a});")
        else: n, a = blk([(f"(uint32_t)(h{k-1} >> 32)", 1 << (32 - W[k-10]))] + t, True); m.append(f"  uint64_t h{k}; {n}(h{k}, {a});")
    for k in range(9):                                       # low half: carry-in + 19 * residue + products
        t = [(f"(uint32_t)h{k+9}", 19)] + terms_of(k) if k < 8 else [(f"(uint32_t)(h16 >> 32)", 19 << (32 - W[7]))] + terms_of(k)
        if k == 0: n, a = blk(t, True); m.append(f"  uint64_t l0; {n}(l0, {a});")
        else: n, a = blk(t, False); m.append(f"  uint64_t l{k} = l{k-1} >> {W[k-1]}; {n}(l{k}, {a});")
    return " \\\n".join(m)

mul_prep = ("  const uint32_t " + ", ".join(f"f{i} = f.v[{i}]" for i in range(9)) + "; \\\n"
            "  const uint32_t " + ", ".join(f"g{i} = g.v[{i}]" for i in range(9)) + "; \\\n"
            "  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f{i}" for i in (1, 4, 7)) + ", " + ", ".join(f"g{i}_2 = 2u * g{i}" for i in (1, 4, 7)) + ";")
used = sorted({int(a[1]) for k in range(17) for a, b in sq_terms(k) if a.endswith("_2")} | {int(b[1]) for k in range(17) for a, b in sq_terms(k) if b.endswith("_2")})
sq_prep = ("  const uint32_t " + ", ".join(f"f{i} = f.v[{i}]" for i in range(9)) + "; \\\n"
           "  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f{i}" for i in used) + ";")
mb, sb = body(mul_terms, mul_prep), body(sq_terms, sq_prep)

# ---- the whole product as ONE asm statement (-DACT_FE_ONE_ASM; device only) -------------------------------------------------
# Same instructions as the per-column form above, scheduled here instead of by the compiler: no wait state is inserted between
# the pieces (the compiler puts an `s_nop 0` after every asm statement whose result the next instruction reads -- ~1 950 of them
# in k_spend_bits -- which the hardware does not need between VALU instructions) and no value is moved between statements.
# The sub-registers of a 64-bit operand cannot be named in an asm template, so the column accumulators live in FIXED registers
# listed as clobbers (TB .. TB+10; the k_spend_* kernels have a 256-register budget): two rotating pairs for the high columns, two
# for the low columns, the pair Q = (l0 residue, 19 * top carry's high bits), one scratch.
# Schedule: h9; then seven phases (h_{j+10} || l_j), two independent chains of 10 multiply-accumulates interleaved instruction by
# instruction; then l7, l8 and the wrap-around.
TB = 244
def pr(p): return f"v[{p[0]}:{p[1]}]"
def one_asm_lines(terms_of, TB=TB, tag="", addin=False):
    """the instruction list of one product: temporaries from register TB on, operands %[<name><tag>] (tag: the second product of a pair).
    addin: limb k of a tight element `a` joins low column k as one more multiply-accumulate (a_k * 1): the result is product + a, carried once"""
    HP = [(TB, TB + 1), (TB + 2, TB + 3)]
    LP = [(TB + 4, TB + 5), (TB + 6, TB + 7)]
    Q = (TB + 8, TB + 9)
    SC = TB + 10
    # the doubled limbs are operands, computed outside the statement: operands shared by several products (ge_from_completed)
    # then share their preparation, as they do in the per-column form
    def R(x): return f"%[{x}{tag}]"
    out = []
    def mads(dst, terms, first_addend):
        r, add = [], first_addend
        for x, y in terms:
            r.append(f"v_mad_u64_u32 {pr(dst)}, vcc, {x}, {y}, {add}"); add = pr(dst)
        return r
    def prod(k): return [(R(a), R(b)) for a, b in terms_of(k)] + ([(R(f"a{k}"), "1")] if addin and k < 9 else [])
    h = HP[0]
    out += mads(h, prod(9), "0")
    for j in range(7):                                        # phase: h_{j+10} || l_j
        hn, hp = HP[(j + 1) & 1], HP[j & 1]
        ln, lp = LP[j & 1], LP[(j + 1) & 1]
        hc = mads(hn, [(f"v{hp[1]}", str(1 << (32 - W[j])))] + prod(j + 10), "0")
        lc = []
        if j == 0:
            lc += mads(ln, [(f"v{hp[0]}", "19")] + prod(0), "0")
        else:
            lc.append(f"v_lshrrev_b64 {pr(ln)}, {W[j-1]}, {pr(lp)}")
            lc += mads(ln, [(f"v{hp[0]}", "19")] + prod(j), pr(ln))
            if j == 1: lc.append(f"v_and_b32 v{Q[0]}, 0x{(1 << W[0]) - 1:x}, v{lp[0]}")
            else: lc.append(f"v_and_b32 %[o{j-1}{tag}], 0x{(1 << W[j-1]) - 1:x}, v{lp[0]}")
        # interleave, high chain first; the longer tail runs on alone
        for i in range(max(len(hc), len(lc))):
            if i < len(hc): out.append(hc[i])
            if i < len(lc): out.append(lc[i])
    h16 = HP[1]                                               # h16 was written in phase j = 6 -> HP[(6 + 1) & 1]
    for j in (7, 8):
        ln, lp = LP[j & 1], LP[(j + 1) & 1]
        out.append(f"v_lshrrev_b64 {pr(ln)}, {W[j-1]}, {pr(lp)}")
        first = (f"v{h16[0]}", "19") if j == 7 else (f"v{h16[1]}", "%[c304]")
        lc = mads(ln, [first] + prod(j), pr(ln))
        lc.insert(2, f"v_and_b32 %[o{j-1}{tag}], 0x{(1 << W[j-1]) - 1:x}, v{lp[0]}")
        out += lc
    l8, t0 = LP[0], LP[1]                                     # l8 in LP[8 & 1]; the other pair is free for t0
    out += [f"v_lshrrev_b32 v{Q[1]}, 28, v{l8[1]}",
            f"v_alignbit_b32 v{SC}, v{l8[1]}, v{l8[0]}, 28",
            f"v_mul_u32_u24 v{Q[1]}, 19, v{Q[1]}",
            f"v_and_b32 %[o8{tag}], 0x{(1 << W[8]) - 1:x}, v{l8[0]}",
            f"v_mad_u64_u32 {pr(t0)}, vcc, v{SC}, 19, {pr(Q)}",
            f"v_alignbit_b32 v{SC}, v{t0[1]}, v{t0[0]}, 29",
            f"v_and_b32 %[o0{tag}], 0x{(1 << W[0]) - 1:x}, v{t0[0]}",
            f"v_add_u32 %[o1{tag}], %[o1{tag}], v{SC}"]
    return out

def asm_stmt(lines, outs, ins, temps):
    clob = ", ".join(f'"v{r}"' for r in temps) + ', "vcc"'
    text = " \\\n      ".join('"' + l + '\\n\\t"' for l in lines)
    return f"asm({text} \\\n      : {outs} \\\n      : {ins} \\\n      : {clob});"

def one_asm(terms_of, doubled, inputs):
    """doubled: names like 'f1_2' (the limb they double is the name without _2); inputs: operand names"""
    out = one_asm_lines(terms_of)
    outs = ", ".join(f'[o{i}] "=&v"(h.v[{i}])' for i in range(9))
    ins = ", ".join(f'[{n}] "v"({n[0]}.v[{n[1]}])' for n in inputs) + ", " + ", ".join(f'[{n}] "v"({n})' for n in doubled) + ', [c304] "s"(304u)'
    return asm_stmt(out, outs, ins, range(TB, TB + 11)), len(out)

# ---- TWO independent products in one statement, interleaved instruction by instruction (-DACT_FE_PAIR_ASM; measurement variant) ------
# ge_double squares X, Y, Z and X + Y; the d-free addition multiplies four independent pairs twice.  Two products side by side give
# every dependent multiply-accumulate -> shift -> multiply-accumulate step of one an independent instruction of the other to issue
# behind, which is what two wavefronts per SIMD cannot always provide (profiles/r03_ubench_dep.txt).  Second product: operands
# <name>b, results `hb`, temporaries TB2 .. TB2 + 10.
TB2 = 232
def pair_asm(terms_of, doubled, inputs, second):
    """second: (struct names of the second product's inputs) e.g. {'f': 'fb', 'g': 'gb'}"""
    a, b = one_asm_lines(terms_of, TB, ""), one_asm_lines(terms_of, TB2, "b")
    lines = [x for pair in zip(a, b) for x in pair]
    outs = ", ".join(f'[o{i}] "=&v"(h.v[{i}])' for i in range(9)) + ", " + ", ".join(f'[o{i}b] "=&v"(hb.v[{i}])' for i in range(9))
    ins = (", ".join(f'[{n}] "v"({n[0]}.v[{n[1]}])' for n in inputs) + ", " + ", ".join(f'[{n}] "v"({n})' for n in doubled) + ", " +
           ", ".join(f'[{n}b] "v"({second[n[0]]}.v[{n[1]}])' for n in inputs) + ", " + ", ".join(f'[{n}b] "v"({n}b)' for n in doubled) + ', [c304] "s"(304u)')
    return asm_stmt(lines, outs, ins, list(range(TB2, TB2 + 11)) + list(range(TB, TB + 11)))

def pair_asm_mixed(terms_a, ops_a, addin_a, terms_b, ops_b):
    """first product: operands ops_a = [(asm name, C expression)], result `h`; second: ops_b (asm names get the tag b), result `hb`"""
    a, b = one_asm_lines(terms_a, TB, "", addin_a), one_asm_lines(terms_b, TB2, "b")
    lines = [x for pair in zip(a, b) for x in pair] + a[len(b):] + b[len(a):]
    outs = ", ".join(f'[o{i}] "=&v"(h.v[{i}])' for i in range(9)) + ", " + ", ".join(f'[o{i}b] "=&v"(hb.v[{i}])' for i in range(9))
    ins = ", ".join(f'[{n}] "v"({e})' for n, e in ops_a) + ", " + ", ".join(f'[{n}b] "v"({e})' for n, e in ops_b) + ', [c304] "s"(304u)'
    return asm_stmt(lines, outs, ins, list(range(TB2, TB2 + 11)) + list(range(TB, TB + 11)))

mul_one, n_mul = one_asm(mul_terms, [f"f{i}_2" for i in (1, 4, 7)] + [f"g{i}_2" for i in (1, 4, 7)], [f"f{i}" for i in range(9)] + [f"g{i}" for i in range(9)])
sq_one, n_sq = one_asm(sq_terms, [f"f{i}_2" for i in used], [f"f{i}" for i in range(9)])

print("// GENERATED by tools/gen_fe_mul.py -- do not edit")
print("#if defined(__HIP_DEVICE_COMPILE__)")
for n, (d, h) in helpers.items(): print(d)
print("#else")
for n, (d, h) in helpers.items(): print(h)
print("#endif")
print("#define ACT_FE_MUL_BODY \\\n" + mb + "\n")
print("#define ACT_FE_SQ_BODY \\\n" + sb)
print()
print("// one statement per product (fe25519.h: ACT_FE_ONE_ASM); result in `fe h`, operands `f`, `g`")
print("#define ACT_FE_MUL_ONE_ASM \\\n  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f.v[{i}]" for i in (1, 4, 7)) + ", " + ", ".join(f"g{i}_2 = 2u * g.v[{i}]" for i in (1, 4, 7)) + "; \\\n  " + mul_one)
print()
print("#define ACT_FE_SQ_ONE_ASM \\\n  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f.v[{i}]" for i in used) + "; \\\n  " + sq_one)
print()
print("// two products per statement (fe25519.h: ACT_FE_PAIR_ASM): h = f * g, hb = fb * gb;  h = f^2, hb = fb^2")
mul_d = [f"f{i}_2" for i in (1, 4, 7)] + [f"g{i}_2" for i in (1, 4, 7)]
print("#define ACT_FE_MUL2_ONE_ASM \\\n  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f.v[{i}]" for i in (1, 4, 7)) + ", " + ", ".join(f"g{i}_2 = 2u * g.v[{i}]" for i in (1, 4, 7)) + ", " +
      ", ".join(f"f{i}_2b = 2u * fb.v[{i}]" for i in (1, 4, 7)) + ", " + ", ".join(f"g{i}_2b = 2u * gb.v[{i}]" for i in (1, 4, 7)) + "; \\\n  " +
      pair_asm(mul_terms, mul_d, [f"f{i}" for i in range(9)] + [f"g{i}" for i in range(9)], {"f": "fb", "g": "gb"}))
print()
print("#define ACT_FE_SQ2_ONE_ASM \\\n  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f.v[{i}]" for i in used) + ", " + ", ".join(f"f{i}_2b = 2u * fb.v[{i}]" for i in used) + "; \\\n  " +
      pair_asm(sq_terms, [f"f{i}_2" for i in used], [f"f{i}" for i in range(9)], {"f": "fb"}))
print()
print("// h = 2 f^2 + a (one carry), hb = fb^2 in one statement (fe25519.h fe_sqda_sq: the doubling's 2 Z^2 + X^2 beside (X + Y)^2)")
ops_a = ([(f"f{i}", f"f.v[{i}]") for i in range(9)] + [(f"f{i}_2", f"f{i}_2") for i in range(9)] + [(f"f{i}_4", f"f{i}_4") for i in (1, 4, 7)] +
         [(f"a{i}", f"a.v[{i}]") for i in range(9)])
ops_b = [(f"f{i}", f"fb.v[{i}]") for i in range(9)] + [(f"f{i}_2", f"f{i}_2b") for i in used]
print("#define ACT_FE_SQDA_SQ_ONE_ASM \\\n  const uint32_t " + ", ".join(f"f{i}_2 = 2u * f.v[{i}]" for i in range(9)) + ", " + ", ".join(f"f{i}_4 = 4u * f.v[{i}]" for i in (1, 4, 7)) + ", " +
      ", ".join(f"f{i}_2b = 2u * fb.v[{i}]" for i in used) + "; \\\n  " + pair_asm_mixed(sq_terms_dbl, ops_a, True, sq_terms, ops_b))
import sys
print(f"// one-statement forms: {n_mul} / {n_sq} instructions", file=sys.stderr)
print(f"// multiply-accumulates: mul {sum(len(mul_terms(k)) for k in range(17)) + 16}, sq {sum(len(sq_terms(k)) for k in range(17)) + 16}; widths {W}", file=sys.stderr)
