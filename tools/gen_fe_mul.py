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
print("// GENERATED by tools/gen_fe_mul.py -- do not edit")
print("#if defined(__HIP_DEVICE_COMPILE__)")
for n, (d, h) in helpers.items(): print(d)
print("#else")
for n, (d, h) in helpers.items(): print(h)
print("#endif")
print("#define ACT_FE_MUL_BODY \\\n" + mb + "\n")
print("#define ACT_FE_SQ_BODY \\\n" + sb)
import sys
print(f"// multiply-accumulates: mul {sum(len(mul_terms(k)) for k in range(17)) + 16}, sq {sum(len(sq_terms(k)) for k in range(17)) + 16}; widths {W}", file=sys.stderr)
