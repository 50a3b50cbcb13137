#!/usr/bin/env python3
"""Writes ../mlp3_kstep.inc: the k-step of the split-bf16 MLP kernel (mlp_bf16x3.hip) as ONE inline-assembly block per
shape, so that its schedule is exactly the one written here: the A operands (three bf16 weight pieces per 32-feature output
tile, from the LDS chunk at `ab`) of tile pair P + 1 are requested before the 12 MFMAs of pair P issue, and each wait is a
hand-placed `s_waitcnt lgkmcnt` (LDS returns in order).  The compiler cannot see LDS reads that overlap the `buffer_load ...
lds` stream without ordering them behind it, and asynchronous reads in separate asm statements are unsafe (it may copy a
destination register before the wait), hence one block.  The `_split` shapes also split the NEXT k-step's 8 activations into their bf16 pieces inside the block
(behind the first LDS requests).

Per tile the six significant products of (w_hi + w_mid + w_lo)(x_hi + x_mid + x_lo) are accumulated smallest first:
    mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi.
Run:  python3 gen/mlp3_kstep.py   (from fv3net_amd/csrc; the output is committed)."""
import os

EARLY_SPLIT = 0  # VALU instructions of the split issued before the first wait (measured: 22 makes the launch 2 % slower -- there is
                 # no LDS bubble to fill there; so does moving the chunk request in front of the first wait: 6 % slower)
PRODUCTS = [(1, "bm"), (0, "bl"), (2, "bh"), (0, "bm"), (1, "bh"), (0, "bh")]  # (weight piece, activation piece)


def split_lines():
    """x = hi + mid + lo in bf16 (round to nearest; the residuals are exact in fp32) for the 8 values of the NEXT k-step:
    11 VALU instructions per pair of values."""
    out = []
    for i in range(4):
        a, b = f"%[xn{2 * i}]", f"%[xn{2 * i + 1}]"
        out += [
            f"v_cvt_pk_bf16_f32 %[nh{i}], {a}, {b}",
            f"v_lshlrev_b32 %[u0], 16, %[nh{i}]",
            f"v_and_b32 %[u1], 0xffff0000, %[nh{i}]",
            f"v_sub_f32 %[r0], {a}, %[u0]",
            f"v_sub_f32 %[r1], {b}, %[u1]",
            f"v_cvt_pk_bf16_f32 %[nm{i}], %[r0], %[r1]",
            f"v_lshlrev_b32 %[u0], 16, %[nm{i}]",
            f"v_and_b32 %[u1], 0xffff0000, %[nm{i}]",
            "v_sub_f32 %[r0], %[r0], %[u0]",
            "v_sub_f32 %[r1], %[r1], %[u1]",
            f"v_cvt_pk_bf16_f32 %[nl{i}], %[r0], %[r1]",
        ]
    return out


def dma_lines(per):
    """`per` buffer_load_dwordx4 ... lds of this wave (1 KB each, 4 KB apart in the chunk and in LDS); M0 = LDS address.
    (An instruction sits between every M0 write and the load that uses it.)"""
    out = ["s_mov_b32 m0, %[dl]"]
    for i in range(per):
        out.append("s_nop 0" if i == 0 else "s_add_u32 %[dg], %[dg], 0x1000")
        out.append("buffer_load_dwordx4 %[dv], %[dr], %[dg] offen lds")
        if i + 1 < per:
            out.append("s_add_u32 m0, m0, 0x1000")
    return out


def block(nt, first, tab, split, per):
    """The instruction list of one k-step.  Measured on gfx950 (benchmarks/mfma_ubench): beside a pair of 32x32x16 bf16 MFMAs (64
    cycles of matrix pipe) about four VALU instructions issue for free, so the side work -- the chunk request (`per` loads
    when > 0) and the activation split -- is spread over the MFMA stream, at most two instructions behind each MFMA."""
    lines = []
    emit = lines.append
    pairs = [(2 * p, 2 * p + 1 if 2 * p + 1 < nt else None) for p in range((nt + 1) // 2)]

    def request(p):
        t0, t1 = pairs[p]
        s = p & 1
        n = 0
        for q in range(3):
            emit(f"ds_read_b128 %[t{s}0{q}], %[ab] offset:{(q * nt + t0) * 1024}")
            n += 1
            if t1 is not None:
                emit(f"ds_read_b128 %[t{s}1{q}], %[ab] offset:{(q * nt + t1) * 1024}")
                n += 1
        return n

    side = (dma_lines(per) if per else []) + (split_lines() if split else [])
    if tab:  # the centre / epsilon rows of a later layer-1 k-step ride along (they land before the first A operands)
        for i in range(4):
            emit(f"ds_read_b128 %[x{i}], %[tb] offset:{16 * i}")
    request(0)
    pending1 = request(1) if len(pairs) > 1 else 0
    if split:
        for _ in range(EARLY_SPLIT):
            emit(side.pop(len(dma_lines(per)) if per else 0))
    for p, (t0, t1) in enumerate(pairs):
        s = p & 1
        if p == 0:
            pending = pending1
        else:
            pending = request(p + 1) if p + 1 < len(pairs) else 0
        emit(f"s_waitcnt lgkmcnt({pending})")
        emit("s_nop 0")
        for i, (q, b) in enumerate(PRODUCTS):
            for tt, t in ((0, t0), (1, t1)):
                if t is None:
                    continue
                c = "0" if (first and i == 0) else f"%[c{t}]"
                emit(f"v_mfma_f32_32x32x16_bf16 %[c{t}], %[t{s}{tt}{q}], %[{b}], {c}")
                for _ in range(2):
                    if side:
                        emit(side.pop(0))
    for x in side:
        emit(x)
    return "\\n\\t".join(lines)


def function(nt, tab, split, per):
    name = f"kstep_asm_{nt}" + ("_tab" if tab else "") + ("_split" if split else "") + (f"_dma{per}" if per else "")
    args = f"f32x16 (&c)[{nt}], uint32_t ab, const B3 &b"
    if tab:
        args += ", uint32_t tb, f32x4 &x0, f32x4 &x1, f32x4 &x2, f32x4 &x3"
    if split:
        args += ", const float (&xn)[8], B3 &bn"
    if per:
        args += ", i32x4 dr, uint32_t dg, uint32_t dl, uint32_t dv"
    temps = [f"t{s}{tt}{q}" for s in range(2) for tt in range(2) for q in range(3)]
    out = [f"template <bool FIRST>\n__device__ __forceinline__ void {name}({args})\n{{"]
    out.append("    f32x4 " + ", ".join(temps) + ";")
    if split:
        out.append("    uint32_t " + ", ".join(f"n{w}{i}" for w in "hml" for i in range(4)) + ";")
        out.append("    float u0, u1, r0, r1;")
    for first in (True, False):
        out.append("    if constexpr (%sFIRST) {" % ("" if first else "!"))
        cons = "=&a" if first else "+a"
        outs = [f'[c{t}] "{cons}"(c[{t}])' for t in range(nt)] + [f'[{t}] "=&v"({t})' for t in temps]
        if tab:
            outs += [f'[x{i}] "=&v"(x{i})' for i in range(4)]
        ins = ['[ab] "v"(ab)', '[bh] "v"(b.hi)', '[bm] "v"(b.mid)', '[bl] "v"(b.lo)'] + (['[tb] "v"(tb)'] if tab else [])
        if split:
            outs += [f'[n{w}{i}] "=&v"(n{w}{i})' for w in "hml" for i in range(4)]
            outs += [f'[{t}] "=&v"({t})' for t in ("u0", "u1", "r0", "r1")]
            ins += [f'[xn{i}] "v"(xn[{i}])' for i in range(8)]
        clob = '"memory"'
        if per:
            outs += ['[dg] "+s"(dg)']
            ins += ['[dr] "s"(dr)', '[dl] "s"(dl)', '[dv] "v"(dv)']
            clob += ', "m0", "scc"'
        out.append(f'        asm volatile("{block(nt, first, tab, split, per)}"')
        out.append("                     : " + ", ".join(outs))
        out.append("                     : " + ", ".join(ins))
        out.append(f"                     : {clob});")
        out.append("    }")
    if split:
        for w, field in (("h", "hi"), ("m", "mid"), ("l", "lo")):
            out.append(f"    bn.{field} = __builtin_bit_cast(bf16x8, u32x4{{n{w}0, n{w}1, n{w}2, n{w}3}});")
    out.append("}\n")
    return "\n".join(out)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    parts = ["// GENERATED by gen/mlp3_kstep.py -- do not edit; see the generator for the schedule.\n"]
    # (tiles, with the layer-1 table row, chunk loads inside: 6 = a hidden-type chunk, 10 / 4 / 3 / 1 = the output-type chunk of
    # 13 / 5 / 3 / 1 tiles, 0 = requested by the caller)
    per_o = {13: 10, 5: 4, 3: 3, 1: 1}
    shapes = [(8, False, 0), (8, False, 6), (8, True, 0), (8, True, 6)]
    for nt in (13, 5, 3, 1):
        shapes += [(nt, False, 0), (nt, False, per_o[nt])]
    for nt, tab, per in shapes:
        for split in (False, True):
            parts.append(function(nt, tab, split, per))
    with open(os.path.join(here, "..", "mlp3_kstep.inc"), "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
