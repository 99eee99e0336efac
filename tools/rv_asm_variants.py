"""Variants of the SLP-vectorised device assembly of conv_bf16_rv.hip for the bisection of the packed-fp32 miscompare (tools/rv_asm_build.py).
Only the packed multiply-adds of the AFFINE COMMIT path are touched -- the two forms the commit compiles to:
   hi : v_pk_fma_f32 v[D:D+1], v[A:A+1], v[S:S+1], v[H:H+1] op_sel_hi:[1,0,0]     (scale = low half of S, shift = low half of H, both lanes)
   sel: v_pk_fma_f32 v[D:D+1], v[A:A+1], v[S:S+1], v[H:H+1] op_sel:[0,1,1]        (scale / shift = the HIGH halves)
usage: rv_asm_variants.py <slp.s> <outdir>   -> writes <outdir>/<variant>.s, prints the variant names"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
out = sys.argv[2]
PAT = re.compile(r"^\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] (op_sel_hi:\[1,0,0\]|op_sel:\[0,1,1\])\s*$")


def scalar(m):
    d, _, a, _, s, _, h, _, form = m.groups()
    d, a, s, h = int(d), int(a), int(s), int(h)
    if form.startswith("op_sel_hi"):
        lo = f"\tv_fma_f32 v{d}, v{a}, v{s}, v{h}"
        hi = f"\tv_fma_f32 v{d + 1}, v{a + 1}, v{s}, v{h}"
        # a destination that is also the scale / shift register of the OTHER lane's instruction goes last
        return [hi, lo] if d in (s, h) else [lo, hi]
    lo = f"\tv_fma_f32 v{d}, v{a}, v{s + 1}, v{h + 1}"
    hi = f"\tv_fma_f32 v{d + 1}, v{a + 1}, v{s + 1}, v{h + 1}"
    return [lo, hi] if d + 1 in (s + 1, h + 1) else [hi, lo] if d in (s + 1, h + 1) else [lo, hi]


def variant(name, fn):
    res, n = [], 0
    for i, line in enumerate(src):
        m = PAT.match(line)
        if m:
            new = fn(m, line, i)
            n += new != [line]
            res += new
        else:
            res.append(line)
    open(f"{out}/{name}.s", "w").write("\n".join(res))
    print(name, n)


def prev_writes_high_of_S(i, m):
    """the instruction(s) right before write the HIGH register of the scale pair (the 'dead half' the compiler re-uses as a temporary)"""
    s = int(m.group(5))
    for j in range(i - 1, max(i - 4, 0), -1):
        if re.match(rf"^\tv_cvt_pk_bf16_f32 v{s + 1},", src[j]):
            return True
    return False


variant("B_all_scalar", lambda m, l, i: scalar(m))
variant("B1_hi_scalar", lambda m, l, i: scalar(m) if m.group(9).startswith("op_sel_hi") else [l])
variant("B2_sel_scalar", lambda m, l, i: scalar(m) if m.group(9).startswith("op_sel:") else [l])
variant("B3_dst_is_src_scalar", lambda m, l, i: scalar(m) if m.group(1) == m.group(5) else [l])
variant("B4_after_deadhalf_write_scalar", lambda m, l, i: scalar(m) if prev_writes_high_of_S(i, m) else [l])
variant("B5_all_but_deadhalf_scalar", lambda m, l, i: [l] if (prev_writes_high_of_S(i, m) or m.group(1) == m.group(5)) else scalar(m))
variant("C_nop_before", lambda m, l, i: ["\ts_nop 7", l])
variant("D_nop_after", lambda m, l, i: [l, "\ts_nop 7"])


# ---- second round: only the `op_sel:[0,1,1]` form (the first round: scalarising these 104 instructions alone makes the kernel exact)
def good_form_after_moves(m, l):
    """the same arithmetic in the form the first round found innocent: the wanted (high) halves copied into the low registers of the SAME
    aligned pairs first (gfx950 wants 64-bit aligned tuples, so the pairs cannot be re-based), then `op_sel_hi:[1,0,0]`.  The low halves
    (channel 0's scale / shift) are dead by then: the first commit serves channel 0 before channel 1 and the loop re-reads its scales."""
    d, d1, a, a1, s, s1, h, h1, _ = m.groups()
    return [f"\tv_mov_b32_e32 v{s}, v{s1}", f"\tv_mov_b32_e32 v{h}, v{h1}",
            f"\tv_pk_fma_f32 v[{d}:{d1}], v[{a}:{a1}], v[{s}:{s1}], v[{h}:{h1}] op_sel_hi:[1,0,0]"]


SEL = lambda m: m.group(9).startswith("op_sel:")   # noqa: E731
variant("G6_sel_as_good_form_after_moves", lambda m, l, i: good_form_after_moves(m, l) if SEL(m) else [l])
variant("G5_sel_nop_both_sides", lambda m, l, i: ["\ts_nop 7", l, "\ts_nop 7"] if SEL(m) else [l])


# ---- third round: which of the `op_sel:[0,1,1]` instructions?  They come in runs of four per commit slot (two back-to-back pairs); a kernel's
# first commit has two such slots per site: the first under the full exec mask, the second under a partial one (lanes 0 .. 35).
_sel_idx = {}


def nth_sel(i):
    """running index of the op_sel:[0,1,1] instruction at source line i (over the whole file)"""
    if not _sel_idx:
        n = 0
        for j, line in enumerate(src):
            m = PAT.match(line)
            if m and SEL(m):
                _sel_idx[j] = n
                n += 1
    return _sel_idx[i]


variant("H1_first_slot_scalar", lambda m, l, i: scalar(m) if SEL(m) and nth_sel(i) % 8 < 4 else [l])
variant("H2_second_slot_scalar", lambda m, l, i: scalar(m) if SEL(m) and nth_sel(i) % 8 >= 4 else [l])
variant("H3_first_of_pair_scalar", lambda m, l, i: scalar(m) if SEL(m) and nth_sel(i) % 2 == 0 else [l])
variant("H4_second_of_pair_scalar", lambda m, l, i: scalar(m) if SEL(m) and nth_sel(i) % 2 == 1 else [l])


# ---- fourth round: the first slot's instructions kept, a LONG quiet period in front of the slot (everything in flight drained)
def drained(m, l, i):
    if SEL(m) and nth_sel(i) % 8 == 0:
        return ["\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"] + ["\ts_nop 7"] * 16 + [l]
    return [l]


variant("I1_first_slot_behind_drain", drained)
