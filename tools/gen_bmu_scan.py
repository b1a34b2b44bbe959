#!/usr/bin/env python3
"""Writes csrc/bmu_scan_asm.inc: the coarse BMU scan loop of csrc/bmu.hip (bmu_coarse_kernel) as one inline-asm
string.  hipcc's scheduler would not keep the step's shape (six chained MFMAs of tile T+1, each with seven scan
instructions of tile T behind it, the LDS reads of tile T+2 ahead of them), whatever sched_group_barrier pipeline
it was given, so the order is written down here:

  registers (clobbered by the statement): accumulators A / B / C = v[128:143] / v[144:159] / v[160:175] in rotation
  (scanned / chained / being loaded), fragment sets v[176:187] / v[188:199] / v[200:211], temporaries v212-v213,
  tile counter v214, running minimum v215-v217 (rotating, so `minimum fell in this tile` is a compare of two
  registers), second-smallest v218, tile of the minimum v219, LDS addresses v220-v223, -inf v224.

Wait states (tools/isa_lint.py R1 / R2): no vector-ALU result feeds an MFMA; an accumulator is read by the scan
at least 16 vector instructions + 8 others after the last MFMA of its chain (12 needed for the 8-pass bf16 MFMA);
every LDS read is waited for (lgkmcnt(0)) one whole step after it was issued."""
import os

ACC = {"A": 128, "B": 144, "C": 160}
FRG = {"A": 176, "B": 188, "C": 200}
T0, T1, TC, BEST, SEC, TIDX, F0, F1, F2, W, NINF = 212, 213, 214, (215, 216, 217), 218, 219, 220, 221, 222, 223, 224


def rng(b, n):
    return f"v[{b}:{b + n - 1}]"


def mfma(acc, frag, x):
    return f"v_mfma_f32_32x32x16_bf16 {rng(acc, 16)}, {rng(frag, 4)}, {x}, {rng(acc, 16)}"


def chain(acc, f):
    """the six products of a tile, smallest magnitudes first: lo.xh, hi.xl, mid.xm, mid.xh, hi.xm, hi.xh"""
    return [mfma(acc, f + 8, "%[xh]"), mfma(acc, f, "%[xl]"), mfma(acc, f + 4, "%[xm]"),
            mfma(acc, f + 4, "%[xh]"), mfma(acc, f, "%[xm]"), mfma(acc, f, "%[xh]")]


def loads(fn, nn, tile):
    o, ow = tile * 1024, tile * 128
    return [f"ds_read_b128 {rng(fn, 4)}, v{F0} offset:{o}", f"ds_read_b128 {rng(fn + 4, 4)}, v{F1} offset:{o}",
            f"ds_read_b128 {rng(fn + 8, 4)}, v{F2} offset:{o}"] + \
           [f"ds_read_b128 {rng(nn + 4 * q, 4)}, v{W} offset:{ow + 16 * q}" for q in range(4)]


def step(cur, nxt, nn, bi, bo, tile):
    """scan accumulator `cur`, chain `nxt` with the fragments that share its letter, load tile `tile` (relative to
    the base addresses) into set `nn`; the minimum moves from BEST[bi] to BEST[bo]"""
    c = ACC[cur]
    valu = []
    tmp = (T0, T1)

    def A(r):
        return f"v_and_or_b32 v{tmp[r & 1]}, v{c + r}, -16, {r}"

    def S(r):
        return f"v_med3_f32 v{SEC}, v{BEST[bi] if r == 0 else BEST[bo]}, v{tmp[r & 1]}, v{SEC}"

    def B(r):
        return f"v_med3_f32 v{BEST[bo]}, v{BEST[bi] if r == 0 else BEST[bo]}, v{tmp[r & 1]}, v{NINF}"

    valu += [A(0), A(1)]
    for r in range(16):
        valu += [S(r), B(r)]
        if r + 2 < 16:
            valu.append(A(r + 2))
    valu += [f"v_cmp_lt_f32_e32 vcc, v{BEST[bo]}, v{BEST[bi]}", f"v_cndmask_b32_e32 v{TIDX}, v{TIDX}, v{TC}, vcc",
             f"v_add_u32_e32 v{TC}, 1, v{TC}"]
    assert len(valu) == 51
    out = ["s_waitcnt lgkmcnt(0)"] + loads(FRG[nn], ACC[nn], tile)
    ms = chain(ACC[nxt], FRG[nxt])
    at = 0
    for i, m in enumerate(ms):
        out.append(m)
        n = 7 if i < 5 else len(valu) - at
        out += valu[at:at + n]
        at += n
    return out


def main():
    L = []
    L += [f"v_mov_b32_e32 v{F0}, %[f0]", f"v_add_u32_e32 v{F1}, %[pl], v{F0}", f"v_add_u32_e32 v{F2}, %[pl], v{F1}",
          f"v_mov_b32_e32 v{W}, %[w]", f"v_mov_b32_e32 v{TC}, 0", f"v_mov_b32_e32 v{BEST[0]}, 0x7f800000",
          f"v_mov_b32_e32 v{SEC}, 0x7f800000", f"v_mov_b32_e32 v{TIDX}, -1", f"v_mov_b32_e32 v{NINF}, 0xff800000"]
    L += loads(FRG["A"], ACC["A"], 0) + loads(FRG["B"], ACC["B"], 1)
    L += ["s_waitcnt lgkmcnt(0)"] + chain(ACC["A"], FRG["A"]) + ["s_nop 7", "s_mov_b32 %[t], 0"]
    L += ["1:", "s_add_i32 %[u], %[t], 3", "s_cmp_gt_i32 %[u], %[nt]", "s_cbranch_scc1 2f"]
    L += step("A", "B", "C", 0, 1, 2) + step("B", "C", "A", 1, 2, 3) + step("C", "A", "B", 2, 0, 4)
    L += [f"v_add_u32_e32 v{F0}, 0xc00, v{F0}", f"v_add_u32_e32 v{F1}, 0xc00, v{F1}", f"v_add_u32_e32 v{F2}, 0xc00, v{F2}",
          f"v_add_u32_e32 v{W}, 0x180, v{W}", "s_add_i32 %[t], %[t], 3", "s_branch 1b"]
    L += ["2:", "s_cmp_ge_i32 %[t], %[nt]", "s_cbranch_scc1 5f"]
    L += step("A", "B", "C", 0, 1, 2)
    L += ["s_add_i32 %[u], %[t], 1", "s_cmp_ge_i32 %[u], %[nt]", "s_cbranch_scc1 4f"]
    L += step("B", "C", "A", 1, 2, 3)
    L += [f"v_mov_b32_e32 %[best], v{BEST[2]}", "s_branch 6f",
          "4:", f"v_mov_b32_e32 %[best], v{BEST[1]}", "s_branch 6f",
          "5:", f"v_mov_b32_e32 %[best], v{BEST[0]}",
          "6:", f"v_mov_b32_e32 %[sec], v{SEC}", f"v_mov_b32_e32 %[tidx], v{TIDX}"]
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "quantized-autoregression-image-generator_amd", "csrc", "bmu_scan_asm.inc")
    with open(dst, "w") as f:
        f.write("// generated by tools/gen_bmu_scan.py -- do not edit\n")
        for line in L:
            f.write(f'"{line}\\n\\t"\n')
    print(f"{len(L)} lines -> {os.path.normpath(dst)}")


if __name__ == "__main__":
    main()
