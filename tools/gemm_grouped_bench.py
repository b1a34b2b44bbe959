#!/usr/bin/env python3
"""Grouped against separate launches on the Linear products of an 8-sequence shard (run on the GPU
box: `python tools/gemm_grouped_bench.py`; ROWS=2048 by default).  HIP events, interleaved rounds in
one process; the separate form is what ops.gemm does on its own (its own split choice, its reduce
launches included), the grouped form one qarig_gemm_f32_grouped call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops  # noqa: E402

M = int(os.environ.get("ROWS", "2048"))
D, H = 512, 2048


def timeit(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(s, device=dev, generator=g)   # noqa: E731
    for G in (int(v) for v in os.environ.get("GROUPS", "3,2,14").split(",")):
        x = r(M, D)
        w1, b1, w2, b2 = [r(H, D) for _ in range(G)], [r(H) for _ in range(G)], [r(D, H) for _ in range(G)], [r(D) for _ in range(G)]
        hs, t1s = r(G, M, H), r(G, M, H)
        ys = torch.empty((G, M, D), device=dev)
        dT2 = [r(M, D) for _ in range(G)]
        dT1s = r(G, M, H)
        dx = torch.empty((M, D), device=dev)
        gw1 = [torch.zeros((H, D), device=dev) for _ in range(G)]
        gb1 = [torch.zeros(H, device=dev) for _ in range(G)]
        gw2 = [torch.zeros((D, H), device=dev) for _ in range(G)]
        gb2 = [torch.zeros(D, device=dev) for _ in range(G)]
        h, t1, dT1 = list(hs.unbind(0)), list(t1s.unbind(0)), list(dT1s.unbind(0))
        gs = ops.grouped_splitk

        def sep_f1():
            for i in range(G):
                ops.gemm(x, w1[i], bias=b1[i], want_preact=True, act=1)

        def grp_f1():
            ops.gemm_grouped([x] * G, w1, h, M, H, D, bias=b1, preact=t1, act=1, splitk=gs(G, M, H, D))

        def sep_f2():
            for i in range(G):
                ops.gemm(h[i], w2[i], bias=b2[i])

        def grp_f2():
            ops.gemm_grouped(h, w2, list(ys.unbind(0)), M, D, H, bias=b2, splitk=gs(G, M, D, H))

        def sep_b1():
            for i in range(G):
                ops.gemm(dT2[i], w2[i], a_kcontig=True, b_kcontig=False, gradz=t1[i], gact=1)

        def grp_b1():
            ops.gemm_grouped(dT2, w2, dT1, M, H, D, True, False, gradz=t1, gact=1, splitk=gs(G, M, H, D))

        def sep_b2():
            ops.gemm(dT1[0], w1[0], a_kcontig=True, b_kcontig=False, out=dx)
            for i in range(1, G):
                ops.gemm(dT1[i], w1[i], a_kcontig=True, b_kcontig=False, out=dx, accumulate=True, splitk=1)

        def grp_b2():
            ops.gemm_grouped(dT1, w1, [dx], M, D, H, True, False, sum_groups=True, splitk=gs(G, M, D, H))

        def sep_w2():
            for i in range(G):
                ops.gemm(dT2[i], h[i], a_kcontig=False, b_kcontig=False, splitk=ops.pick_splitk(D, H, M),
                         out=gw2[i], accumulate=True, a_rowsum=gb2[i])

        def grp_w2():
            ops.gemm_grouped(dT2, h, gw2, D, H, M, False, False, accumulate=True, a_rowsum=gb2, splitk=gs(G, D, H, M))

        def sep_w1():
            for i in range(G):
                ops.gemm(dT1[i], x, a_kcontig=False, b_kcontig=False, splitk=ops.pick_splitk(H, D, M),
                         out=gw1[i], accumulate=True, a_rowsum=gb1[i])

        def grp_w1():
            ops.gemm_grouped(dT1, [x] * G, gw1, H, D, M, False, False, accumulate=True, a_rowsum=gb1,
                             splitk=gs(G, H, D, M))

        cases = [("F1  x W1^T +b,silu,preact", sep_f1, grp_f1, (M, H, D)), ("F2  h W2^T +b", sep_f2, grp_f2, (M, D, H)),
                 ("B1  dT2 W2 * act'", sep_b1, grp_b1, (M, H, D)), ("B2  sum dT1 W1", sep_b2, grp_b2, (M, D, H)),
                 ("W2g dT2^T h +rowsum", sep_w2, grp_w2, (D, H, M)), ("W1g dT1^T x +rowsum", sep_w1, grp_w1, (H, D, M))]
        tot_s = tot_g = 0.0
        print(f"--- G = {G}, rows = {M}")
        for name, sep, grp, (m, n, k) in cases:
            ts, tg = [], []
            for _ in range(5):
                ts.append(timeit(sep))
                tg.append(timeit(grp))
            s, q = sorted(ts)[2], sorted(tg)[2]
            fl = 2.0 * G * m * n * k
            tot_s += s
            tot_g += q
            print(f"{name:28s} separate {s:8.1f} us ({fl / s / 1e6:6.1f} TF)   grouped {q:8.1f} us ({fl / q / 1e6:6.1f} TF)"
                  f"   splitk {gs(G, m, n, k)}")
        print(f"{'all six products':28s} separate {tot_s:8.1f} us   grouped {tot_g:8.1f} us")


if __name__ == "__main__":
    main()
