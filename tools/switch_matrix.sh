#!/bin/bash
# Every QARIG_* kernel-selection switch against the parity tests (run ON THE GPU BOX): the
# switches only choose between kernels that must give the same results.
cd "$GRAFT_REPO_ROOT"
for e in QARIG_GEMM_PF=0 QARIG_GEMM_DMA=0 QARIG_GEMM_PAIR=0 QARIG_GEMM_PAIR=1 QARIG_BMU_RESIDENT=0 QARIG_BMU_CS=1 \
         QARIG_BMU_GROUPS=1 QARIG_ATTN_QW=1 QARIG_ATTN_BW=1; do
    echo "== $e"
    env $e timeout -k 10 300 python -m pytest tests/test_gpu_core.py tests/test_gpu_transformer.py tests/test_gpu_codebook.py -x -q 2>&1 | tail -1
done
for e in QARIG_LP_BIG=0 QARIG_LP_MFMA16=0 "QARIG_LP_BIG=1 QARIG_LP_MFMA16=0" "QARIG_LP_BIG=1"; do
    echo "== $e"
    env $e timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fp8.py -x -q 2>&1 | tail -1
done
for e in QARIG_CONVT_PAIR=0 QARIG_CONVT_PAIR=1; do
    echo "== $e"
    env $e timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -1
done
for e in QARIG_CONV_RING=0; do
    echo "== $e"
    env $e timeout -k 10 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_cli.py -x -q 2>&1 | tail -1
done
