"""ctypes binding of libqarig_hip.so (C ABI declared in include/qarig.h).

There is no CPU fallback: if the library is missing, or a CPU tensor reaches an op,
the call raises.  torch is used only to own device memory and streams.
"""
import ctypes
import os
from ctypes import c_int, c_int64, c_size_t, c_void_p, c_float, c_char_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib",
                        os.environ.get("QARIG_LIB", "libqarig_hip.so"))   # QARIG_LIB: ablation builds

P = c_void_p
I = c_int
L = c_int64
Z = c_size_t
F = c_float

# name -> (restype, argtypes); must list every symbol include/qarig.h declares
# (tests/test_abi.py cross-checks the two).
SIGNATURES = {
    "qarig_version": (I, []),
    "qarig_target_arch": (c_char_p, []),
    "qarig_last_error": (I, [c_char_p, Z]),
    "qarig_set_option": (I, [c_char_p, I]),
    "qarig_bmu_workspace_bytes": (Z, [L, I]),
    "qarig_bmu_fwd": (I, [P, I, I, I, I, I, I, P, I, I, P, P, Z, P]),
    "qarig_bmu_fwd_prepared": (I, [P, I, I, I, I, I, I, P, I, I, P, P, Z, P, P]),
    "qarig_bmu_fwd_coarse": (I, [P, I, I, I, I, I, I, P, I, I, P, P, P, P]),
    "qarig_bmu_prepare_bytes": (Z, [I, I]),
    "qarig_bmu_prepare": (I, [P, I, I, P, P]),
    "qarig_gemm_workspace_bytes": (Z, [I, I, I]),
    "qarig_gemm_f32": (I, [P, L, I, P, L, I, P, L, I, I, I, P, P, L, P, L, I, P, L, I, I, I, P, P, Z,
                           P]),
    "qarig_gemm_tile64": (I, [I, I, I]),
    "qarig_gemm_x3_ok": (I, [I, I, I, I]),
    "qarig_gemm_grouped_supported": (I, [I, I, I, I]),
    "qarig_gemm_grouped_workspace_bytes": (Z, [I, I, I, I, I]),
    "qarig_gemm_f32_grouped": (I, [I, P, L, I, P, L, I, P, L, I, I, I, P, P, L, P, L, I, P, L, I, I, I, I, P,
                                   P, Z, P]),
    "qarig_gemm_lp_supported": (I, [I, I, I, I]),
    "qarig_gemm_lp_workspace_bytes": (Z, [I, I, I]),
    "qarig_gemm_lp": (I, [P, L, P, L, I, P, L, I, I, I, P, P, L, P, L, I, P, L, I, I, I, I, P, L, P, L, P, Z,
                          P]),
    "qarig_gemm_f8_supported": (I, [I, I, I]),
    "qarig_cast_fp8": (I, [P, L, P, P, P, P, P]),
    "qarig_gemm_f8": (I, [P, L, P, L, P, P, P, L, I, I, I, P, P, L, P, L, I, P, L, P, L, P]),
    "qarig_cast_colsum_workspace_bytes": (Z, [I, I]),
    "qarig_cast_colsum": (I, [P, L, I, I, I, P, P, I, P, Z, P]),
    "qarig_cast_bf16": (I, [P, P, L, P]),
    "qarig_cast_transpose_bf16": (I, [P, L, I, I, P, P]),
    "qarig_colsum_workspace_bytes": (Z, [I, I]),
    "qarig_colsum_f32": (I, [P, L, I, I, P, I, P, Z, P]),
    "qarig_patchify_fwd": (I, [P, I, I, I, I, I, I, P, P]),
    "qarig_unpatchify_fwd": (I, [P, I, I, I, I, I, I, P, P]),
    "qarig_codebook_gather_image": (I, [P, I, I, I, I, I, I, P, I, P, P, P]),
    "qarig_gather_rows": (I, [P, L, I, I, P, P, P, P]),
    "qarig_som_weights_fwd": (I, [P, L, I, F, P, P]),
    "qarig_som_band": (I, [P, I, I, F, I, P, P]),
    "qarig_index_histogram": (I, [P, L, I, P, P, P]),
    "qarig_posemb_fwd": (I, [P, I, I, P, P, P]),
    "qarig_assemble_tokens": (I, [P, I, P, I, I, I, I, I, P, I, P, P, P, P, P]),
    "qarig_embedding_fwd": (I, [P, I, I, I, I, P, P, P, P, P]),
    "qarig_embedding_bwd": (I, [P, I, I, I, P, P, P]),
    "qarig_layernorm_fwd": (I, [P, I, I, F, P, P, P, P, P, P, P, P, P]),
    "qarig_layernorm_bwd": (I, [P, P, P, P, P, P, P, I, I, P, P, P, P]),
    "qarig_rowmap_build": (I, [P, I, I, P, P, P, P, P]),
    "qarig_segment_sum": (I, [P, P, P, I, I, P, P]),
    "qarig_mul_rows_fwd": (I, [P, P, P, P, I, I, P]),
    "qarig_mul_rows_bwd": (I, [P, P, P, P, P, P, I, I, P]),
    "qarig_attention_fwd": (I, [P, P, P, I, I, I, I, I, I, F, P, P, P]),
    "qarig_attention_lp_fwd": (I, [P, P, P, I, I, I, I, I, I, F, P, P, P]),
    "qarig_attention_lp_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, P, P, P, P, P]),
    "qarig_attention_decode": (I, [P, P, P, P, P, I, I, I, I, P, I, L, F, P, P, P]),
    "qarig_gemm_grouped_skinny_f32": (I, [P, L, L, P, L, L, P, L, L, P, L, I, I, I, I, I, P]),
    "qarig_gemm_skinny_ln_f32": (I, [P, L, F, P, P, P, P, L, P, L, L, P, L, L, P, L, P, L, I, I, I, I, I, P]),
    "qarig_decode_linear_supported": (I, [I, I, I, I]),
    "qarig_decode_linear_f32": (I, [P, L, L, F, P, P, P, P, L, P, L, L, P, L, P, L, P, L, P, L, L, I, I, I, I, I,
                                    P]),
    "qarig_decode_embed": (I, [P, I, I, I, P, P, P, I, I, P, L, P, P, P, P]),
    "qarig_decode_attention": (I, [P, P, P, P, P, I, I, I, I, P, I, L, L, L, F, P, L, P, P]),
    "qarig_decode_sample": (I, [P, L, I, I, F, I, I, L, P, P, P, I, I, I, I, I, P, P, P, P, P]),
    "qarig_decode_decide": (I, [P, I, I, I, I, P, P, P, P, P, P]),
    "qarig_decode_rows": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "qarig_decode_commit": (I, [P, I, I, I, P, P, L, P, P]),
    "qarig_decode_advance": (I, [P, I, P]),
    "qarig_attention_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, F, P, P, P, P, P]),
    "qarig_cross_entropy_fwd": (I, [P, P, I, I, P, P, P, P, P]),
    "qarig_mse_workspace_bytes": (Z, []),
    "qarig_mse_fwd": (I, [P, P, L, P, P, P, P]),
    "qarig_adam_step": (I, [P, P, P, P, L, F, F, F, F, F, F, P, P, P]),
    "qarig_mul_fwd": (I, [P, P, P, L, P]),
    "qarig_mul_bwd": (I, [P, P, P, P, P, L, P]),
    "qarig_act_fwd": (I, [P, P, L, I, P]),
    "qarig_act_bwd": (I, [P, P, P, L, I, P]),
    "qarig_scale_by": (I, [P, P, P, L, P]),
    "qarig_conv2d_fwd": (I, [P, I, I, I, I, P, P, I, I, I, I, I, P, P, P]),
    "qarig_conv2d_fwd_workspace_bytes": (Z, [I, I, I]),
    "qarig_conv2d_fwd_workspace_bytes_n": (Z, [I, I, I, I, I, I, I]),
    "qarig_conv2d_fwd_ws": (I, [P, I, I, I, I, P, P, I, I, I, I, I, P, P, P, Z, I, P]),
    "qarig_conv_transpose2d_workspace_bytes": (Z, [I, I]),
    "qarig_conv_transpose2d_workspace_bytes_n": (Z, [I, I, I, I, I]),
    "qarig_conv_transpose2d_fwd": (I, [P, I, I, I, I, P, P, I, I, P, P, P, Z, I, P]),
    "qarig_conv2d_bwd_data_workspace_bytes": (Z, [I, I, I]),
    "qarig_conv2d_bwd_data_workspace_bytes_n": (Z, [I, I, I, I, I, I, I]),
    "qarig_conv_transpose2d_bwd_data_workspace_bytes_n": (Z, [I, I, I, I, I]),
    "qarig_conv2d_bwd_data": (I, [P, I, I, I, I, P, I, I, I, I, I, I, P, P, Z, P]),
    "qarig_conv_transpose2d_bwd_data": (I, [P, I, I, I, I, P, I, P, P]),
    "qarig_conv_transpose2d_bwd_data_ws": (I, [P, I, I, I, I, P, I, P, P, Z, P]),
    "qarig_conv_wgrad_workspace_bytes": (Z, [I, I, I]),
    "qarig_conv_wgrad": (I, [P, I, I, I, I, P, I, I, I, I, I, I, P, P, Z, P]),
    "qarig_conv_bias_grad": (I, [P, I, I, I, P, P]),
}

_lib = None


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with "
            "`python quantized-autoregression-image-generator_amd/build.py` "
            "(there is no CPU fallback for the qarig ops)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    for name in OPTIONS:      # QARIG_<OPTION>=<int> in the environment seeds an option (tools, A/B runs)
        v = os.environ.get("QARIG_" + name.upper())
        if v is not None:
            lib.qarig_set_option(name.encode(), int(v))
    return lib


OPTIONS = ("gemm_dma", "gemm_pair", "bmu_cs", "bmu_groups", "bmu_coarse", "attn_qw", "attn_bw", "lp_big",
           "lp_mfma16", "convt_pair", "conv_ring", "gemm_xcd_splits", "decode_stream", "decode_rows", "gemm_tile64", "gemm_x3")


OPTION_EPOCH = 0     # bumped by set_option: cached artefacts whose layout depends on the kernel family carry it in their key


def set_option(name, value):
    """Kernel-selection option (include/qarig.h qarig_set_option); returns the previous value."""
    global OPTION_EPOCH
    OPTION_EPOCH += 1
    old = load().qarig_set_option(name.encode(), int(value))
    if old == -2 ** 31:
        raise KeyError(f"{name}: {last_error()}")
    return old


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().qarig_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


N_CALLS = 0      # C-ABI calls checked so far (pipeline._SegmentedCapture: "has this segment launched anything?")


def check(status, what):
    global N_CALLS
    N_CALLS += 1
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Handle of torch's current HIP stream on the current device (what every launch uses)."""
    if _raw_stream is not None:      # same value as below without building a Stream object
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "qarig ops run on MI355X only: got a CPU tensor "
                "(the CPU restatement lives in oracle/ and is test infrastructure)")


def f32c(t):
    """fp32 + contiguous view/copy of t (no-op when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_ws_cache = {}
_ws_retired = []     # outgrown buffers whose address a captured graph may still hold


def workspace(nbytes, device, tag="default"):
    """Grow-only per-device scratch buffer (uint8).  Kernels that use it run on the
    current stream in issue order, so one buffer per tag can be shared.

    A captured HIP graph (GraphedTrainStep, the decode graphs) bakes the buffer's raw address
    into its kernel nodes.  An outgrown buffer is therefore never handed back to the caching
    allocator: it is parked in `_ws_retired`, so a later replay of an older graph writes into
    memory nobody else owns (a few MB per growth step, bounded by the largest request)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
