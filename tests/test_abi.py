"""The C-ABI shared library loads without a GPU and exports every symbol that
include/qarig.h declares; the ctypes table in qarig/_lib.py covers the same set."""
import ctypes
import os
import re

from conftest import PKG, ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "qarig.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qarig_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    import build as qbuild
    so = qbuild.build_lib(verbose=False)
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/qarig.h but not exported"
    from qarig import _lib
    assert sorted(_lib.SIGNATURES) == names
    h = _lib.load()
    assert h.qarig_version() >= 100
    assert h.qarig_target_arch() == b"gfx950"
    # argument validation works without touching a GPU
    assert h.qarig_gemm_f32(None, 0, 1, None, 0, 1, None, 0, 1, 1, 1, None, None, 0, None, 0, 0, None,
                            0, 0, 1, 0, None, None, 0, None) == -1
    assert "null operand" in _lib.last_error()


def test_code_object_targets_gfx950_only():
    import subprocess
    so = os.path.join(PKG, "lib", "libqarig_hip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", so], capture_output=True,
                         text=True).stdout
    assert ".hip_fatbin" in out
    strings = subprocess.run(["strings", "-n", "6", so], capture_output=True, text=True).stdout
    assert "gfx950" in strings and "gfx942" not in strings and "sm_" not in strings


def test_product_never_imports_oracle():
    bad = []
    for dp, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(f)
    assert not bad, f"product files import the oracle: {bad}"


def test_cpu_tensor_refused_by_product_modules():
    import pytest
    import torch
    from models.Codebook import Codebook
    from models.Transformer import Transformer
    cb = Codebook(patch_dim=(2, 2), image_dim=(8, 8), image_channel=4, num_embeddings=16)
    with pytest.raises(RuntimeError):
        cb.get_patches_bmu(torch.zeros(1, 4, 8, 8))
    m = Transformer(use_encoder=False, use_pos_cond=False, num_dec_layers=1, num_dec_embedding=8,
                    self_attn_heads=2, transformer_in_dim=16, transformer_out_dim=9,
                    transformer_hidden_dim=32)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, dtype=torch.long))
