"""CPU: the C-ABI library builds/loads and exports every symbol include/gomatching_hip.h declares (no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gomatching_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gom_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    import __graft_entry__ as entry
    entry.build()
    from gomatching_amd import lib
    handle = lib.load()
    names = _declared()
    assert len(names) >= 28
    for n in names:
        assert hasattr(handle, n), "symbol %s declared in the header but not exported" % n
        assert n in lib.SIGNATURES, "symbol %s has no ctypes signature" % n
    for n in lib.SIGNATURES:
        assert n in names, "bound symbol %s is not declared in include/gomatching_hip.h" % n
    assert handle.gom_abi_version() == 1


def test_product_has_no_oracle_dependency():
    """The shipped package must never import the oracle (or /root/reference)."""
    pkg = os.path.join(ROOT, "gomatching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "smoke.py":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
                assert "/root/reference" not in re.sub(r'""".*?"""', "", src, flags=re.S), f


def test_missing_library_fails_loudly(monkeypatch):
    from gomatching_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libgomatching_hip.so")
    with pytest.raises(lib.GomError):
        lib.load()


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument checks run before any HIP call: the newer entry points turn nonsense into GOM_ERR_INVALID_ARG /
    GOM_ERR_UNSUPPORTED instead of launching (error behaviour of the boundary; no compute, no device needed)."""
    import ctypes
    from gomatching_amd import lib
    L = lib.load()
    INVALID = 1                                                   # GOM_ERR_INVALID_ARG (include/gomatching_hip.h)
    p = ctypes.c_void_p(0x1000)                                   # a non-null, 16-byte aligned address that is never dereferenced
    assert L.gom_flash_attention_f32(p, p, p, p, 1, 64, 2, 32, 192, 64, None, None) == INVALID         # head_dim 32: not served
    assert L.gom_flash_attention_f32(None, p, p, p, 1, 64, 2, 64, 384, 128, None, None) == INVALID
    assert L.gom_flash_attention_f32(p, p, p, p, 1, 64, 2, 64, 100, 128, None, None) == INVALID         # row stride < 3C
    assert L.gom_im2col_nhwc_f32(p, p, 1, 8, 8, 3, 3, 3, 1, 1, 1, 32, None) == INVALID                  # C % 4 != 0
    assert L.gom_im2col_nhwc_f32(p, p, 1, 8, 8, 4, 3, 3, 1, 1, 1, 20, None) == INVALID                  # ldo < KH*KW*C
    assert L.gom_grouped_conv3x3_nhwc_f32(p, p, None, p, None, 0, p, 1, 8, 8, 64, 64, 8, 1, None) == INVALID   # 8 channels per group: only 4 and 16 are built
    assert L.gom_grouped_conv3x3_nhwc_f32(p, p, None, p, None, 7, p, 1, 8, 8, 64, 64, 16, 1, None) == INVALID  # unknown activation
    assert L.gom_vitae_window_attention_f32(p, p, 4, 3, 96, None) == INVALID                             # head_dim 32
    assert L.gom_softmax_rows_scaled_f32(p, 4, 9000, 9000, 1.0, None) == INVALID                         # > 8192 columns
    assert L.gom_transpose_f32(p, p, 8, 8, 4, 8, None) == INVALID                                        # ld < cols
    assert L.gom_copy_words(None, p, 4, None) == INVALID
    assert L.gom_tracker_create(0, 0.2, 1, 1, 1, 1.0, None, 0, None, 0, 1024, 8, 1024) is None           # test_len < 1
    assert L.gom_tracker_create(6, 0.2, 1, 1, 1, 1.0, None, 1, None, 0, 1024, 8, 1024) is None           # layers without weights
    # round 6: the CU-cooperative decoder tail
    assert L.gom_dec_tail2_wave_bytes(256, 1024, 1, 1, 4) == (64 + 12 * 64) * 1024 and L.gom_dec_tail2_wave_bytes(256, 1000, 1, 1, 4) == -1
    assert L.gom_dec_tail2_wave_bytes(256, 1024, 1, 1, 8) == (32 + 12 * 32) * 1024 and L.gom_dec_tail2_wave_bytes(256, 1024, 1, 1, 6) == -1   # 4 or 8 waves
    assert L.gom_dec_tail2_image_lin(p, 1, 128, p, 1 << 20, 0, 8, None) == INVALID                       # ld < 256
    assert L.gom_dec_tail2_image_lin(p, 1, 256, p, 1 << 20, 0, 5, None) == INVALID                       # waves
    assert L.gom_dec_tail2_image_mlp(p, 1, 256, p, 1, 256, 192, p, 1 << 20, 0, 8, None) == INVALID        # hidden % 128 != 0
    # round 6: the self-attention blocks on 16-token waves (same contract as gom_dec_attn_*)
    assert L.gom_dec_attn2_image_bytes(256, 8) == (4 + 32 * 33) * 1024 and L.gom_dec_attn2_image_bytes(128, 8) == -1
    assert L.gom_dec_attn2_raw_image_bytes() == (4 + 44 * 33) * 1024
    assert L.gom_dec_attn2_image(p, 1, 128, p, p, p, 1, 256, p, p, p, p, 0, p, 1 << 22, None) == INVALID    # ld_in < 256
    assert L.gom_dec_attn2_image(p, 1, 256, p, p, p, 1, 256, p, p, p, p, 0, p, 1024, None) == INVALID       # image too small
    assert L.gom_dec_attn2_f32(p, 256, None, 0, p, 1e-5, p, 256, 4, 25, 1, 0, None, None) == INVALID        # intra without query_pos
    assert L.gom_dec_attn2_f32(p, 256, p, 256, p, 1e-5, p, 256, 4, 33, 1, 0, None, None) == INVALID         # > 32 points
    assert L.gom_dec_attn2_f32(p, 256, None, 0, p, 1e-5, p, 256, 4, 129, 25, 1, None, None) == INVALID      # > 128 queries
    assert L.gom_dec_attn2_raw_f32(p, 256, p, 1e-5, p, 256, p, 256, p, 380, 4, 100, 25, None, None) == INVALID   # ldraw < 384


def test_library_has_no_packed_fp32_instructions():
    """DESIGN.md "Tracker determinism": `v_pk_fma_f32` gave wrong low halves beside the bf16x6 GEMM kernel on MI355X, so
    the device code is built with the packed-fp32 target feature off.  Checked on the disassembly of the shipped library."""
    import shutil
    import subprocess
    from gomatching_amd import build
    lib = build.build()
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    if not (os.path.exists(objdump) and os.path.exists(bundler) and os.path.exists(objcopy)):
        pytest.skip("ROCm llvm tools not found")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        total = 0
        for src in build.SOURCES:
            if not src.endswith(".hip") or src == "abi.hip":       # abi.hip holds no kernel
                continue
            obj = os.path.join(build.OBJ, src + ".o")
            fat, co = os.path.join(tmp, src + ".fatbin"), os.path.join(tmp, src + ".co")
            r = subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True, text=True)
            if r.returncode != 0 and "not found" in r.stderr:      # host-only translation unit (no kernel)
                continue
            assert r.returncode == 0 and os.path.getsize(fat) > 0, (src, r.stderr)
            r = subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + fat, "--output=" + co], capture_output=True, text=True)
            assert r.returncode == 0 and os.path.getsize(co) > 0, (src, r.stderr)
            dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            assert "s_endpgm" in dis, src
            bad = [l for l in dis.splitlines() if "v_pk_fma_f32" in l or "v_pk_add_f32" in l or "v_pk_mul_f32" in l]
            assert not bad, (src, bad[:3])
            total += dis.count("s_endpgm")
        assert total > 50
    assert os.path.exists(lib)
