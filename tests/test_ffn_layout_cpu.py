"""CPU model of the fused FFN kernel's data flow (csrc/ffn_fused.hip): the fragment-linear weight image, the transposed
products and the accumulator-as-operand hand-over are replayed in numpy with the MFMA operand maps of
cdna_hip_programming.md §3 (v_mfma_f32_32x32x16: A lane (r,h) = A[r][8h+j], B lane (r,h) = B[8h+j][r],
D lane (r,h) reg g = D[(g&3)+8(g>>2)+4h][r]) and must reproduce relu(X W1^T + b1) W2^T exactly (fp64 arithmetic, so any
index slip shows as an O(1) error).  The GPU test (tests/test_ffn_gpu.py) checks the kernel itself."""
import numpy as np

D, CH, F = 256, 32, 96


def mfma(A_frag, B_frag, acc):
    """A_frag, B_frag: [64 lanes][8]; acc [64 lanes][16] -> acc + A.B in the hardware's layout."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        r, h = l & 31, l >> 5
        A[r, 8 * h:8 * h + 8] = A_frag[l]
        B[8 * h:8 * h + 8, r] = B_frag[l]
    Dm = A @ B
    out = acc.copy()
    for l in range(64):
        r, h = l & 31, l >> 5
        for g in range(16):
            out[l, g] += Dm[(g & 3) + 8 * (g >> 2) + 4 * h, r]
    return out


def build_image(W1, W2):
    """The element map of ffn_image_kernel (one plane, unscaled): [chunks][64 frags][64 lanes][8]."""
    chunks = F // CH
    img = np.zeros((chunks, 64, 64, 8))
    for c in range(chunks):
        for f in range(64):
            for l in range(64):
                r, h = l & 31, l >> 5
                for j in range(8):
                    if f < 32:
                        s = f >> 1
                        img[c, f, l, j] = W1[CH * c + r, 16 * s + 8 * h + j]
                    else:
                        idx = f - 32
                        t, u = idx >> 2, (idx >> 1) & 1
                        img[c, f, l, j] = W2[32 * t + r, CH * c + 16 * u + 8 * (j >> 2) + 4 * h + (j & 3)]
    return img


def test_fused_ffn_dataflow_reproduces_the_ffn():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((32, D))
    W1 = rng.standard_normal((F, D)) * 0.1
    b1 = rng.standard_normal(F)
    W2 = rng.standard_normal((D, F)) * 0.1
    img = build_image(W1, W2)
    xf = np.zeros((16, 64, 8))                                 # B fragments of X: lane (r,h) holds X[r][16 s + 8 h + j]
    for s in range(16):
        for l in range(64):
            r, h = l & 31, l >> 5
            xf[s, l] = X[r, 16 * s + 8 * h:16 * s + 8 * h + 8]
    acc2 = np.zeros((8, 64, 16))
    for c in range(F // CH):
        acc1 = np.zeros((64, 16))
        for s in range(16):
            acc1 = mfma(img[c, 2 * s], xf[s], acc1)            # plane 0 only in this model (planes share the element map)
        hf = np.zeros((2, 64, 8))
        for l in range(64):
            h = l >> 5
            for g in range(16):
                hid = CH * c + (g & 3) + 8 * (g >> 2) + 4 * h  # the aux fragment is read at [8 q + 4 h + e], q = g >> 2
                hf[g >> 3, l, g & 7] = max(acc1[l, g] + b1[hid], 0.0)
        for t in range(8):
            for u in range(2):
                acc2[t] = mfma(img[c, 32 + 4 * t + 2 * u], hf[u], acc2[t])
    Y = np.zeros((32, D))
    for l in range(64):
        r, h = l & 31, l >> 5
        for t in range(8):
            for g in range(16):
                Y[r, 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h] = acc2[t, l, g]
    ref = np.maximum(X @ W1.T + b1, 0) @ W2.T
    assert np.abs(Y - ref).max() < 1e-9


def test_epilogue_staging_is_a_bijection_without_bank_conflicts():
    """Y^T -> LDS: lane (r,h) writes chunk (8t + 2q + h) ^ (r & 7) of row r; the row pass reads chunk lane ^ (row & 7)."""
    for r in range(32):
        seen = set()
        for h in range(2):
            for t in range(8):
                for q in range(4):
                    seen.add((8 * t + 2 * q + h) ^ (r & 7))
        assert seen == set(range(64))
    for h in range(2):                                         # ds_write_b128: 8 consecutive lanes must hit 8 distinct 16-B slots mod 128 B
        for t in range(8):
            for q in range(4):
                for r0 in range(0, 32, 8):
                    slots = {((8 * t + 2 * q + h) ^ (r & 7)) % 8 for r in range(r0, r0 + 8)}
                    assert len(slots) == 8
