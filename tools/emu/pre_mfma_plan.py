"""Index-plan check for moving the residual GCP2 pre-phase of k_edge_msg_x3 onto v_mfma_f32_16x16x32_f16: a NumPy emulation of
what each LANE loads, what the MFMA returns to it under the documented operand maps, and what it stores, compared with the direct formula.

    A (16x32):  lane l holds row l&15,  k = (l>>4)*8 + j, j < 8          (8 f16 per lane)
    B (32x16):  lane l holds col l&15,  k = (l>>4)*8 + j
    D (16x16):  lane l holds col l&15,  rows (l>>4)*4 + i, i < 4          (cdna_hip_programming.md, fragment layout)

No GPU involved: this validates the data movement the kernel variant implements, not the kernel.
"""
import numpy as np

rng = np.random.default_rng(0)
ET, TP, H, V = 32, 33, 8, 32
ROWS = H + 3
wdd = rng.normal(size=(ROWS, V)).astype(np.float32)            # [W_down (8) ; W_frames (3)] x 32 channels
VV = rng.normal(size=(V * 3, TP)).astype(np.float32)           # LDS: VV[(c*3+k)*TP + e]
FR = rng.normal(size=(9, TP)).astype(np.float32)               # LDS: FR[r*TP + e]


def f16split(x):
    """x = hi + 2^-11 lo', images pre-scaled by 2^-8 (gcdm_edge_x3.hip.h split16)."""
    xs = x.astype(np.float32) * np.float32(1 / 256)
    hi = xs.astype(np.float16)
    lo = ((xs - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo


def wsplit(w):
    """host split_f16: weights carry 2^8."""
    ws = w.astype(np.float32) * np.float32(256)
    hi = ws.astype(np.float16)
    lo = ((ws - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo


# ---- host packing: A operand of the single k-block, [64 lanes][8] ----
Apad = np.zeros((16, V), np.float32)
Apad[:ROWS] = wdd
Ahi = np.zeros((64, 8), np.float16)
Alo = np.zeros((64, 8), np.float16)
for l in range(64):
    for j in range(8):
        Ahi[l, j], Alo[l, j] = wsplit(Apad[l & 15, (l >> 4) * 8 + j])


def mfma_16x16x32(a, b, c):
    """a, b: [64][8] per-lane operands; c: [64][4] accumulators.  Returns d [64][4] under the maps above (fp32 accumulate)."""
    A = np.zeros((16, 32), np.float32)
    B = np.zeros((32, 16), np.float32)
    for l in range(64):
        for j in range(8):
            A[l & 15, (l >> 4) * 8 + j] = np.float32(a[l, j])
            B[(l >> 4) * 8 + j, l & 15] = np.float32(b[l, j])
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    d = c.copy()
    for l in range(64):
        for i in range(4):
            d[l, i] += D[(l >> 4) * 4 + i, l & 15]
    return d


n_out = np.zeros((H, ET), np.float32)        # norms -> extended-K rows
q_out = np.zeros((9, ET), np.float32)        # frame scalars
VH = np.zeros((H * 3, TP), np.float32)
for wave in range(ET // 16):                 # waves 0..1 work, the others idle
    out = np.zeros((3, 64, 4), np.float32)
    for k in range(3):
        bh = np.zeros((64, 8), np.float16)
        bl = np.zeros((64, 8), np.float16)
        for l in range(64):
            e = 16 * wave + (l & 15)
            for j in range(8):
                c = (l >> 4) * 8 + j
                bh[l, j], bl[l, j] = f16split(VV[c * 3 + k, e])       # 8 ds_read_b32 + 4 split16x2 per lane and component
        am = mfma_16x16x32(Ahi, bh, np.zeros((64, 4), np.float32))
        al = mfma_16x16x32(Ahi, bl, np.zeros((64, 4), np.float32))
        al = mfma_16x16x32(Alo, bh, al)
        out[k] = am + al * np.float32(1 / 2048)
    for l in range(64):
        e, q = 16 * wave + (l & 15), l >> 4
        if q < 2:
            for i in range(4):
                r = 4 * q + i
                v = out[:, l, i]
                n_out[r, e] = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + np.float32(1e-8)) + np.float32(1e-8)
                for k in range(3):
                    VH[r * 3 + k, e] = v[k]
        elif q == 2:
            f = FR[:, e]
            for j in range(3):                                        # W_frames row j -> u_j
                u = out[:, l, j]
                for r in range(3):
                    q_out[3 * j + r, e] = f[3 * r] * u[0] + f[3 * r + 1] * u[1] + f[3 * r + 2] * u[2]

# ---- direct formula (gcp2_pre_x3 / gcpnet.py:442-459, scalarize components/__init__.py:174-219) ----
ref = np.einsum("rc,cke->rke", wdd.astype(np.float64), VV.reshape(V, 3, TP)[:, :, :ET].astype(np.float64))
n_ref = np.sqrt((ref[:H] ** 2).sum(1) + 1e-8) + 1e-8
q_ref = np.zeros((9, ET))
for j in range(3):
    for r in range(3):
        q_ref[3 * j + r] = (FR[3 * r:3 * r + 3, :ET].astype(np.float64) * ref[H + j]).sum(0)
err = max(np.abs(n_out - n_ref).max() / np.abs(n_ref).max(), np.abs(q_out - q_ref).max() / np.abs(q_ref).max(),
          np.abs(VH[:, :ET].reshape(H, 3, ET) - ref[:H]).max() / np.abs(ref[:H]).max())
print(f"max relative error vs fp64 direct formula: {err:.2e}")
assert err < 5e-7
print("plan ok")
