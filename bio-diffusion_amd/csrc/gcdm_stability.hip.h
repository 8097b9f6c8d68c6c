// gcdm_stability.hip.h -- molecular stability of a sampled batch on the device (gfx950).
//
// One wave per molecule.  Lane i owns atoms i, i+64, ...: for each it walks all atoms j of the molecule (positions and types of
// the molecule are staged in LDS for n <= STAB_LDS_ATOMS, read through L1 otherwise), classifies the pair by the three
// length + margin thresholds (later thresholds overwrite earlier ones, edm/__init__.py:76-81), sums the bond orders of the row
// (diagonal excluded, :108-109) and tests the valence against the type's allowed set (:111-117).  Integer work on 16 B per atom:
// HBM/latency bound, ~ microseconds for a whole batch; it exists so that the evaluation loop never leaves the device.
//
// Distances: sqrt(dx^2 + dy^2 + dz^2) in fp32, sequentially summed without FMA contraction -- the direct formula ATen's cdist uses
// for n <= 25; for n > 25 the reference switches to the matmul expansion, whose result differs in the last bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gcdm_hip.h"

constexpr int STAB_LDS_ATOMS = 1024;

__global__ __launch_bounds__(64) void k_stability(GcdmBondTables tb, const float* __restrict__ x, long stride,
                                                  const int32_t* __restrict__ types, const int32_t* __restrict__ off,
                                                  int32_t* __restrict__ out, const int64_t* __restrict__ pair_off = nullptr,
                                                  uint8_t* __restrict__ orders = nullptr) {
    __shared__ float sx[STAB_LDS_ATOMS], sy[STAB_LDS_ATOMS], sz[STAB_LDS_ATOMS];
    __shared__ int st[STAB_LDS_ATOMS];
    const int m = blockIdx.x, lane = threadIdx.x;
    const int a0 = off[m], n = off[m + 1] - a0;
    const bool in_lds = n <= STAB_LDS_ATOMS;
    if (in_lds) {
        for (int i = lane; i < n; i += 64) {
            const float* p = x + (long)(a0 + i) * stride;
            sx[i] = p[0]; sy[i] = p[1]; sz[i] = p[2];
            st[i] = types[a0 + i];
        }
        __syncthreads();
    }
    int stable = 0;
    for (int i = lane; i < n; i += 64) {
        float xi, yi, zi; int ti;
        if (in_lds) { xi = sx[i]; yi = sy[i]; zi = sz[i]; ti = st[i]; }
        else { const float* p = x + (long)(a0 + i) * stride; xi = p[0]; yi = p[1]; zi = p[2]; ti = types[a0 + i]; }
        int nb = 0;
        for (int j = 0; j < n; ++j) {
            float xj, yj, zj; int tj;
            if (in_lds) { xj = sx[j]; yj = sy[j]; zj = sz[j]; tj = st[j]; }
            else { const float* p = x + (long)(a0 + j) * stride; xj = p[0]; yj = p[1]; zj = p[2]; tj = types[a0 + j]; }
            const float dx = xi - xj, dy = yi - yj, dz = zi - zj;
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            const float d = __fmul_rn(100.0f, __fsqrt_rn(d2));          // "we change the metric" (:69), fp32 like the reference
            const int k = ti * GCDM_STABILITY_MAX_TYPES + tj;
            int o = d < tb.thr1[k] ? 1 : 0;
            o = d < tb.thr2[k] ? 2 : o;
            o = d < tb.thr3[k] ? 3 : o;
            if (tb.limit_bonds_to_one && o > 1) o = 1;
            if (orders) orders[pair_off[m] + (int64_t)i * n + j] = (uint8_t)((j == i) ? 0 : o);     // n x n bond-order matrix (get_bond_order_batch, :61-87)
            nb += (j == i) ? 0 : o;
        }
        stable += (nb < 32 && ((tb.allowed_mask[ti] >> nb) & 1u)) ? 1 : 0;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) stable += __shfl_xor(stable, s);
    if (lane == 0 && out) {
        out[3 * m + 0] = (stable == n) ? 1 : 0;
        out[3 * m + 1] = stable;
        out[3 * m + 2] = n;
    }
}
