// Can the SiLU + gate split + state image work of ONE 32-edge half of an edge tile ride between the MFMAs of the scalar GEMM of the OTHER
// half, issued by the same wave, with two such waves per SIMD?  (round 4: the "N-skewed" schedule of k_edge_msg_x3 -- DESIGN.md 3.4.)
//
// One phase of the skewed schedule, per wave: 18 k-blocks x 3 MFMAs (v_mfma_f32_32x32x16_f16: A.Bh -> am, A.Bl -> al, Alo.Bh -> al; B operands
// from LDS, A operands streamed from L2 in BOTH phases: 2 KB per k-block and wave for 3 MFMAs) and the "finish" work of the other half's previous result (16 accumulator values per lane):
// merge am + al * inv, SiLU in scaled units (exp2, add, rcp, mul), hi / lo' split of the activations (gate operand), 6 gate MFMAs, residual
// add, hi / lo' split of the new state and its image stores (ds_write_b64).   ~210 VALU + 8 LDS stores + 6 MFMAs.
//   V0: GEMM only        V1: finish only        V2: GEMM, then finish (today's order)        V3: finish interleaved, 1 MFMA : 4 others
//   NACC = 2: am / al as in the kernel (al gets two MFMAs per k-block);  NACC = 3: three accumulators (no back-to-back same-accumulator MFMAs)
// 256 workgroups x 512 threads (2 waves per SIMD); prints shader cycles per phase (mean over workgroups of the slowest wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split16x2(float x0, float x1, h2& hi, h2& lo, float pre, float neg) {
    uint32_t hiu, lou;
    const float t0 = x0 * pre, t1 = x1 * pre;
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t0), "v"(t1));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    __builtin_memcpy(&hi, &hiu, 4);
    __builtin_memcpy(&lo, &lou, 4);
}

constexpr int KB = 18, TP = 65;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WPool {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    __device__ __forceinline__ h8 ld(uint32_t soff) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        h8 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    }
};

template <int VAR, int NACC, int PER>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void kb(int n, float pre, float neg, float inv, const h8* __restrict__ W, float* out, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h8* XH = (h8*)smem;                      // [36][65] images
    h8* XL = XH + 36 * TP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 36 * TP; i += 512) {
        h8 v;
        for (int s = 0; s < 8; ++s) v[s] = (_Float16)(0.001f * ((i + s) & 63));
        XH[i] = v;
    }
    __syncthreads();
    // (a first version kept the 36 A operands of the wave's M-tile resident in registers across the two phases of a pair -- 144 registers: with two
    //  waves per SIMD a wave has 256, the rest of the kernel does not fit beside them; so the weights are streamed again in the second phase)
    h8 ah[3], alo[3];
    WPool wp;
    wp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(W), 0, 8 * 2 * (KB + 4) * 1024, 0x00020000);
    wp.voff = (uint32_t)lane * 16u;
    const uint32_t wo = (uint32_t)wave * 2 * (KB + 4) * 1024;
    h8 gwh[2], gwl[2];
    for (int j = 0; j < 2; ++j)
        for (int s = 0; s < 8; ++s) { gwh[j][s] = (_Float16)(0.003f * (lane + j)); gwl[j][s] = (_Float16)(0.0001f * (s + j)); }
    f32x16 pam, pal, st;                     // the other half's previous result and its fp32 state
    for (int r = 0; r < 16; ++r) { pam[r] = 0.01f * (lane + r); pal[r] = 0.3f * r; st[r] = 0.1f * r; }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int boff = (lane >> 5) * TP + (lane & 31);
    const int half = lane >> 5, l31 = lane & 31;
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
      for (int ph = 0; ph < 2; ++ph) {
        f32x16 am = zero, al = zero, al2 = zero, gm = zero, gl = zero;
        auto gemm = [&](auto ldc) {
            constexpr bool LD = decltype(ldc)::value;
            h8 bh = XH[boff], bl = XL[boff];
#pragma unroll
            for (int k = 0; k < 2; ++k) { ah[k] = wp.ld(wo + (2 * k) * 1024); alo[k] = wp.ld(wo + (2 * k + 1) * 1024); }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                { ah[(k + 2) % 3] = wp.ld(wo + (2 * (k + 2)) * 1024); alo[(k + 2) % 3] = wp.ld(wo + (2 * (k + 2) + 1) * 1024); }
                const h8 nbh = XH[boff + (k + 1 < KB ? k + 1 : k) * 2 * TP], nbl = XL[boff + (k + 1 < KB ? k + 1 : k) * 2 * TP];
                am = MFMA16(ah[k % 3], bh, am);
                al = MFMA16(ah[k % 3], bl, al);
                if (NACC == 3) al2 = MFMA16(alo[k % 3], bh, al2); else al = MFMA16(alo[k % 3], bh, al);
                bh = nbh; bl = nbl;
            }
        };
        auto finish = [&] {
            f32x16 act;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = pam[r] + pal[r] * inv;
                act[r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {           // gate contraction over the wave's 32 channels: 2 k-blocks
                h8 bh, bl;
#pragma unroll
                for (int s = 0; s < 8; s += 2) {
                    h2 hi, lo;
                    split16x2(act[8 * j + s], act[8 * j + s + 1], hi, lo, pre, neg);
                    bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
                }
                asm("s_nop 1" : "+v"(bh), "+v"(bl));
                gm = MFMA16(gwh[j], bh, gm);
                gl = MFMA16(gwh[j], bl, gl);
                gl = MFMA16(gwl[j], bh, gl);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] += act[r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // state images of this wave's channels: 4 x (ds_write_b64 hi, ds_write_b64 lo')
                h4 vh, vl;
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    h2 hi, lo;
                    split16x2(st[4 * q + t], st[4 * q + t + 1], hi, lo, pre, neg);
                    vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
                }
                const int off = ((18 + (wave & 7) * 2 + (q >> 1)) * TP + 32 + l31) * 16 + 8 * half + 0 * q;   // groups the GEMM above does not read
                *(h4*)((char*)XH + off) = vh;
                *(h4*)((char*)XL + off) = vl;
            }
        };
        auto gemm2 = [&] { if (ph == 0) gemm(std::true_type{}); else gemm(std::false_type{}); };
        if (VAR == 0) gemm2();
        if (VAR == 1) finish();
        if (VAR == 2) { gemm2(); __builtin_amdgcn_sched_barrier(0); finish(); }
        if (VAR == 3) {
            if (ph == 0) gemm(std::true_type{}); else gemm(std::false_type{});
            finish();
#pragma unroll
            for (int i = 0; i < 3 * KB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x200 | 0x100, PER, 0);       // VALU, DS write, DS read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) { pam[r] = am[r] * 1e-3f + gm[r] * 1e-6f + 0.5f; pal[r] = (al[r] + al2[r]) * 1e-3f + gl[r] * 1e-6f; st[r] *= 0.5f; }
        __syncthreads();
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 16; ++r) sink += pam[r] + pal[r] + st[r];
    out[blockIdx.x * 512 + tid] = sink;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR, int NACC, int PER>
void run(const char* name) {
    const int blocks = 256, n = 300;
    float* out; unsigned long long* ticks; h8* W;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    (void)hipMalloc(&W, 8 * 2 * (KB + 4) * 1024); (void)hipMemset(W, 0x11, 8 * 2 * (KB + 4) * 1024);
    const size_t lds = 2 * 36 * TP * 16;
    (void)hipFuncSetAttribute((const void*)kb<VAR, NACC, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kb<VAR, NACC, PER>), dim3(blocks), dim3(512), lds, 0, 5, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((kb<VAR, NACC, PER>), dim3(blocks), dim3(512), lds, 0, n, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * blocks);
    (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < 8; ++w) m = m > h[i * 8 + w] ? m : h[i * 8 + w]; mx += m; }
    mx /= blocks;
    printf("%-52s %8.0f clk per PAIR of phases (MFMA floor per SIMD: GEMM %d%s)\n", name, mx / n, 2 * 2 * 54 * 32, VAR >= 1 ? " + gate 768" : "");
    (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(W);
}

int main() {
    run<0, 2, 0>("V0 GEMM only, 2 accumulators");
    run<0, 3, 0>("V0 GEMM only, 3 accumulators");
    run<1, 2, 0>("V1 finish only");
    run<2, 2, 0>("V2 GEMM then finish, 2 acc");
    run<2, 3, 0>("V2 GEMM then finish, 3 acc");
    run<3, 2, 3>("V3 interleaved 1 MFMA : 3, 2 acc");
    run<3, 2, 4>("V3 interleaved 1 MFMA : 4, 2 acc");
    run<3, 2, 5>("V3 interleaved 1 MFMA : 5, 2 acc");
    run<3, 3, 4>("V3 interleaved 1 MFMA : 4, 3 acc");
    run<3, 3, 5>("V3 interleaved 1 MFMA : 5, 3 acc");
    run<3, 3, 6>("V3 interleaved 1 MFMA : 6, 3 acc");
    return 0;
}
