// Issue cost of the instruction kinds the VALU phases of the edge kernel are made of (gfx950), one and two waves per SIMD:
//   v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_exp_f32, v_rcp_f32, v_fma_mixlo_f16, a SiLU chain in scalar and in packed form,
//   ds_write_b64 / ds_write_b128 / ds_read_b128.
// Every kind runs N x 32 independent instructions per wave (32 register chains, so dependent-issue latency is hidden); reported: cycles per
// instruction as seen by one wave (s_memtime), with 1 wave per SIMD (256 threads) and with 2 (512 threads).
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_ubench8 tools/valu_ubench8.hip && build/valu_ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { FMA, PKFMA, PKMUL, PKADD, EXP, RCP, MIX, SILU, SILU_PK, DSW64, DSW128, DSR128, MIXHI, MIXF32, CVTPK, CVTF32, CVTF16, NKIND };
static const char* kKindName[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32", "v_fma_mixlo_f16",
                                       "silu (fma exp add rcp mul) /5", "silu packed (pkfma 2exp pkadd 2rcp pkmul) /7", "ds_write_b64", "ds_write_b128", "ds_read_b128",
                                       "v_fma_mixhi_f16 (merging dest)", "v_fma_mix_f32 (f16 src0)", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f16_f32"};
static const int kInstrPerSlot[NKIND] = {1, 1, 1, 1, 1, 1, 1, 5, 7, 1, 1, 1, 1, 1, 1, 1, 1};

template <int K>
__global__ __launch_bounds__(512) void kb(int n, float* out, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) float lds[512 * 4 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x2 x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = (f32x2){0.001f * (lane + i), 0.002f * (lane + i)}; y[i] = x[i] * 0.5f; }
    const f32x2 c = {1.0001f, 0.9999f}, d = {0.0003f, 0.0001f};
    const float inv = 0.00048828125f;
    float* my = lds + tid * 4;
    for (int i = 0; i < 8; ++i) lds[tid + 512 * i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (K == FMA) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i][0]) : "v"(c[0]), "v"(d[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i][1]) : "v"(c[0]), "v"(d[0]));
            } else if constexpr (K == PKFMA) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(c), "v"(d));
            } else if constexpr (K == PKMUL) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(c));
            } else if constexpr (K == PKADD) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(d));
            } else if constexpr (K == EXP) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[i][0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[i][1]));
            } else if constexpr (K == RCP) {
                asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i][0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i][1]));
            } else if constexpr (K == MIX) {
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(y[i][0]) : "v"(x[i][0]), "v"(c[0]));
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(y[i][1]) : "v"(x[i][1]), "v"(c[0]));
            } else if constexpr (K == MIXHI) {
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(y[i][0]) : "v"(x[i][0]), "v"(c[0]));
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(y[i][1]) : "v"(x[i][1]), "v"(c[0]));
            } else if constexpr (K == MIXF32) {
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(y[i][0]) : "v"(x[i][0]), "v"(c[0]), "v"(x[i][1]));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(y[i][1]) : "v"(x[i][0]), "v"(c[0]), "v"(x[i][1]));
            } else if constexpr (K == CVTPK) {
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y[i][0]) : "v"(x[i][0]), "v"(x[i][1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y[i][1]) : "v"(x[i][1]), "v"(x[i][0]));
            } else if constexpr (K == CVTF32) {
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(y[i][0]) : "v"(x[i][0]));
                asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(y[i][1]) : "v"(x[i][1]));
            } else if constexpr (K == CVTF16) {
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(y[i][0]) : "v"(x[i][0]));
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(y[i][1]) : "v"(x[i][1]));
            } else if constexpr (K == SILU) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float t, u;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(y[i][h]), "v"(inv), "v"(x[i][h]));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(u) : "v"(t));
                    asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(u));
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(u));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i][h]) : "v"(t), "v"(u));
                }
            } else if constexpr (K == SILU_PK) {
                f32x2 t, u;
                const f32x2 inv2 = {inv, inv}, one2 = {1.f, 1.f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(y[i]), "v"(inv2), "v"(x[i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(u[0]) : "v"(t[0]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(u[1]) : "v"(t[1]));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(u) : "v"(one2));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(u[0]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(u[1]));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(u));
            } else if constexpr (K == DSW64) {
                asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(tid * 8)), "v"(x[i]) : "memory");
                asm volatile("ds_write_b64 %0, %1 offset:4096" ::"v"((uint32_t)(tid * 8)), "v"(y[i]) : "memory");
            } else if constexpr (K == DSW128) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 v = {x[i][0], x[i][1], y[i][0], y[i][1]};
                asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(tid * 16)), "v"(v) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"((uint32_t)(tid * 16)), "v"(v) : "memory");
            } else if constexpr (K == DSR128) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v, w;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(tid * 16)) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(w) : "v"((uint32_t)(tid * 16)) : "memory");
                if (i == 15) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x[0][0] += v[0] + w[1]; }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = my[0];
    for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1] + y[i][0] + y[i][1];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

// ---- the state-image phase of the edge kernel: 32 fp32 values per lane -> hi / lo' f16 images -> 16 ds_write_b64, in the compiler's own schedule ----
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int V>
__device__ __forceinline__ void split_pair(float x0, float x1, float pre, float neg, uint32_t& hiu, uint32_t& lou) {
    if constexpr (V == 0) {          // the kernel's form: four mixed-precision FMAs, two of them merging into a half-written register
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hiu) : "v"(x0), "s"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hiu) : "v"(x1), "s"(pre));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lou) : "v"(hiu), "s"(neg), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lou) : "v"(hiu), "s"(neg), "v"(x1));
    } else if constexpr (V == 1) {   // packed multiply, packed convert, two f32 mixed FMAs reading the f16 halves, packed convert
        f32x2 t; const f32x2 x = {x0, x1};
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "s"(f32x2{pre, pre}));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t[0]), "v"(t[1]));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    } else if constexpr (V == 2) {   // two plain multiplies instead of the packed one
        float t0, t1;
        asm("v_mul_f32 %0, %1, %2" : "=v"(t0) : "s"(pre), "v"(x0));
        asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "s"(pre), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t0), "v"(t1));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    } else if constexpr (V == 3) {   // hi as in the kernel, lo' through f32 + packed convert
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hiu) : "v"(x0), "s"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hiu) : "v"(x1), "s"(pre));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    }
}
template <int V>
__global__ __launch_bounds__(512) void ksplit(int n, float pre, float neg, float* out, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.37f * (lane + 1) + 0.011f * i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t h0, l0, h1, l1;
            split_pair<V>(x[4 * q], x[4 * q + 1], pre, neg, h0, l0);
            split_pair<V>(x[4 * q + 2], x[4 * q + 3], pre, neg, h1, l1);
            const int off = (q * 512 + tid) * 8;
            *(uint2*)(sm + off) = make_uint2(h0, h1);
            *(uint2*)(sm + 32768 + off) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] += 0.25f;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = ((float*)sm)[tid] + ((float*)sm)[8192 + tid];
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int V>
void run_split(const char* name, std::vector<float>* first) {
    const int blocks = 256, N = 500;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    double res[2];
    std::vector<float> o(512);
    for (int two = 0; two < 2; ++two) {
        const int threads = two ? 512 : 256, nw = threads / 64;
        hipLaunchKernelGGL((ksplit<V>), dim3(blocks), dim3(threads), 65536, 0, 3, 0.00048828125f, -2048.f, out, ticks);
        (void)hipDeviceSynchronize();
        if (two) (void)hipMemcpy(o.data(), out, 4 * 512, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL((ksplit<V>), dim3(blocks), dim3(threads), 65536, 0, N, 0.00048828125f, -2048.f, out, ticks);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * blocks);
        (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
        double a = 0;
        for (int i = 0; i < blocks; ++i) for (int w = 0; w < nw; ++w) a += h[i * 8 + w];
        res[two] = a / (nw * blocks) / N;
    }
    bool same = true;
    if (first->empty()) *first = o; else for (int i = 0; i < 512; ++i) same &= (__builtin_memcmp(&o[i], &(*first)[i], 4) == 0);
    printf("state images, 32 values/lane: %-58s 1 wave/SIMD %7.0f clk   2 waves/SIMD %7.0f clk   %s\n", name, res[0], res[1], same ? "bits = variant 0" : "BITS DIFFER");
    (void)hipFree(out); (void)hipFree(ticks);
}

template <int K>
void run() {
    const int blocks = 256, N = 500;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    double res[2];
    for (int two = 0; two < 2; ++two) {
        const int threads = two ? 512 : 256, nw = threads / 64;
        hipLaunchKernelGGL((kb<K>), dim3(blocks), dim3(threads), 0, 0, 10, out, ticks);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL((kb<K>), dim3(blocks), dim3(threads), 0, 0, N, out, ticks);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * blocks);
        (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
        double a = 0;
        for (int i = 0; i < blocks; ++i) for (int w = 0; w < nw; ++w) a += h[i * 8 + w];
        res[two] = a / (nw * blocks) / (32.0 * N * kInstrPerSlot[K]);
    }
    printf("%-48s  1 wave/SIMD %6.2f clk/instr   2 waves/SIMD %6.2f clk/instr per wave (= %5.2f per SIMD)\n", kKindName[K], res[0], res[1], res[1] / 2);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<FMA>(); run<PKFMA>(); run<PKMUL>(); run<PKADD>(); run<EXP>(); run<RCP>(); run<MIX>(); run<SILU>(); run<SILU_PK>();
    run<DSW64>(); run<DSW128>(); run<DSR128>();
    run<MIXHI>(); run<MIXF32>(); run<CVTPK>(); run<CVTF32>(); run<CVTF16>();
    std::vector<float> first;
    run_split<0>("4 x v_fma_mix{lo,hi}_f16 (kernel)", &first);
    run_split<1>("pk_mul, cvt_pk, 2 x fma_mix_f32, cvt_pk", &first);
    run_split<2>("2 x mul, cvt_pk, 2 x fma_mix_f32, cvt_pk", &first);
    run_split<3>("mixlo, mixhi, 2 x fma_mix_f32, cvt_pk", &first);
    return 0;
}
