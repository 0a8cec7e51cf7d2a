// Measures the issue cost of the vector instructions the compositing kernels are made of: cycles per
// wave64-instruction per SIMD at full occupancy (8 waves per SIMD, 8 independent chains per wave), derived from
// the run time at the clock the plain FMA sustains.  Used to decide which instruction mix is cheapest
// (packed vs plain FP32, DPP vs lane swaps, the price of exp / rcp).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITERS 2048

#define PROBE(NAME, ASM, EXTRA)                                                                            \
    __global__ __launch_bounds__(256) void NAME(float* out, float s)                                       \
    {                                                                                                      \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f, a4 = a0 + .4f,        \
              a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;                                                 \
        for (int i = 0; i < ITERS; i++) {                                                                  \
            asm volatile(ASM : "+v"(a0) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a1) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a2) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a3) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a4) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a5) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a6) : "v"(s) EXTRA);                                                   \
            asm volatile(ASM : "+v"(a7) : "v"(s) EXTRA);                                                   \
        }                                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
    }

PROBE(k_fma, "v_fma_f32 %0, %0, %1, 1.0", )
PROBE(k_mul, "v_mul_f32 %0, %0, %1", )
PROBE(k_add, "v_add_f32 %0, %0, %1", )
PROBE(k_exp, "v_exp_f32 %0, %0", )
PROBE(k_rcp, "v_rcp_f32 %0, %0", )
PROBE(k_rndne, "v_rndne_f32 %0, %0", )
PROBE(k_ldexp, "v_ldexp_f32 %0, %0, 1", )
PROBE(k_cvt, "v_cvt_i32_f32 %0, %0", )
PROBE(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", : "vcc")
PROBE(k_cmp, "v_cmp_lt_f32 vcc, %0, %1", : "vcc")
PROBE(k_dpp, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1", )
PROBE(k_mov, "v_mov_b32 %0, %1", )

// two-operand forms
__global__ __launch_bounds__(256) void k_swap32(float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(a2));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a3), "+v"(a4));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a5), "+v"(a6));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a7), "+v"(a0));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s;
}

__global__ __launch_bounds__(256) void k_swap16(float* out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(a2));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a3), "+v"(a4));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a5), "+v"(a6));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a7), "+v"(a0));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s;
}

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma(float* out, float s)
{
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2f sv = {s, s};
    for (int i = 0; i < ITERS; i++) {
#define PK(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(sv))
        PK(a0); PK(a1); PK(a2); PK(a3); PK(a4); PK(a5); PK(a6); PK(a7);
    }
    const v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}

template <typename K> static float time_kernel(K k, float* out, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, 0.999f);
    (void)hipEventRecord(e0);
    k<<<blocks, 256>>>(out, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const float base = time_kernel(k_fma, out, blocks);
    printf("v_fma_f32: %.3f ms for %d x 8 instructions per wave, 8 waves per SIMD\n", base, ITERS);
#define REPORT(name, k) printf("%-22s %.2f x the cost of v_fma_f32\n", name, time_kernel(k, out, blocks) / base)
    REPORT("v_mul_f32", k_mul);
    REPORT("v_add_f32", k_add);
    REPORT("v_pk_fma_f32", k_pkfma);
    REPORT("v_exp_f32", k_exp);
    REPORT("v_rcp_f32", k_rcp);
    REPORT("v_rndne_f32", k_rndne);
    REPORT("v_ldexp_f32", k_ldexp);
    REPORT("v_cvt_i32_f32", k_cvt);
    REPORT("v_cndmask_b32", k_cndmask);
    REPORT("v_cmp_lt_f32", k_cmp);
    REPORT("v_mov_b32", k_mov);
    REPORT("v_add_f32_dpp+s_nop1", k_dpp);
    REPORT("v_permlane32_swap", k_swap32);
    REPORT("v_permlane16_swap", k_swap16);
    return 0;
}
