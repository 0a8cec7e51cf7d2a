// Checks the 9-value wave reduce-scatter used by the backward compositing kernel against a host sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__device__ __forceinline__ float fold32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    unsigned x = r[0], y = r[1];
    // hipcc (ROCm 7.2) folds "r[0] + r[1]" of a lane swap into "r[0] + r[0]"; an opaque use of the second result
    // keeps the two registers apart.
    asm volatile("" : "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float fold16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    unsigned x = r[0], y = r[1];
    // hipcc (ROCm 7.2) folds "r[0] + r[1]" of a lane swap into "r[0] + r[0]"; an opaque use of the second result
    // keeps the two registers apart.
    asm volatile("" : "+v"(y));
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float row_sum_lane15(float v)
{
#define NGS_ROW_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    NGS_ROW_ADD(0x111); NGS_ROW_ADD(0x112); NGS_ROW_ADD(0x114); NGS_ROW_ADD(0x118);
    return v;
}
__global__ void k(const float* in, float* out)
{
    __shared__ float s_acc[16];
    const int lane = threadIdx.x;
    if (lane < 16) s_acc[lane] = 0.f;
    __syncthreads();
    float v[9];
    for (int i = 0; i < 9; i++) v[i] = in[i * 64 + lane];
    const int lrow = lane >> 4;
    const int slot0 = lrow == 0 ? 0 : lrow == 1 ? 3 : lrow == 2 ? 5 : 8;
    const int slot1 = lrow == 0 ? 1 : lrow == 1 ? 4 : lrow == 2 ? 6 : -1;
    const int slot2 = lrow == 0 ? 2 : lrow == 2 ? 7 : -1;
    const float r0 = fold32(v[0], v[5]), r1 = fold32(v[1], v[6]), r2 = fold32(v[2], v[7]), r3 = fold32(v[3], v[8]), r4 = fold32(v[4], 0.0f);
    float s0 = fold16(r0, r3), s1 = fold16(r1, r4), s2 = fold16(r2, 0.0f);
    s0 = row_sum_lane15(s0); s1 = row_sum_lane15(s1); s2 = row_sum_lane15(s2);
    if ((lane & 15) == 15) {
        atomicAdd(s_acc + slot0, s0);
        if (slot1 >= 0) atomicAdd(s_acc + slot1, s1);
        if (slot2 >= 0) atomicAdd(s_acc + slot2, s2);
    }
    __syncthreads();
    if (lane < 9) out[lane] = s_acc[lane];
}
int main()
{
    float h[9 * 64], ref[9] = {0};
    for (int i = 0; i < 9; i++) for (int l = 0; l < 64; l++) { h[i * 64 + l] = (float)((i + 1) * 100 + l) * ((l % 3) - 1); ref[i] += h[i * 64 + l]; }
    float *din, *dout; hipMalloc(&din, sizeof h); hipMalloc(&dout, 9 * 4);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    float o[9]; hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 9; i++) { printf("v%d: got %g want %g\n", i, o[i], ref[i]); if (fabsf(o[i] - ref[i]) > 1e-3f * fabsf(ref[i]) + 1e-3f) bad++; }
    printf(bad ? "MISMATCH\n" : "OK\n");
    return bad;
}
