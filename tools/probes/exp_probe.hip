// Checks on the GPU that exp_nonpos() (render_common.h) returns the same bits as expf() for arguments in
// (-103, 0], which is every argument whose result the compositing kernels use.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I neuralgaussiansplatting_amd/csrc -o tools/probes/bin/exp_probe tools/probes/exp_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "render_common.h"

__global__ void probe(unsigned n, unsigned* mismatches, float* worst)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // dense sweep of (-103, 0] plus the exact float grid near zero
    const float x = (i & 1u) ? -103.0f * ((float)(i >> 1) / (float)(n >> 1)) : -__builtin_bit_cast(float, 0x3f800000u - ((i >> 1) & 0xffffffu) * 63u);
    const float a = expf(x), b = exp_nonpos(x);
    if (__builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b)) {
        atomicAdd(mismatches, 1u);
        *worst = x;
    }
}

int main()
{
    unsigned* d;
    float* w;
    hipMalloc(&d, 4);
    hipMalloc(&w, 4);
    hipMemset(d, 0, 4);
    hipMemset(w, 0, 4);
    const unsigned n = 1u << 26;
    probe<<<n / 256, 256>>>(n, d, w);
    unsigned h = 0;
    float hw = 0;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&hw, w, 4, hipMemcpyDeviceToHost);
    printf("exp_nonpos vs expf over %u arguments: %u mismatches (last at x=%g)\n", n, h, hw);
    return h != 0;
}
