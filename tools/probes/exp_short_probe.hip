// Accuracy of shorter exponentials against the device library's expf over (-16, 0], the range whose results the
// compositing kernels use (alpha >= 1/255 needs power >= -5.6): error in units of the last place against the
// double-precision value, worst case and histogram of the deviation from expf.
//   variant 1: exp2(x * log2e) with a first-order correction for the rounding of the product (6 instructions)
//   variant 2: the same without the low word of log2e (5 instructions)
//   variant 3: exp2(x * log2e) alone (2 instructions)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/probes/bin/exp_short_probe tools/probes/exp_short_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

__device__ inline float exp_v(int v, float x)
{
    const float hi = __builtin_bit_cast(float, 0x3fb8aa3bu), lo = __builtin_bit_cast(float, 0x32a5705fu);
    const float ln2 = 0.69314718f;
    if (v == 0) return expf(x);
    const float ph = x * hi;
    const float e = __builtin_amdgcn_exp2f(ph);
    if (v == 3) return e;
    float d = fmaf(x, hi, -ph);
    if (v == 1) d = fmaf(x, lo, d);
    return fmaf(e * d, ln2, e);
}

__global__ void probe(unsigned n, float* worst /* [4] */, unsigned* differ /* [4][4]: by 0, 1, 2, more ulps from expf */)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = -16.0f * ((float)i / (float)n);
    const double t = exp((double)x);
    const float tf = (float)t;
    const double ulp = (double)(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, tf) + 1u) - tf);
    const float ref = exp_v(0, x);
    for (int v = 0; v < 4; v++) {
        const float r = exp_v(v, x);
        const float err = (float)(fabs((double)r - t) / ulp);
        if (err > worst[v]) atomicMax((int*)&worst[v], __builtin_bit_cast(int, err));
        const int dd = abs((int)(__builtin_bit_cast(unsigned, r) - __builtin_bit_cast(unsigned, ref)));
        atomicAdd(&differ[v * 4 + (dd > 3 ? 3 : dd)], 1u);
    }
}

int main()
{
    float* w;
    unsigned* d;
    hipMalloc(&w, 16);
    hipMalloc(&d, 64);
    hipMemset(w, 0, 16);
    hipMemset(d, 0, 64);
    const unsigned n = 1u << 26;
    probe<<<n / 256, 256>>>(n, w, d);
    float hw[4];
    unsigned hd[16];
    hipMemcpy(hw, w, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hd, d, 64, hipMemcpyDeviceToHost);
    const char* names[4] = {"expf", "corrected, full log2e", "corrected, high word only", "exp2 of the product"};
    for (int v = 0; v < 4; v++)
        printf("%-28s worst error %.2f ulp; differs from expf by 0/1/2/more ulp: %u / %u / %u / %u\n", names[v], hw[v],
               hd[v * 4], hd[v * 4 + 1], hd[v * 4 + 2], hd[v * 4 + 3]);
    return 0;
}
