// What a plain streaming kernel reaches on this GPU, as the yardstick for the per-Gaussian kernels (which move
// 1-1.7 GB per launch at 5.2-6.0 TB/s): read-only sum, write-only fill and copy of 1.5 GB with 16-byte accesses,
// plain and non-temporal.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/stream_probe tools/probes/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(const v4* __restrict__ in, v4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (MODE == 1) { if (NT) __builtin_nontemporal_store(acc, out + i); else out[i] = acc; }
        if (MODE == 2) { const v4 v = NT ? __builtin_nontemporal_load(in + i) : in[i]; if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
    if (MODE == 0 && acc.x == 12345.f) out[0] = acc;
}

template <int MODE, bool NT>
static void run(const char* name, const v4* in, v4* out, size_t n, double bytes_per_elem)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int grid : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            k<MODE, NT><<<grid, 256>>>(in, out, n);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
    }
    printf("%-28s %.3f ms  %.2f TB/s\n", name, best, n * bytes_per_elem / best * 1e-9);
}

int main()
{
    const size_t n = (size_t)1500 * 1000 * 1000 / 16;
    v4 *in, *out;
    hipMalloc(&in, n * 16);
    hipMalloc(&out, n * 16);
    hipMemset(in, 0, n * 16);
    hipMemset(out, 0, n * 16);
    run<0, false>("read", in, out, n, 16);
    run<0, true>("read, non-temporal", in, out, n, 16);
    run<1, false>("write", in, out, n, 16);
    run<1, true>("write, non-temporal", in, out, n, 16);
    run<2, false>("copy (read + write)", in, out, n, 32);
    run<2, true>("copy, non-temporal", in, out, n, 32);
    return 0;
}
