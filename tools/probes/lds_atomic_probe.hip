// Cost of LDS float atomics against plain LDS stores and reads, per wave-instruction, as a function of the number of
// active lanes and of how their addresses fall: the compositing backward kernel adds 36 lanes' sums per step to a
// table in LDS.  One workgroup of 256 threads per CU, every wave issuing the operation back to back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/lds_atomic_probe tools/probes/lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096

// mode 0: ds_add_f32, 1: ds_add_u32, 2: ds_write_b32, 3: ds_read_b32 + add + ds_write_b32
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int active_per_row, int stride_dw, int spread)
{
    __shared__ float s_tab[256 * 12];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4, li = lane & 15;
    for (int i = tid; i < 256 * 12; i += 256) s_tab[i] = 0.f;
    __syncthreads();
    // lanes li < active_per_row of each 16-lane row are active; row r works on table row (r * spread + i) & 255
    const bool on = li < active_per_row;
    float acc = 0.f;
    for (int i = 0; i < ITERS; i++) {
        const int j = (row * spread + i * 7 + (tid >> 6) * 61) & 255;
        float* p = &s_tab[j * stride_dw + li];
        if (on) {
            if (MODE == 0) atomicAdd(p, 1.0f);
            if (MODE == 1) atomicAdd((unsigned*)p, 1u);
            if (MODE == 2) *(volatile float*)p = (float)i;
            if (MODE == 3) { const float v = *(volatile float*)p; *(volatile float*)p = v + 1.0f; }
        }
    }
    __syncthreads();
    for (int i = tid; i < 256 * 12; i += 256) acc += s_tab[i];
    out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static void run(const char* name, int active, int stride, int spread)
{
    float* out;
    hipMalloc(&out, 256 * 256 * 4 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<MODE><<<256, 256>>>(out, active, stride, spread);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<256, 256>>>(out, active, stride, spread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // 4 waves per CU issue ITERS instructions each; cycles at 2.4 GHz per wave-instruction per CU
    printf("%-28s active/row %2d stride %2d spread %3d: %.3f ms -> %.1f cycles per wave-instruction (CU-wide)\n", name, active,
           stride, spread, ms, ms * 1e-3 * 2.4e9 / (4.0 * ITERS));
    hipFree(out);
}

int main()
{
    for (int active : {1, 4, 9, 16}) {
        run<0>("ds_add_f32", active, 9, 1);
        run<1>("ds_add_u32", active, 9, 1);
        run<2>("ds_write_b32", active, 9, 1);
        run<3>("read+add+write", active, 9, 1);
    }
    run<0>("ds_add_f32 same row", 9, 9, 0);
    run<0>("ds_add_f32 stride 12", 9, 12, 1);
    run<0>("ds_add_f32 stride 16", 9, 16, 1);
    run<0>("ds_add_f32 spread 37", 9, 9, 37);
    return 0;
}
