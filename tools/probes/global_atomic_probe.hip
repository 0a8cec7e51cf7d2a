// Throughput of device-wide integer atomics on a small table of counters, as a one-pass binning of the instances by
// tile would issue them: every Gaussian touches the counters of the tiles in its rectangle (2 x 2 here), 3 M Gaussians,
// 8160 tiles of a 120 x 68 grid.
//   mode 0: atomicAdd without a use of the result (counting), counters packed
//   mode 1: the same, one counter per 64-byte line
//   mode 2: atomicAdd whose result addresses an 8-byte store into the tile's segment (cursor scatter), counters packed
//   mode 3: as 2, one counter per 64-byte line
//   mode 4: counting into a table of blocks x 64 digits (per-block histograms of a later radix pass), addressed by a
//           pseudo-random destination
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/global_atomic_probe tools/probes/global_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define TX 120
#define TY 68

__device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* counters, uint2* out, int P, int per_tile_cap, int table)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P) return;
    const uint32_t h = mix(g);
    const int x0 = h % (TX - 1), y0 = (h >> 12) % (TY - 1);
    for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
            const int tile = (y0 + dy) * TX + x0 + dx;
            if (MODE == 0) atomicAdd(&counters[tile], 1u);
            if (MODE == 1) atomicAdd(&counters[tile * 16], 1u);
            if (MODE == 2 || MODE == 3) {
                const uint32_t slot = atomicAdd(&counters[MODE == 2 ? tile : tile * 16], 1u);
                if (slot < (uint32_t)per_tile_cap) out[(size_t)tile * per_tile_cap + slot] = make_uint2(g, slot);
            }
            if (MODE == 4) atomicAdd(&counters[mix(g * 4 + dy * 2 + dx) % (uint32_t)table], 1u);
        }
}

template <int MODE>
static void run(const char* name)
{
    const int P = 3000000, cap = 2048, table = 2913 * 64;
    uint32_t* counters;
    uint2* out;
    hipMalloc(&counters, (size_t)table * 4 > 8160 * 64 ? (size_t)table * 4 : 8160 * 64);
    hipMalloc(&out, (size_t)8160 * cap * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(counters, 0, (size_t)table * 4 > 8160 * 64 ? (size_t)table * 4 : 8160 * 64);
        hipDeviceSynchronize();
        hipEventRecord(a);
        probe<MODE><<<(P + 255) / 256, 256>>>(counters, out, P, cap, table);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("%-44s %.4f ms for %d atomics (%.1f G/s)\n", name, best, P * 4, P * 4 / best * 1e-6);
    hipFree(counters);
    hipFree(out);
}

int main()
{
    run<0>("count, packed counters");
    run<1>("count, one counter per line");
    run<2>("cursor + 8-byte store, packed counters");
    run<3>("cursor + 8-byte store, one counter per line");
    run<4>("count into 2913 x 64 table");
    return 0;
}
