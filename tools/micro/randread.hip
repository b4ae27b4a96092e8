// microbenchmark: bandwidth of reads in contiguous pieces of P bytes at random / sequential piece order
// hipcc --offload-arch=gfx950 -O3 -o randread randread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef float v4f __attribute__((ext_vector_type(4)));
// every wave processes pieces order[k], k = wave id, wave id + nwaves, ...; a piece = P bytes = P/1024 loads of 1 KB per wave
template <int LOADS>
__global__ __launch_bounds__(256) void rd(const v4f *__restrict__ buf, const unsigned *__restrict__ order, unsigned npieces, float *out) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (unsigned k = wave; k < npieces; k += nw) {
        const size_t base = (size_t)order[k] * (LOADS * 64);
        v4f r[LOADS];
#pragma unroll
        for (int u = 0; u < LOADS; ++u) r[u] = buf[base + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < LOADS; ++u) acc += r[u].w;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}
int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 206) << 20;
    v4f *buf; float *out; unsigned *dorder;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int loads : {1, 4, 16}) {
        const unsigned np = bytes / (loads * 1024);
        std::vector<unsigned> ord(np);
        for (unsigned i = 0; i < np; ++i) ord[i] = i;
        hipMalloc(&dorder, np * 4);
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 1) { std::mt19937 g(1); std::shuffle(ord.begin(), ord.end(), g); }
            if (mode == 2) {   // shuffled in groups of 16 consecutive pieces
                for (unsigned i = 0; i < np; ++i) ord[i] = i;
                std::vector<unsigned> grp(np / 16); for (unsigned i = 0; i < grp.size(); ++i) grp[i] = i;
                std::mt19937 g(2); std::shuffle(grp.begin(), grp.end(), g);
                for (unsigned i = 0; i < grp.size() * 16; ++i) ord[i] = grp[i / 16] * 16 + i % 16;
            }
            hipMemcpy(dorder, ord.data(), np * 4, hipMemcpyHostToDevice);
            for (int wgs : {512, 1024, 2048}) {
                float best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (loads == 1) rd<1><<<wgs, 256>>>(buf, dorder, np, out);
                    else if (loads == 4) rd<4><<<wgs, 256>>>(buf, dorder, np, out);
                    else rd<16><<<wgs, 256>>>(buf, dorder, np, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
                }
                printf("piece %5d B  order %-8s wgs %4d : %.3f ms  %.2f TB/s\n", loads * 1024,
                       mode == 0 ? "seq" : (mode == 1 ? "random" : "grp16"), wgs, best, bytes / best / 1e9);
            }
        }
        hipFree(dorder);
    }
    return 0;
}
