// Per-CU fill rate of L2-resident data: every workgroup (one per CU) streams the SAME `wbytes` buffer (packed weights of one layer:
// 0.15 .. 4 MB) `reps` times with 16-byte buffer loads, `depth` loads in flight per wave.  Question behind it: the
// weights-in-registers convolutions draw 2.0 - 2.4 MB of A fragments per CU and launch from L2 and take 95 - 97 us = 21 - 25 GB/s per CU --
// is that the rate a CU can pull from its XCD's L2 at all?   usage: l2_stream [wbytes_KB] [waves] [depth] [stride_cus]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ w, long long n16, int reps, unsigned* out, int distinct) {
    const int nthr = blockDim.x;
    // distinct = 1: every workgroup reads its own copy (no sharing in L2: the data comes from the Infinity Cache / HBM)
    const u32x4* base = w + (distinct ? (long long)blockIdx.x * n16 : 0);
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (long long i = threadIdx.x; i < n16; i += (long long)nthr * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { long long j = i + (long long)d * nthr; v[d] = j < n16 ? __builtin_nontemporal_load(base + j) : acc; }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel_plain(const u32x4* __restrict__ w, long long n16, int reps, unsigned* out, int distinct) {
    const int nthr = blockDim.x;
    const u32x4* base = w + (distinct ? (long long)blockIdx.x * n16 : 0);
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (long long i = threadIdx.x; i < n16; i += (long long)nthr * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { long long j = i + (long long)d * nthr; v[d] = j < n16 ? base[j] : acc; }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
int main(int argc, char** argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned* out; hipMalloc(&out, 4);
    const int kbs[] = {147, 1024, 2048, 4096};
    printf("CUs %d\n", ncu);
    for (int distinct = 0; distinct < 2; ++distinct)
    for (int kb : kbs) {
        const long long n16 = (long long)kb * 1024 / 16;
        u32x4* w; hipMalloc(&w, (size_t)n16 * 16 * (distinct ? ncu : 1)); hipMemset(w, 1, (size_t)n16 * 16 * (distinct ? ncu : 1));
        for (int waves : {4, 8, 16}) {
            for (int plain = 0; plain < 2; ++plain) {
                const int reps = distinct ? 4 : 16;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto go = [&]() { if (plain) hipLaunchKernelGGL(stream_kernel_plain<8>, dim3(ncu), dim3(64 * waves), 0, 0, w, n16, reps, out, distinct);
                                  else hipLaunchKernelGGL(stream_kernel<8>, dim3(ncu), dim3(64 * waves), 0, 0, w, n16, reps, out, distinct); };
                go(); hipDeviceSynchronize();
                hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes_per_cu = (double)n16 * 16 * reps;
                printf("%s %4d KB per CU x %2d reps, %2d waves, %s loads: %7.1f us -> %6.1f GB/s per CU (%5.1f B/clk at 2.1 GHz), chip %5.2f TB/s\n",
                       distinct ? "own copy " : "shared   ", kb, reps, waves, plain ? "plain" : "nt   ", ms * 1e3, bytes_per_cu / ms / 1e6, bytes_per_cu / ms / 1e6 / 2.1,
                       bytes_per_cu * ncu / ms / 1e9);
            }
        }
        hipFree(w);
    }
    return 0;
}
