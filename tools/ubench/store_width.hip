// How much slower are 2-byte stores than 4- / 8- / 16-byte stores of the same bytes?  Every wave writes rows of 32 consecutive fp16 pixels per
// half wave (the accumulator layout of the MFMA kernels with fp16 outputs: LAMA_BUF_STORE_B16, 64 bytes per half-wave row) -- as 2-byte stores
// from all lanes, as 4-byte stores from the even lanes (two pixels merged through DPP), and fp32 rows as 4-byte stores for comparison.
// usage: store_width   (writes 512 MB per variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(char* y, long long rows_per_wave, int row_pitch_bytes) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0x40000000, 0x00020000);
    const int khalf = lane >> 5, l31 = lane & 31;
    for (long long i = 0; i < rows_per_wave; i += 2) {
        // two rows per iteration (khalf selects the row), 32 pixels each
        const long long row = (long long)wave * rows_per_wave + i + khalf;
        const unsigned v = (unsigned)(row + l31);
        if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)(row * row_pitch_bytes + l31 * 2), 0, 0);              // fp16, 2-byte stores
        else if (MODE == 1) {                                                                                                                  // fp16, even lanes store a dword
            const unsigned nb = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]: the neighbour's value
            const unsigned w = (v & 0xffffu) | (nb << 16);
            __builtin_amdgcn_raw_buffer_store_b32(w, r, (l31 & 1) ? 0x7ffffff0 : (int)(row * row_pitch_bytes + l31 * 2), 0, 0);
        } else if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)(row * row_pitch_bytes * 2 + l31 * 4), 0, 0);                   // fp32 rows, dword stores
    }
}
int main() {
    const long long total = 512ll << 20;
    char* y; hipMalloc(&y, (size_t)total * 2 + 4096);
    const int grid = 256 * 8, waves = grid * 4;
    for (int mode = 0; mode < 3; ++mode) {
        const int pitch = 64;                                     // bytes per fp16 row of 32 pixels (contiguous rows)
        const long long bytes = mode == 2 ? total * 2 : total;
        const long long rows = total / pitch, rpw = rows / waves;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto go = [&]() { if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, 0, y, rpw, pitch);
                          else if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, 0, y, rpw, pitch);
                          else hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(256), 0, 0, y, rpw, pitch); };
        go(); hipDeviceSynchronize();
        hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %lld MB in %.1f us = %.2f TB/s\n", mode == 0 ? "fp16 rows, 2-byte stores (all lanes)      " : mode == 1 ? "fp16 rows, 4-byte stores (even lanes, DPP) " : "fp32 rows, 4-byte stores (all lanes)       ",
               bytes >> 20, ms * 1e3, bytes / ms / 1e9);
    }
    return 0;
}
