// Quality metrics of the evaluator: SSIM, saicinpainting/evaluation/losses/ssim.py:46-71.
//
//   mu = W * x (depthwise window_size x window_size gaussian, sigma 1.5, ZERO padding window_size / 2: F.conv2d(padding=...)),
//   sigma_xx = W * x^2 - mu_x^2, sigma_xy = W * xy - mu_x mu_y,
//   ssim = (2 mu_x mu_y + C1)(2 sigma_xy + C2) / ((mu_x^2 + mu_y^2 + C1)(sigma_xx + sigma_yy + C2)),  C1 = 0.01^2, C2 = 0.03^2,
//   one value per image = mean over (C, H, W) (size_average=False: ssim_map.mean(1).mean(1).mean(1)).
//
// HBM-bound (two reads of each image, nothing written but one partial sum per tile): one workgroup = one 16 x 32 tile of one
// (image, channel) plane; the tile + halo of both images goes to LDS once, the window is applied separably (the reference's 2-D
// window IS the outer product of the normalised 1-D one) to the five maps x, y, x^2, y^2, xy -- horizontal pass into LDS, vertical
// pass in registers --, the tile's SSIM sum is reduced in the workgroup and written to a per-tile slot; a second launch adds the
// slots of an image in a fixed order (bit-reproducible, no atomics).
#include "common.h"

namespace {
constexpr int SS_TH = 16, SS_TW = 32, SS_MAXR = 7, SS_THREADS = 256;

struct SsimParams {
    const float* a; long long a_bs;
    const float* b; long long b_bs;
    float* partial;          // [B][C * tiles_y * tiles_x]
    int C, H, W, tiles_x, tiles_y, r;
    float g[2 * SS_MAXR + 1];
};

__global__ __launch_bounds__(SS_THREADS) void ssim_tile_kernel(SsimParams p) {
    // dynamic LDS (common.h: no static __shared__): [a tile + halo][b tile + halo][5 horizontally filtered maps][per-wave sums]
    constexpr int NPATCH = (SS_TH + 2 * SS_MAXR) * (SS_TW + 2 * SS_MAXR), NHZ = (SS_TH + 2 * SS_MAXR) * SS_TW;
    float* sa = reinterpret_cast<float*>(lama_smem);
    float* sb = sa + NPATCH;
    float (*hz)[NHZ] = reinterpret_cast<float (*)[NHZ]>(sb + NPATCH);
    float* red = sb + NPATCH + 5 * NHZ;
    const int r = p.r, PH = SS_TH + 2 * r, PW = SS_TW + 2 * r;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; t /= p.tiles_y;
    const int c = t % p.C, img = t / p.C;
    const float* pa = p.a + (long long)img * p.a_bs + (long long)c * p.H * p.W;
    const float* pb = p.b + (long long)img * p.b_bs + (long long)c * p.H * p.W;
    const int y0 = ty * SS_TH - r, x0 = tx * SS_TW - r;
    for (int i = threadIdx.x; i < PH * PW; i += SS_THREADS) {
        const int py = i / PW, px = i - py * PW;
        const int y = y0 + py, x = x0 + px;
        const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
        sa[i] = in ? pa[(long long)y * p.W + x] : 0.0f;
        sb[i] = in ? pb[(long long)y * p.W + x] : 0.0f;
    }
    __syncthreads();
    const int K = 2 * r + 1;
    for (int i = threadIdx.x; i < PH * SS_TW; i += SS_THREADS) {   // horizontal pass
        const int py = i / SS_TW, ox = i - py * SS_TW;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = p.g[k], u = sa[py * PW + ox + k], v = sb[py * PW + ox + k];
            s0 += w * u; s1 += w * v; s2 += w * (u * u); s3 += w * (v * v); s4 += w * (u * v);
        }
        hz[0][i] = s0; hz[1][i] = s1; hz[2][i] = s2; hz[3][i] = s3; hz[4][i] = s4;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int i = threadIdx.x; i < SS_TH * SS_TW; i += SS_THREADS) {   // vertical pass + the SSIM map
        const int oy = i / SS_TW, ox = i - oy * SS_TW;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = p.g[k];
            const int j = (oy + k) * SS_TW + ox;
            m1 += w * hz[0][j]; m2 += w * hz[1][j]; e11 += w * hz[2][j]; e22 += w * hz[3][j]; e12 += w * hz[4][j];
        }
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float v = ((2.0f * m12 + C1) * (2.0f * s12 + C2)) / ((m11 + m22 + C1) * (s11 + s22 + C2));
        if (ty * SS_TH + oy < p.H && tx * SS_TW + ox < p.W) acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < SS_THREADS / 64; ++w) s += red[w];
        p.partial[blockIdx.x] = s;
    }
}

// out[img] = (sum of the image's tile sums, pairwise in a fixed order) / (C * H * W)
__global__ __launch_bounds__(256) void ssim_reduce_kernel(const float* partial, float* out, int per_image, float inv_n) {
    double* red = reinterpret_cast<double*>(lama_smem);
    const float* src = partial + (long long)blockIdx.x * per_image;
    double s = 0.0;
    for (int i = threadIdx.x; i < per_image; i += 256) s += (double)src[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] * (double)inv_n);
}
}  // namespace

extern "C" size_t lama_ssim_workspace_bytes(int32_t batch, int32_t C, int32_t H, int32_t W) {
    if (batch <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)batch * C * lama_ceil_div(H, SS_TH) * lama_ceil_div(W, SS_TW) * sizeof(float);
}

extern "C" int lama_ssim_fwd(void* stream, const lama_tensor* img1, const lama_tensor* img2, int32_t batch, int32_t window_size,
                             const float* window1d, float* out, void* workspace, size_t workspace_bytes) {
    auto ok = [](const lama_tensor* t) {
        return t && t->ptr && t->dtype == LAMA_DT_F32 && t->C > 0 && t->H > 0 && t->W > 0 && t->batch_stride >= (int64_t)t->C * t->H * t->W;
    };
    if (!ok(img1) || !ok(img2) || batch <= 0 || !out || !window1d || img1->C != img2->C || img1->H != img2->H || img1->W != img2->W)
        return LAMA_ERR_BAD_ARG;
    if (window_size < 1 || window_size > 2 * SS_MAXR + 1 || (window_size & 1) == 0) return LAMA_ERR_UNSUPPORTED;   // odd windows up to 15
    if (!workspace || workspace_bytes < lama_ssim_workspace_bytes(batch, img1->C, img1->H, img1->W)) return LAMA_ERR_BAD_ARG;
    SsimParams p;
    memset(&p, 0, sizeof(p));
    p.a = (const float*)img1->ptr; p.a_bs = img1->batch_stride;
    p.b = (const float*)img2->ptr; p.b_bs = img2->batch_stride;
    p.partial = (float*)workspace;
    p.C = img1->C; p.H = img1->H; p.W = img1->W;
    p.tiles_x = lama_ceil_div(p.W, SS_TW); p.tiles_y = lama_ceil_div(p.H, SS_TH);
    p.r = window_size / 2;
    for (int i = 0; i < window_size; ++i) p.g[i] = window1d[i];   // HOST array: the normalised 1-D gaussian of ssim.py:36-40
    const int per_image = p.C * p.tiles_x * p.tiles_y;
    constexpr size_t tile_lds = ((size_t)2 * (SS_TH + 2 * SS_MAXR) * (SS_TW + 2 * SS_MAXR) + 5 * (SS_TH + 2 * SS_MAXR) * SS_TW + SS_THREADS / 64) * sizeof(float);
    hipLaunchKernelGGL(ssim_tile_kernel, dim3(batch * per_image), dim3(SS_THREADS), tile_lds, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ssim_reduce_kernel, dim3(batch), dim3(256), 256 * sizeof(double), (hipStream_t)stream, (const float*)workspace, out, per_image,
                       1.0f / ((float)p.C * (float)p.H * (float)p.W));
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}
