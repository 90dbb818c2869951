// Batch assembly + rotation augmentation on the device (gfx950).
//
// The reference augments on the host, one slice at a time: keras ImageDataGenerator(rotation_range=20).flow(...)
// (model_executors/base_executor.py:37-78,103-110) rotates every sample of a batch about the image centre with
// scipy.ndimage.affine_transform(order=1, mode='nearest') per channel.  Here the whole training set stays resident in
// HBM and one launch gathers the B slices of the batch (rows[]) and resamples them with a per-sample 2x3 matrix:
//     src(r, c) = (m0*r + m1*c + m2,  m3*r + m4*c + m5)      coordinates clamped to the image ('nearest' extension),
//     out[b, r, c, :] = bilinear(data[rows[b]], src(r, c))         (order 1; order 0 = the tap at floor(src + 0.5))
// HBM-bound: one read of ~B slices + one write.  One thread per output element; consecutive threads walk the
// channel-fastest NHWC order, so stores are fully coalesced and the 4 taps of neighbouring pixels share cache lines.
#include "common.hpp"

__global__ void affine_gather_kernel(const float* __restrict__ data, const int* __restrict__ rows, const float* __restrict__ mat,
                                     float* __restrict__ out, int H, int W, int C, long per_sample, int order) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= per_sample) return;
    const int ch = (int)(e % C);
    const int p = (int)(e / C);
    const int r = p / W, c = p - r * W;
    const float* m = mat + b * 6;
    float sr = m[0] * (float)r + m[1] * (float)c + m[2];
    float sc = m[3] * (float)r + m[4] * (float)c + m[5];
    sr = fminf(fmaxf(sr, 0.f), (float)(H - 1));
    sc = fminf(fmaxf(sc, 0.f), (float)(W - 1));
    if (order == 0) { sr = floorf(sr + 0.5f); sc = floorf(sc + 0.5f); }
    const int r0 = min((int)sr, H - 1), c0 = min((int)sc, W - 1);
    const int r1 = min(r0 + 1, H - 1), c1 = min(c0 + 1, W - 1);
    const float ar = sr - (float)r0, ac = sc - (float)c0;
    const float* src = data + (size_t)(rows ? rows[b] : b) * per_sample + ch;
    const float v00 = src[((size_t)r0 * W + c0) * C], v01 = src[((size_t)r0 * W + c1) * C];
    const float v10 = src[((size_t)r1 * W + c0) * C], v11 = src[((size_t)r1 * W + c1) * C];
    out[(size_t)b * per_sample + e] = (1.f - ar) * ((1.f - ac) * v00 + ac * v01) + ar * ((1.f - ac) * v10 + ac * v11);
}

extern "C" {

// data [N,H,W,C] (N > max(rows)), rows [B] int32 or nullptr (= identity), mat [B,6], out [B,H,W,C]
int mmseg_affine_gather(const float* data, const int* rows, const float* mat, float* out, int B, int H, int W, int C, int order,
                        void* stream) {
    if (B <= 0) return 0;
    if (H < 1 || W < 1 || C < 1 || B > 65535 || order < 0 || order > 1) return (int)hipErrorInvalidValue;
    const long per = (long)H * W * C;
    dim3 grid((unsigned)((per + 255) / 256), B);
    hipLaunchKernelGGL(affine_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, data, rows, mat, out, H, W, C, per, order);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
