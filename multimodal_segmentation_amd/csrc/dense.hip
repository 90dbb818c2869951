// keras Dense for tiny batches (rows <= 32) on gfx950: GEMV-shaped and bound by streaming the weight
// matrix once from HBM, so no MFMA (a 32-row MFMA tile would be >= 75 % padding).  Used by
// model_components/modality_encoder.py:45-50, decoder.py:37-40,68, layers/stn_spline.py:115-116,
// models/discriminator.py:32-33, model_components/balancer.py:24-25 of the reference.
#include "common.hpp"

#define DENSE_KSLICES_MAX 1024

static inline int dense_kslices(int K) {
    int ks = (K + 127) / 128;
    if (ks > DENSE_KSLICES_MAX) ks = DENSE_KSLICES_MAX;
    if (ks < 1) ks = 1;
    return ks;
}

// part[ks][b][n] = sum_{k in slice ks} x[b][k] * W[k][n]; block = 4 k-lanes x 64 n-lanes
template <int RB>
__global__ void dense_fwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                         int R, int K, int N, int kper) {
    __shared__ float red[4][RB][64];
    const int nl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
    float acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) acc[b] = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int k = kbeg + kl; k < kend; k += 4) {
            const float wv = w[(size_t)k * N + n];
#pragma unroll
            for (int b = 0; b < RB; ++b)
                if (b < R) acc[b] += x[(size_t)b * K + k] * wv;
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) red[kl][b][nl] = acc[b];
    __syncthreads();
    if (kl == 0 && n < N) {
#pragma unroll
        for (int b = 0; b < RB; ++b)
            if (b < R) part[((size_t)blockIdx.y * R + b) * N + n] = red[0][b][nl] + red[1][b][nl] + red[2][b][nl] + red[3][b][nl];
    }
}

// The same with 4 consecutive k per thread step (K % 4 == 0, kper % 4 == 0, x 16-byte aligned): the RB rows of x come as 16-byte loads,
// 20 memory instructions per 4 k instead of 68 -- the scalar form was bound by issuing them (25 us for a 26 MB weight matrix).
template <int RB>
__global__ __launch_bounds__(256) void dense_fwd_partial_k4_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                                   int R, int K, int N, int kper) {
    __shared__ float red[4][RB][64];
    const int nl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
    float acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) acc[b] = 0.f;
    if (n < N) {
#pragma unroll 2
        for (int k = kbeg + 4 * kl; k < kend; k += 16) {           // kend - kbeg and K are multiples of 4: k + 3 < kend
            float wv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[e] = w[(size_t)(k + e) * N + n];
#pragma unroll
            for (int b = 0; b < RB; ++b)
                if (b < R) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k);
                    acc[b] += xv[0] * wv[0]; acc[b] += xv[1] * wv[1]; acc[b] += xv[2] * wv[2]; acc[b] += xv[3] * wv[3];
                }
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) red[kl][b][nl] = acc[b];
    __syncthreads();
    if (kl == 0 && n < N) {
#pragma unroll
        for (int b = 0; b < RB; ++b)
            if (b < R) part[((size_t)blockIdx.y * R + b) * N + n] = red[0][b][nl] + red[1][b][nl] + red[2][b][nl] + red[3][b][nl];
    }
}

// N <= 4 (discriminator head, K = 373248): lanes run along K so that every lane is useful.  A block owns SMALLN_KPB
// consecutive k; with K % 4 == 0 and N == 1 every thread streams 16-byte pieces of the rows (the op is a batched dot product,
// bound by reading x once), otherwise 4-byte pieces.
#define SMALLN_KPB 1024
template <int RB, int NB>
__global__ __launch_bounds__(256) void dense_fwd_smalln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ part, int R, int K, int N) {
    __shared__ float red[17];
    float acc[RB][NB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[b][n] = 0.f;
    const int k0 = blockIdx.x * SMALLN_KPB;
    if (NB == 1 && (K & 3) == 0) {
        const int k = k0 + 4 * threadIdx.x;
        if (k < K) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                if (b < R) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k);
                    acc[b][0] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
                }
            }
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < SMALLN_KPB / 256; ++i) {
            const int k = k0 + i * 256 + threadIdx.x;
            if (k < K) {
                float wv[NB];
#pragma unroll
                for (int n = 0; n < NB; ++n) wv[n] = n < N ? w[(size_t)k * N + n] : 0.f;
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    if (b < R) {
                        const float xv = x[(size_t)b * K + k];
#pragma unroll
                        for (int n = 0; n < NB; ++n) acc[b][n] += xv * wv[n];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float t = block_sum(acc[b][n], red);
            if (threadIdx.x == 0 && b < R && n < N) part[((size_t)blockIdx.x * R + b) * N + n] = t;
        }
}

// y[i] = act(bias + sum_s part[s][i]); block 256 = 64 outputs x 4 slice lanes (fixed order -> deterministic)
__global__ __launch_bounds__(256) void dense_fwd_final_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              float* __restrict__ y, int R, int N, int ks, int act, float alpha) {
    __shared__ float sm[256];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float t = 0.f;
    if (i < R * N) {
#pragma unroll 4
        for (int s = lane; s < ks; s += 4) t += part[(size_t)s * R * N + i];
    }
    sm[threadIdx.x] = t;
    __syncthreads();
    if (lane == 0 && i < R * N) {
        t = sm[threadIdx.x] + sm[64 + threadIdx.x] + sm[128 + threadIdx.x] + sm[192 + threadIdx.x];
        if (bias) t += bias[i % N];
        y[i] = act_apply(t, act, alpha);
    }
}

// dx[b][k] = sum_n dy[b][n] * W[k][n]; 64 rows of W per block staged through LDS in N-chunks of 128
__global__ void dense_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                   int R, int K, int N) {
    __shared__ float Ws[64 * 129];
    __shared__ float Ds[32 * 128];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64;
    const int r = tid & 63, bg = tid >> 6;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int nb = 0; nb < N; nb += 128) {
        const int nw = min(128, N - nb);
        for (int i = tid; i < 64 * nw; i += 256) {
            const int rr = i / nw, nn = i - rr * nw;
            Ws[rr * 129 + nn] = (k0 + rr < K) ? w[(size_t)(k0 + rr) * N + nb + nn] : 0.f;
        }
        for (int i = tid; i < R * nw; i += 256) {
            const int bb = i / nw, nn = i - bb * nw;
            Ds[bb * 128 + nn] = dy[(size_t)bb * N + nb + nn];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = bg + 4 * i;
            if (b < R) {
                float a = 0.f;
                for (int nn = 0; nn < nw; ++nn) a += Ds[b * 128 + nn] * Ws[r * 129 + nn];
                acc[i] += a;
            }
        }
        __syncthreads();
    }
    if (k0 + r < K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = bg + 4 * i;
            if (b < R) dx[(size_t)b * K + k0 + r] = acc[i];
        }
    }
}

// dx[b][k] = sum_n dy[b][n] * W[k][n] for a dense layer with FEW inputs and many outputs (the SPADE decoder's Dense(8 -> H*W/8),
// decoder.py:68 of the reference): one block per row b, the threads stride over n (coalesced reads of dy[b][:] and of every row of
// W), K <= 16 accumulators per thread, fixed-order block sums.  The 64-rows-of-W-per-block kernel above is ONE block for K = 8 and
// walks N = 8192 in 64 synchronised chunks (87 us); this one is latency of a single pass.
template <int KMAX>
__global__ __launch_bounds__(256) void dense_dgrad_smallk_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int K, int N) {
    __shared__ float red[17];
    const int b = blockIdx.x;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float d = dy[(size_t)b * N + n];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) acc[k] = fmaf(d, w[(size_t)k * N + n], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float t = block_sum(acc[k], red);
            if (threadIdx.x == 0) dx[(size_t)b * K + k] = t;
        }
    }
}

// dW[k][n] = sum_b x[b][k] * dy[b][n]
__global__ void dense_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                   int R, int K, int N, int accumulate) {
    const long total = (long)K * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i / N, n = i - (long)k * N;
        float a = 0.f;
        for (int b = 0; b < R; ++b) a += x[(size_t)b * K + k] * dy[(size_t)b * N + n];
        dw[i] = accumulate ? dw[i] + a : a;
    }
}

extern "C" {

long mmseg_dense_workspace_floats(int R, int K, int N) {
    const long a = (long)dense_kslices(K) * R * N, b = (long)((K + SMALLN_KPB - 1) / SMALLN_KPB) * R * N;
    return a > b ? a : b;
}

int mmseg_dense_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int R, int K, int N, int act,
                    float alpha, void* stream) {
    if (R < 1 || R > 32) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 4 && (R <= 16 || N == 1) && K >= 4096) {
        const int nb = (K + SMALLN_KPB - 1) / SMALLN_KPB;
        if (R > 16) {
            hipLaunchKernelGGL((dense_fwd_smalln_kernel<32, 1>), dim3(nb), dim3(256), 0, st, x, w, ws, R, K, N);
        } else if (R <= 8) {
            if (N == 1) hipLaunchKernelGGL((dense_fwd_smalln_kernel<8, 1>), dim3(nb), dim3(256), 0, st, x, w, ws, R, K, N);
            else hipLaunchKernelGGL((dense_fwd_smalln_kernel<8, 4>), dim3(nb), dim3(256), 0, st, x, w, ws, R, K, N);
        } else {
            if (N == 1) hipLaunchKernelGGL((dense_fwd_smalln_kernel<16, 1>), dim3(nb), dim3(256), 0, st, x, w, ws, R, K, N);
            else hipLaunchKernelGGL((dense_fwd_smalln_kernel<16, 4>), dim3(nb), dim3(256), 0, st, x, w, ws, R, K, N);
        }
        hipLaunchKernelGGL(dense_fwd_final_kernel, dim3((R * N + 63) / 64), dim3(256), 0, st, (const float*)ws, bias, y, R, N, nb, act, alpha);
        return MMSEG_CHECK_LAUNCH();
    }
    const int ks = dense_kslices(K);
    int kper = (K + ks - 1) / ks;
    dim3 grid((N + 63) / 64, ks), block(256);
    if (K % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        kper = (kper + 3) / 4 * 4;                 // slices of whole k quads (the last slices may come out empty: they write zeros)
        if (R <= 8) hipLaunchKernelGGL(dense_fwd_partial_k4_kernel<8>, grid, block, 0, st, x, w, ws, R, K, N, kper);
        else if (R <= 16) hipLaunchKernelGGL(dense_fwd_partial_k4_kernel<16>, grid, block, 0, st, x, w, ws, R, K, N, kper);
        else hipLaunchKernelGGL(dense_fwd_partial_k4_kernel<32>, grid, block, 0, st, x, w, ws, R, K, N, kper);
        hipLaunchKernelGGL(dense_fwd_final_kernel, dim3((R * N + 63) / 64), dim3(256), 0, st, (const float*)ws, bias, y, R, N, ks, act, alpha);
        return MMSEG_CHECK_LAUNCH();
    }
    if (R <= 8) hipLaunchKernelGGL(dense_fwd_partial_kernel<8>, grid, block, 0, st, x, w, ws, R, K, N, kper);
    else if (R <= 16) hipLaunchKernelGGL(dense_fwd_partial_kernel<16>, grid, block, 0, st, x, w, ws, R, K, N, kper);
    else hipLaunchKernelGGL(dense_fwd_partial_kernel<32>, grid, block, 0, st, x, w, ws, R, K, N, kper);
    hipLaunchKernelGGL(dense_fwd_final_kernel, dim3((R * N + 63) / 64), dim3(256), 0, st, (const float*)ws, bias, y, R, N, ks, act, alpha);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_dense_dgrad(const float* dy, const float* w, float* dx, int R, int K, int N, void* stream) {
    if (R < 1 || R > 32) return (int)hipErrorInvalidValue;
    if (K <= 16 && N >= 1024) {
        hipLaunchKernelGGL(dense_dgrad_smallk_kernel<16>, dim3(R), dim3(256), 0, (hipStream_t)stream, dy, w, dx, K, N);
        return MMSEG_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3((K + 63) / 64), dim3(256), 0, (hipStream_t)stream, dy, w, dx, R, K, N);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_dense_wgrad(const float* x, const float* dy, float* dw, int R, int K, int N, int accumulate, void* stream) {
    if (R < 1 || R > 32) return (int)hipErrorInvalidValue;
    long blocks = ((long)K * N + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, dw, R, K, N, accumulate);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
