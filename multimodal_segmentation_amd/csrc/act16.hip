// Reduced-precision STORAGE of the convolutional trunk (build-defined extension for BASELINE configs #3 / #5): BatchNorm, 2x2 max
// pooling and the gradient of nearest x2 up-sampling with every activation / gradient tensor stored either as fp32 or as a 16-bit type
// (h code 0 = fp32, 1 = bf16, 2 = fp16), chosen per tensor.  Arithmetic, statistics, gamma / beta, moving averages and the
// per-channel reductions are fp32 exactly as in norm.hip / pointwise.hip (same two-stage, fixed-order reductions -> run-to-run
// bitwise reproducible); only the loads and stores differ: 8-byte accesses of four 16-bit elements instead of 16-byte accesses of four
// floats, i.e. half the HBM bytes of these HBM-bound passes.  The fp32 entry points of norm.hip / pointwise.hip are untouched.
// Reference ops: keras BatchNormalization / MaxPooling2D / UpSampling2D as used in models/unet.py:37-101,
// model_components/segmentor.py:16-21.
#include "common.hpp"

// ---- BatchNorm ------------------------------------------------------------------------------------------------------------------
// C % 64 == 0 (every BatchNorm of the trunk).  grid (row blocks, C/64), block = 16 float4 column lanes x 16 row lanes.
// MODE 0: (x - shift, (x - shift)^2), shift = x[0][c]; MODE 1: (g, g * xhat), g = dy * [y > 0 if relu]
template <int MODE, int HX, int HY>
__global__ __launch_bounds__(256) void bn16_partial_kernel(const void* __restrict__ x, const void* __restrict__ dy, const void* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ part, long M, int C, long rows_per_block, int relu) {
    __shared__ f32x4 sm[2][16][16];
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int C4 = C >> 2;
    const int col = blockIdx.y * 16 + c4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        const f32x4 sh = ld4<HX>(x, col);
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += 16) {
            const f32x4 d = ld4<HX>(x, r * C4 + col) - sh;
            s1 += d; s2 += d * d;
        }
    } else {
        const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[col], is = reinterpret_cast<const f32x4*>(invstd)[col];
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += 16) {
            f32x4 g = ld4<HY>(dy, r * C4 + col);
            if (relu) {
                const f32x4 o = ld4<HY>(y, r * C4 + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
            }
            s1 += g; s2 += g * (ld4<HX>(x, r * C4 + col) - mu) * is;
        }
    }
    sm[0][rl][c4] = s1; sm[1][rl][c4] = s2;
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, cc = tid & 15;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[which][k][cc];
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + which) * C + (size_t)(blockIdx.y * 16 + cc) * 4) = t;
    }
}

template <int HX>
__global__ __launch_bounds__(FIN_NT) void bn16_stats_final_kernel(const float* __restrict__ part, const void* __restrict__ x, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd,
                                        float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mov_mean,
                                        float* __restrict__ mov_var, int nblk, int C, long M, float eps, float momentum) {
    __shared__ float sm[2 * FIN_NT];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s1, s2;
    reduce_partials_64(part, nblk, C, c, lane, s1, s2, sm);
    if (c >= C || lane != 0) return;
    const float invM = 1.f / (float)M;
    const float d = s1 * invM;
    const float mu = ld1<HX>(x, c) + d;
    float var = s2 * invM - d * d;
    var = var > 0.f ? var : 0.f;
    const float is = rsqrtf(var + eps);
    mean[c] = mu; invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - mu * sc;
    if (mov_mean) {
        const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        mov_mean[c] -= (mov_mean[c] - mu) * (1.f - momentum);
        mov_var[c] -= (mov_var[c] - unb) * (1.f - momentum);
    }
}

template <int HX, int HY>
__global__ void bn16_apply_kernel(const void* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  void* __restrict__ y, long n4, int C4, int relu) {
    const f32x4* S = reinterpret_cast<const f32x4*>(scale);
    const f32x4* T = reinterpret_cast<const f32x4*>(shift);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        f32x4 v = ld4<HX>(x, i) * S[c] + T[c];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        st4<HY>(y, i, v);
    }
}

__global__ __launch_bounds__(FIN_NT) void bn16_bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ mean,
                                      const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                      float* __restrict__ coef /* [3][C] */, int nblk, int C, long M, int accumulate) {
    __shared__ float sm[2 * FIN_NT];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s1, s2;
    reduce_partials_64(part, nblk, C, c, lane, s1, s2, sm);
    if (c >= C || lane != 0) return;
    if (dgamma) {
        if (accumulate) { dbeta[c] += s1; dgamma[c] += s2; }
        else { dbeta[c] = s1; dgamma[c] = s2; }
    }
    const float invM = 1.f / (float)M, is = invstd[c], ga = gamma[c], mu = mean[c];
    const float A = ga * is;
    const float Bc = -ga * is * is * s2 * invM;
    coef[c] = A; coef[C + c] = Bc; coef[2 * C + c] = -A * s1 * invM - Bc * mu;
}

template <int HX, int HY>
__global__ void bn16_bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ y, const void* __restrict__ x,
                                      const float* __restrict__ coef, void* __restrict__ dx, long n4, int C4, int relu) {
    const f32x4* A = reinterpret_cast<const f32x4*>(coef);
    const f32x4* Bc = A + C4;
    const f32x4* Cc = A + 2 * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        f32x4 g = ld4<HY>(dy, i);
        if (relu) {
            const f32x4 o = ld4<HY>(y, i);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
        st4<HX>(dx, i, A[c] * g + Bc[c] * ld4<HX>(x, i) + Cc[c]);
    }
}

// ---- 2x2 max pooling / gradient of nearest x2 up-sampling -------------------------------------------------------------------------
template <int H>
__global__ void maxpool16_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int Hh, int W, int C4) {
    const int Ho = Hh / 2, Wo = W / 2;
    const long n = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int wo = r % Wo; r /= Wo;
        const int ho = r % Ho; const int b = r / Ho;
        const long base = (((long)b * Hh + 2 * ho) * W + 2 * wo) * C4 + c;
        const f32x4 v00 = ld4<H>(x, base), v01 = ld4<H>(x, base + C4), v10 = ld4<H>(x, base + (long)W * C4),
                    v11 = ld4<H>(x, base + (long)W * C4 + C4);
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
        st4<H>(y, i, m);
    }
}
// gradient goes to the FIRST maximum in row-major window order (as in pointwise.hip)
template <int H>
__global__ void maxpool16_bwd_kernel(const void* __restrict__ x, const void* __restrict__ y, const void* __restrict__ dy,
                                     void* __restrict__ dx, int B, int Hh, int W, int C4, const void* __restrict__ add = nullptr) {
    const int Ho = Hh / 2, Wo = W / 2;
    const long n = (long)B * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int wo = r % Wo; r /= Wo;
        const int ho = r % Ho; const int b = r / Ho;
        const long base = (((long)b * Hh + 2 * ho) * W + 2 * wo) * C4 + c;
        const long o01 = C4, o10 = (long)W * C4, o11 = (long)W * C4 + C4;
        const f32x4 v00 = ld4<H>(x, base), v01 = ld4<H>(x, base + o01), v10 = ld4<H>(x, base + o10), v11 = ld4<H>(x, base + o11);
        const f32x4 m = ld4<H>(y, i), g = ld4<H>(dy, i);
        f32x4 g00, g01, g10, g11;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = v00[e] == m[e];
            const bool bb = !a && v01[e] == m[e];
            const bool cc = !a && !bb && v10[e] == m[e];
            const bool dd = !a && !bb && !cc;
            g00[e] = a ? g[e] : 0.f; g01[e] = bb ? g[e] : 0.f; g10[e] = cc ? g[e] : 0.f; g11[e] = dd ? g[e] : 0.f;
        }
        if (add != nullptr) {      // the tensor has a second consumer (UNet skip): its gradient is added in the same pass
            g00 += ld4<H>(add, base); g01 += ld4<H>(add, base + o01); g10 += ld4<H>(add, base + o10); g11 += ld4<H>(add, base + o11);
        }
        st4<H>(dx, base, g00); st4<H>(dx, base + o01, g01); st4<H>(dx, base + o10, g10); st4<H>(dx, base + o11, g11);
    }
}
template <int H>
__global__ void upsample16_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int B, int Hh, int W, int C4) {
    const long n = (long)B * Hh * W * C4;      // Hh, W: low-res dims; dy is [B, 2H, 2W, C]
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int w = r % W; r /= W;
        const int h = r % Hh; const int b = r / Hh;
        const long base = (((long)b * 2 * Hh + 2 * h) * 2 * W + 2 * w) * C4 + c;
        st4<H>(dx, i, ld4<H>(dy, base) + ld4<H>(dy, base + C4) + ld4<H>(dy, base + 2L * W * C4) + ld4<H>(dy, base + 2L * W * C4 + C4));
    }
}

static inline int rows16(long M, int C) {
    long nb = 1024 / (C / 64);
    if (nb > 512) nb = 512;
    const long maxb = (M + 63) / 64;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    return (int)nb;
}
static inline int grid16(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- activation gradient (+ bias gradient) --------------------------------------------------------------------------------------
// dx = dy * act'(y) of a convolution's fused activation (y = its OUTPUT), all three tensors stored with element code H; with
// `part` also the per-block column sums of dx -- the bias gradient of that convolution -- in the same pass (one read of dy and y,
// one write of dx, instead of act_bwd followed by a column-sum pass over dx).  C % 64 == 0; grid (row blocks, C/64), block = 16
// float4 column lanes x 16 row lanes; deterministic two-stage reduction as mmseg_colsum.
template <int H, bool SUM>
__global__ __launch_bounds__(256) void act16_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ y, void* __restrict__ dx,
                                                        float* __restrict__ part, long M, int C, long rows_per_block, int act, float alpha) {
    __shared__ f32x4 sm[16][16];
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int C4 = C >> 2, col = blockIdx.y * 16 + c4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += 16) {
        const f32x4 g = ld4<H>(dy, r * C4 + col), o = ld4<H>(y, r * C4 + col);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = g[e] * act_grad_from_out(o[e], act, alpha);
        st4<H>(dx, r * C4 + col, d);
        if (SUM) {
            if (H != 0) {                      // the bias gradient sums what the weight / data gradients will read: the stored values
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = H == 1 ? (float)(__bf16)d[e] : (float)(_Float16)d[e];
            }
            s += d;
        }
    }
    if (SUM) {
        sm[rl][c4] = s;
        __syncthreads();
        if (tid < 16) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) t += sm[k][tid];
            *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (size_t)(blockIdx.y * 16 + tid) * 4) = t;
        }
    }
}
// part[blockIdx.x][c] = sum over this block's rows of x[., c] for a tensor stored with element code H (C % 64 == 0; grid (row blocks,
// C / 64), block = 16 float4 column lanes x 16 row lanes): the bias gradient of a convolution whose output gradient is a 16-bit tensor
template <int H>
__global__ __launch_bounds__(256) void colsum16_partial_kernel(const void* __restrict__ x, float* __restrict__ part, long M, int C,
                                                               long rows_per_block) {
    __shared__ f32x4 sm[16][16];
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int C4 = C >> 2, col = blockIdx.y * 16 + c4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += 16) s += ld4<H>(x, r * C4 + col);
    sm[rl][c4] = s;
    __syncthreads();
    if (tid < 16) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][tid];
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (size_t)(blockIdx.y * 16 + tid) * 4) = t;
    }
}
// out[c] (+)= sum over the row blocks of part[b][c]; one block per channel, fixed order
__global__ __launch_bounds__(256) void colsum16_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C,
                                                             int accumulate) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float a = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) a += part[(size_t)b * C + c];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + red[0] : red[0];
}
// element-wise form for any n (n % 4 == 0): no bias gradient
template <int H>
__global__ void act16_bwd_flat_kernel(const void* __restrict__ dy, const void* __restrict__ y, void* __restrict__ dx, long n4, int act, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 g = ld4<H>(dy, i), o = ld4<H>(y, i);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = g[e] * act_grad_from_out(o[e], act, alpha);
        st4<H>(dx, i, d);
    }
}


// ---- im2col of an 8-channel tensor for a 3x3 'same' convolution ------------------------------------------------------------------
// out[p][tap * 8 + c] = x[p + delta(tap)][c] (zero outside the image), taps 9..11 zero: rows of 96 elements of the 16-bit operand
// type, i.e. the input of a 1x1 fast-path convolution with K = 96 (three 32-wide K tiles).  The SPADE units' shared convolution
// (8 -> 128, K = 72: layers/spade.py:28) then runs on the 16-bit MFMA fast path instead of the generic per-lane gather.  One
// thread per (pixel, tap): a 32-byte (fp32 source) or 16-byte load, one 16-byte store; consecutive threads write consecutive chunks.
template <int HX, int HY>
__global__ __launch_bounds__(256) void im2col8_kernel(const void* __restrict__ x, void* __restrict__ out, int B, int H, int W) {
    const long n = (long)B * H * W * 12;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 12);
        const long p = i / 12;
        const int wq = (int)(p % W);
        const long r = p / W;
        const int hq = (int)(r % H);
        const long b = r / H;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int hi = hq + kh - 1, wi = wq + kw - 1;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi4 = lo;
        if (tap < 9 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
            const long src = ((b * H + hi) * W + wi) * 2;      // in units of 4 elements
            lo = ld4<HX>(x, src); hi4 = ld4<HX>(x, src + 1);
        }
        st4<HY>(out, i * 2, lo);
        st4<HY>(out, i * 2 + 1, hi4);
    }
}

// ---- fan-out gradients, batch concatenation, row gather (plumbing the autograd engine / torch.cat / index_select did with torch kernels) ----
struct Ptr8 { const void* p[8]; };
// out = p[0] + p[1] + ... + p[n-1] (left to right, fp32 accumulation), n4 groups of 4 elements + a scalar tail
template <int H>
__global__ __launch_bounds__(256) void sum_n_kernel(Ptr8 in, int n, void* __restrict__ out, long n4, long numel) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 acc = ld4<H>(in.p[0], i);
        for (int k = 1; k < n; ++k) acc += ld4<H>(in.p[k], i);
        st4<H>(out, i, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = n4 * 4; i < numel; ++i) {
            float acc = ld1<H>(in.p[0], i);
            for (int k = 1; k < n; ++k) acc += ld1<H>(in.p[k], i);
            if constexpr (H == 0) reinterpret_cast<float*>(out)[i] = acc;
            else if constexpr (H == 1) reinterpret_cast<__bf16*>(out)[i] = (__bf16)acc;
            else reinterpret_cast<_Float16*>(out)[i] = (_Float16)acc;
        }
}
// out[k * words .. (k + 1) * words) = p[k][0 .. words) in 4-byte words (a NULL part is written as zeros): concatenation of up to 8
// equally sized contiguous tensors along the leading axis in ONE launch (type-agnostic: bytes are moved, nothing is computed)
__global__ __launch_bounds__(256) void cat_words_kernel(Ptr8 in, int n, unsigned* __restrict__ out, long words) {
    const long w4 = words / 4;
    for (int k = 0; k < n; ++k) {
        const unsigned* src = reinterpret_cast<const unsigned*>(in.p[k]);
        unsigned* dst = out + (long)k * words;
        const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
        if (vec) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < w4; i += (long)gridDim.x * blockDim.x)
                reinterpret_cast<u4*>(dst)[i] = src ? reinterpret_cast<const u4*>(src)[i] : u4{0u, 0u, 0u, 0u};
            if (blockIdx.x == 0 && threadIdx.x == 0)
                for (long i = w4 * 4; i < words; ++i) dst[i] = src ? src[i] : 0u;
        } else {
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long)gridDim.x * blockDim.x)
                dst[i] = src ? src[i] : 0u;
        }
    }
}
// out[r] = src[idx[r]] for rows of `words` 4-byte words (index_select along the leading axis: sampling a fake pool)
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned* __restrict__ src, const long long* __restrict__ idx,
                                                          unsigned* __restrict__ out, int rows, long words, int src_rows) {
    const bool vec = (words % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        long long s = idx[r];
        if (s < 0 || s >= src_rows) continue;        // (the host validates the indices; never read out of range)
        const unsigned* a = src + (long)s * words;
        unsigned* o = out + (long)r * words;
        if (vec) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < words / 4; i += (long)gridDim.x * blockDim.x)
                reinterpret_cast<u4*>(o)[i] = reinterpret_cast<const u4*>(a)[i];
        } else {
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long)gridDim.x * blockDim.x) o[i] = a[i];
        }
    }
}
// out[m][0..C) = msk[m][0..C), out[m][C] = 1 unless some msk[m][c] == 1 exactly (model_executors/base_executor.py:83-87 of the
// reference: the background channel appended to a batch of masks)
__global__ __launch_bounds__(256) void add_residual_kernel(const float* __restrict__ msk, float* __restrict__ out, long M, int C) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        bool any = false;
        for (int c = 0; c < C; ++c) {
            const float v = msk[m * C + c];
            out[m * (C + 1) + c] = v;
            any = any || (v == 1.f);
        }
        out[m * (C + 1) + C] = any ? 0.f : 1.f;
    }
}


static inline bool hcode_ok(int h) { return h >= 0 && h <= 2; }
// (hx, hy) pairs that occur: a tensor is fp32 or THE 16-bit type of the run
#define DISPATCH_HH(hx, hy, LAUNCH)                                              \
    do {                                                                         \
        if (hx == 0 && hy == 0) { LAUNCH(0, 0); }                                \
        else if (hx == 0 && hy == 1) { LAUNCH(0, 1); }                           \
        else if (hx == 1 && hy == 1) { LAUNCH(1, 1); }                           \
        else if (hx == 1 && hy == 0) { LAUNCH(1, 0); }                           \
        else if (hx == 0 && hy == 2) { LAUNCH(0, 2); }                           \
        else if (hx == 2 && hy == 2) { LAUNCH(2, 2); }                           \
        else if (hx == 2 && hy == 0) { LAUNCH(2, 0); }                           \
        else return (int)hipErrorInvalidValue;                                   \
    } while (0)
#define DISPATCH_H(h, LAUNCH)                                                    \
    do {                                                                         \
        if (h == 0) { LAUNCH(0); } else if (h == 1) { LAUNCH(1); } else if (h == 2) { LAUNCH(2); } \
        else return (int)hipErrorInvalidValue;                                   \
    } while (0)

extern "C" {

// workspace: mmseg_norm_workspace_floats(C) floats.  hx: element code of x (0 fp32, 1 bf16, 2 fp16)
int mmseg_bn_stats_t(const void* x, const float* gamma, const float* beta, float* mean, float* invstd, float* scale, float* shift,
                     float* mov_mean, float* mov_var, float* ws, long M, int C, float eps, float momentum, int hx, void* stream) {
    if ((C & 63) || !hcode_ok(hx)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rows16(M, C);
    const long rpb = (M + nblk - 1) / nblk;
#define L(HX) hipLaunchKernelGGL((bn16_partial_kernel<0, HX, 0>), dim3(nblk, C / 64), dim3(256), 0, st, x, (const void*)nullptr, (const void*)nullptr, \
                                 (const float*)nullptr, (const float*)nullptr, ws, M, C, rpb, 0);                                                       \
              hipLaunchKernelGGL((bn16_stats_final_kernel<HX>), dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, x, gamma, beta, mean, invstd, \
                                 scale, shift, mov_mean, mov_var, nblk, C, M, eps, momentum)
    DISPATCH_H(hx, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_apply_t(const void* x, const float* scale, const float* shift, void* y, long M, int C, int relu, int hx, int hy, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n4 = M * (C / 4);
#define L(HX, HY) hipLaunchKernelGGL((bn16_apply_kernel<HX, HY>), dim3(grid16(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, n4, C / 4, relu)
    DISPATCH_HH(hx, hy, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
// dy, y: element code hy; x, dx: element code hx.  dgamma / dbeta may be NULL (frozen).  coef: 3*C floats of scratch
int mmseg_bn_bwd_t(const void* dy, const void* y, const void* x, const float* gamma, const float* mean, const float* invstd, void* dx,
                   float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu, int accumulate, int hx, int hy, void* stream) {
    if (C & 63) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rows16(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    const long n4 = M * (C / 4);
#define L(HX, HY) hipLaunchKernelGGL((bn16_partial_kernel<1, HX, HY>), dim3(nblk, C / 64), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu); \
                  hipLaunchKernelGGL(bn16_bwd_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, gamma, mean, invstd, dgamma, dbeta,   \
                                     coef, nblk, C, M, accumulate);                                                                                      \
                  hipLaunchKernelGGL((bn16_bwd_apply_kernel<HX, HY>), dim3(grid16(n4)), dim3(256), 0, st, dy, y, x, (const float*)coef, dx, n4, C / 4, relu)
    DISPATCH_HH(hx, hy, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_maxpool2_fwd_t(const void* x, void* y, int B, int H, int W, int C, int h, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
#define L(HH) hipLaunchKernelGGL((maxpool16_fwd_kernel<HH>), dim3(grid16(n)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4)
    DISPATCH_H(h, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_maxpool2_bwd_t(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, int h, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
#define L(HH) hipLaunchKernelGGL((maxpool16_bwd_kernel<HH>), dim3(grid16(n)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, B, H, W, C / 4)
    DISPATCH_H(h, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
// dx = (gradient of 2x2 max pooling) + add: a pooled tensor that also feeds a skip connection gets both gradients in one pass (the
// autograd engine would add them with a separate kernel: one more read and write of the largest activations).  add may be NULL.
int mmseg_maxpool2_bwd_add_t(const void* x, const void* y, const void* dy, const void* add, void* dx, int B, int H, int W, int C, int h,
                             void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
#define L(HH) hipLaunchKernelGGL((maxpool16_bwd_kernel<HH>), dim3(grid16(n)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, B, H, W, C / 4, add)
    DISPATCH_H(h, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_sum_n_t(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5, const void* p6,
                  const void* p7, int n, void* out, long numel, int h, void* stream) {
    if (n < 1 || n > 8 || numel < 0) return (int)hipErrorInvalidValue;
    Ptr8 in = {{p0, p1, p2, p3, p4, p5, p6, p7}};
    for (int k = 0; k < n; ++k) if (in.p[k] == nullptr) return (int)hipErrorInvalidValue;
    const long n4 = numel / 4;
#define L(HH) hipLaunchKernelGGL((sum_n_kernel<HH>), dim3(grid16(n4 + 1)), dim3(256), 0, (hipStream_t)stream, in, n, out, n4, numel)
    DISPATCH_H(h, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_cat_words(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5, const void* p6,
                    const void* p7, int n, void* out, long words, void* stream) {
    if (n < 1 || n > 8 || words < 0) return (int)hipErrorInvalidValue;
    Ptr8 in = {{p0, p1, p2, p3, p4, p5, p6, p7}};
    hipLaunchKernelGGL(cat_words_kernel, dim3(grid16(words / 4 + 1)), dim3(256), 0, (hipStream_t)stream, in, n, (unsigned*)out, words);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_gather_rows(const void* src, const long long* idx, void* out, int rows, long words, int src_rows, void* stream) {
    if (rows < 0 || words < 0) return (int)hipErrorInvalidValue;
    if (rows == 0 || words == 0) return 0;
    long bx = (words / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)bx, (unsigned)(rows < 4096 ? rows : 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)src, idx, (unsigned*)out, rows, words, src_rows);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_add_residual(const float* msk, float* out, long M, int C, void* stream) {
    if (C < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(add_residual_kernel, dim3(grid16(M)), dim3(256), 0, (hipStream_t)stream, msk, out, M, C);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_upsample2_bwd_t(const void* dy, void* dx, int B, int H, int W, int C, int h, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n = (long)B * H * W * (C / 4);
#define L(HH) hipLaunchKernelGGL((upsample16_bwd_kernel<HH>), dim3(grid16(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C / 4)
    DISPATCH_H(h, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}


// dx = dy * act'(y) (y = the activation's output; codes as mmseg_act_fwd), dy / y / dx stored with element code h; with bias_grad != NULL
// also bias_grad[c] (+)= sum over the M rows of dx[., c] in the same pass (needs C % 64 == 0 and ws = mmseg_colsum_workspace_floats(M, C)
// floats); otherwise M * C % 4 == 0 suffices.  Replaces act_bwd + colsum of a convolution with a fused activation and a bias.
int mmseg_act_bwd_bias_t(const void* dy, const void* y, void* dx, float* bias_grad, float* ws, long M, int C, int act, float alpha,
                         int accumulate, int h, void* stream) {
    if (!hcode_ok(h) || M <= 0 || C <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (bias_grad == nullptr) {
        const long n = M * C;
        if (n % 4 != 0) return (int)hipErrorInvalidValue;
#define L(HH) hipLaunchKernelGGL((act16_bwd_flat_kernel<HH>), dim3(grid16(n / 4)), dim3(256), 0, st, dy, y, dx, n / 4, act, alpha)
        DISPATCH_H(h, L);
#undef L
        return MMSEG_CHECK_LAUNCH();
    }
    if (C % 64 != 0 || ws == nullptr) return (int)hipErrorInvalidValue;
    long nb = 1024 / (C / 64);
    if (nb > 512) nb = 512;
    const long maxb = (M + 63) / 64;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    const long rpb = (M + nb - 1) / nb;
#define L(HH) hipLaunchKernelGGL((act16_bwd_kernel<HH, true>), dim3((unsigned)nb, C / 64), dim3(256), 0, st, dy, y, dx, ws, M, C, rpb, act, alpha)
    DISPATCH_H(h, L);
#undef L
    hipLaunchKernelGGL(colsum16_final_kernel, dim3(C), dim3(256), 0, st, (const float*)ws, bias_grad, (int)nb, C, accumulate);
    return MMSEG_CHECK_LAUNCH();
}


// out[c] (+)= sum over the M rows of x[., c], x stored with element code h (C % 64 == 0; ws = mmseg_colsum_workspace_floats(M, C))
int mmseg_colsum_t(const void* x, float* out, float* ws, long M, int C, int accumulate, int h, void* stream) {
    if (!hcode_ok(h) || M <= 0 || C <= 0 || (C % 64) != 0 || ws == nullptr) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    long nb = 1024 / (C / 64);
    if (nb > 512) nb = 512;
    const long maxb = (M + 63) / 64;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    const long rpb = (M + nb - 1) / nb;
#define L(HH) hipLaunchKernelGGL((colsum16_partial_kernel<HH>), dim3((unsigned)nb, C / 64), dim3(256), 0, st, x, ws, M, C, rpb)
    DISPATCH_H(h, L);
#undef L
    hipLaunchKernelGGL(colsum16_final_kernel, dim3(C), dim3(256), 0, st, (const float*)ws, out, (int)nb, C, accumulate);
    return MMSEG_CHECK_LAUNCH();
}

// out[B*H*W][96] (element code hy: 1 bf16 / 2 fp16) = im2col of x[B,H,W,8] (element code hx) for a 3x3 stride-1 'same' convolution,
// K index = tap * 8 + channel, columns 72..95 zero -- see im2col8_kernel
int mmseg_im2col8_t(const void* x, void* out, int B, int H, int W, int hx, int hy, void* stream) {
    if (!hcode_ok(hx) || (hy != 1 && hy != 2) || B <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    if (hx != 0 && hx != hy) return (int)hipErrorInvalidValue;
    const long n = (long)B * H * W * 12;
#define L(HX, HY) hipLaunchKernelGGL((im2col8_kernel<HX, HY>), dim3(grid16(n)), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W)
    DISPATCH_HH(hx, hy, L);
#undef L
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
