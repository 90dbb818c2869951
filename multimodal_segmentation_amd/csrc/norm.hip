// BatchNorm (keras BatchNormalization, axis=-1, momentum .99, eps 1e-3) and InstanceNorm
// (keras_contrib InstanceNormalization(axis=None, center=False, scale=False)) for NHWC fp32 on gfx950.
// HBM-bound: every pass reads/writes each element once with 16-byte accesses; per-channel reductions are
// two-stage (per-block partials in a workspace, then a tiny finalize) so results are run-to-run bitwise
// reproducible.  Reference call sites: utils/model_utils.py:6-12, models/unet.py:94-101,
// model_components/segmentor.py:16-21, layers/spade.py:27.
#include "common.hpp"

#define NORM_MAX_BLOCKS 1024

static inline int norm_blocks(long M) {
    long nb = (M + 511) / 512;
    if (nb > NORM_MAX_BLOCKS) nb = NORM_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// MODE 0: (x - shift, (x - shift)^2) with shift = x[0][c]      -> batch statistics
// MODE 1: (g, g * xhat), g = dy * [y > 0 if relu], xhat = (x - mean) * invstd   -> BN backward sums
template <int MODE>
__global__ void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                  float* __restrict__ part, long M, int C, long rows_per_block, int relu,
                                  const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr) {
    __shared__ float sm[2 * 256];
    const int tid = threadIdx.x;
    const int cw = min(C, 256), rl = 256 / cw;
    const int c_in = tid % cw, r_in = tid / cw;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int cb = 0; cb < C; cb += cw) {
        const int c = cb + c_in;
        float s1 = 0.f, s2 = 0.f;
        if (c < C && r_in < rl) {
            if (MODE == 0) {
                const float sh = x[c];
                for (long r = r0 + r_in; r < r1; r += rl) { const float d = x[r * C + c] - sh; s1 += d; s2 += d * d; }
            } else {
                const float mu = mean[c], is = invstd[c];
                const float sc = y ? 0.f : scale[c], sh = y ? 0.f : shift[c];
                for (long r = r0 + r_in; r < r1; r += rl) {
                    float g = dy[r * C + c];
                    const float xv = x[r * C + c];
                    // ReLU mask from the saved output, or recomputed from the input exactly as bn_apply_kernel computed the output
                    if (relu && !((y ? y[r * C + c] : fmaf(xv, sc, sh)) > 0.f)) g = 0.f;
                    s1 += g; s2 += g * (xv - mu) * is;
                }
            }
        }
        sm[tid] = s1; sm[256 + tid] = s2;
        __syncthreads();
        if (tid < cw && c < C) {
            float t1 = 0.f, t2 = 0.f;
            for (int k = 0; k < rl; ++k) { t1 += sm[k * cw + tid]; t2 += sm[256 + k * cw + tid]; }
            part[((size_t)blockIdx.x * 2) * C + c] = t1;
            part[((size_t)blockIdx.x * 2 + 1) * C + c] = t2;
        }
        __syncthreads();
    }
}

// Fast path for C % 64 == 0 (every BatchNorm of the models): grid (row blocks, C/64); a block is 16 float4 column
// lanes x 16 row lanes, so a wave reads 4 rows x 256 contiguous bytes per load instruction; the row loop is unrolled
// to keep >= 4 independent 16-byte loads per lane in flight (HBM-bound streaming, ~32 KB in flight per CU).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ part,
                                                            long M, int C, long rows_per_block, int relu,
                                                            const float* __restrict__ scale = nullptr,
                                                            const float* __restrict__ shift = nullptr) {
    __shared__ f32x4 sm[2][16][16];
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int C4 = C >> 2;
    const int col = blockIdx.y * 16 + c4;           // float4 column
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const f32x4* X = reinterpret_cast<const f32x4*>(x);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        const f32x4 sh = X[col];
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += 16) {
            const f32x4 d = X[r * C4 + col] - sh;
            s1 += d; s2 += d * d;
        }
    } else {
        const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[col], is = reinterpret_cast<const f32x4*>(invstd)[col];
        const f32x4* DY = reinterpret_cast<const f32x4*>(dy);
        const f32x4* Y = reinterpret_cast<const f32x4*>(y);
        if (relu && y == nullptr) {
            // ReLU mask recomputed from the input with bn_apply_kernel's own expression (same fused multiply-add, same bits):
            // one tensor less to read
            const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[col], sh = reinterpret_cast<const f32x4*>(shift)[col];
#pragma unroll 4
            for (long r = r0 + rl; r < r1; r += 16) {
                f32x4 g = DY[r * C4 + col];
                const f32x4 xv = X[r * C4 + col];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = fmaf(xv[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
                s1 += g; s2 += g * (xv - mu) * is;
            }
        } else {
#pragma unroll 4
            for (long r = r0 + rl; r < r1; r += 16) {
                f32x4 g = DY[r * C4 + col];
                if (relu) {
                    const f32x4 o = Y[r * C4 + col];
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
                }
                s1 += g; s2 += g * (X[r * C4 + col] - mu) * is;
            }
        }
    }
    sm[0][rl][c4] = s1; sm[1][rl][c4] = s2;
    __syncthreads();
    if (tid < 32) {                                   // 16 columns x {s1, s2}
        const int which = tid >> 4, cc = tid & 15;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[which][k][cc];
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + which) * C + (size_t)(blockIdx.y * 16 + cc) * 4) = t;
    }
}

// (reduce_partials_64 / FIN_NT: common.hpp -- shared with the typed-I/O kernels of act16.hip, which must sum in the same order)
// training statistics -> mean, invstd, fused scale/shift, moving-average update.  grid = ceil(C/64), block FIN_NT
__global__ __launch_bounds__(FIN_NT) void bn_stats_final_kernel(const float* __restrict__ part, const float* __restrict__ x, const float* __restrict__ gamma,
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
    const float mu = x[c] + d;
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

// ---- synchronised BatchNorm across data-parallel ranks (optional; parallel/dp.py) -------------------------------------------
// Local half of the statistics: per-channel mean and BIASED variance of this rank's rows -> stat2 [2][C].
__global__ __launch_bounds__(FIN_NT) void bn_stats_local_final_kernel(const float* __restrict__ part, const float* __restrict__ x, float* __restrict__ stat2,
                                            int nblk, int C, long M) {
    __shared__ float sm[2 * FIN_NT];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s1, s2;
    reduce_partials_64(part, nblk, C, c, lane, s1, s2, sm);
    if (c >= C || lane != 0) return;
    const float invM = 1.f / (float)M;
    const float d = s1 * invM;
    float var = s2 * invM - d * d;
    stat2[c] = x[c] + d;
    stat2[C + c] = var > 0.f ? var : 0.f;
}
// Combine the (mean, biased variance) pairs of R equally sized shards (gathered [R][2][C], fixed rank order -> every rank computes
// bit-identical statistics): mean = avg(mean_r), var = avg(var_r + (mean_r - mean)^2); then exactly bn_stats_final_kernel's outputs
// with M_total rows (moving variance from the UNBIASED global variance).
__global__ void bn_stats_combine_kernel(const float* __restrict__ gathered, int R, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ invstd,
                                        float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mov_mean,
                                        float* __restrict__ mov_var, long M_total, int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mu = 0.f;
    for (int r = 0; r < R; ++r) mu += gathered[((size_t)r * 2) * C + c];
    mu /= (float)R;
    float var = 0.f;
    for (int r = 0; r < R; ++r) {
        const float dm = gathered[((size_t)r * 2) * C + c] - mu;
        var += gathered[((size_t)r * 2 + 1) * C + c] + dm * dm;
    }
    var /= (float)R;
    const float is = rsqrtf(var + eps);
    mean[c] = mu; invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - mu * sc;
    if (mov_mean) {
        const float unb = M_total > 1 ? var * ((float)M_total / (float)(M_total - 1)) : var;
        mov_mean[c] -= (mov_mean[c] - mu) * (1.f - momentum);
        mov_var[c] -= (mov_var[c] - unb) * (1.f - momentum);
    }
}
// backward sums of this rank -> sums [2][C] = (sum g, sum g * xhat)
__global__ __launch_bounds__(FIN_NT) void bn_bwd_sums_final_kernel(const float* __restrict__ part, float* __restrict__ sums, int nblk, int C) {
    __shared__ float sm[2 * FIN_NT];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s1, s2;
    reduce_partials_64(part, nblk, C, c, lane, s1, s2, sm);
    if (c >= C || lane != 0) return;
    sums[c] = s1; sums[C + c] = s2;
}
// dgamma / dbeta take THIS rank's sums (the data-parallel gradient all-reduce averages them like every other weight gradient);
// the dx coefficients take the sums over ALL ranks and the global row count
__global__ void bn_bwd_finish_kernel(const float* __restrict__ local, const float* __restrict__ global, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ coef, int C, long M_total, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dgamma) {
        if (accumulate) { dbeta[c] += local[c]; dgamma[c] += local[C + c]; }
        else { dbeta[c] = local[c]; dgamma[c] = local[C + c]; }
    }
    const float s1 = global[c], s2 = global[C + c];
    const float invM = 1.f / (float)M_total, is = invstd[c], ga = gamma[c], mu = mean[c];
    const float A = ga * is;
    const float Bc = -ga * is * is * s2 * invM;
    coef[c] = A; coef[C + c] = Bc; coef[2 * C + c] = -A * s1 * invM - Bc * mu;
}

// inference: scale/shift from the moving statistics
__global__ void bn_infer_prep_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mov_mean,
                                     const float* __restrict__ mov_var, float* __restrict__ scale, float* __restrict__ shift,
                                     int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(mov_var[c] + eps);
    scale[c] = sc; shift[c] = beta[c] - mov_mean[c] * sc;
}

// inference BatchNorm folded behind a convolution with bias b: bn(conv + b) = conv * scale + (shift + b * scale)
__global__ void bn_infer_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mov_mean,
                                     const float* __restrict__ mov_var, const float* __restrict__ conv_bias, float* __restrict__ scale,
                                     float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(mov_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mov_mean[c] * sc + (conv_bias ? conv_bias[c] * sc : 0.f);
}

// y = x * scale[c] + shift[c]  (+ReLU)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                float* __restrict__ y, long n4, int C4, int relu) {
    const f32x4* S = reinterpret_cast<const f32x4*>(scale);
    const f32x4* T = reinterpret_cast<const f32x4*>(shift);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i], sc = S[c], sh = T[c];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(xv[e], sc[e], sh[e]);      // (the backward pass recomputes the ReLU mask with this expression)
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// finalize backward sums: dbeta, dgamma and per-channel coefficients  dx = A*g + Bc*x + Cc
__global__ __launch_bounds__(FIN_NT) void bn_bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    float* __restrict__ coef /* [3][C] */, int nblk, int C, long M, int accumulate) {
    __shared__ float sm[2 * FIN_NT];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s1, s2;
    reduce_partials_64(part, nblk, C, c, lane, s1, s2, sm);
    if (c >= C || lane != 0) return;
    if (accumulate) { dbeta[c] += s1; dgamma[c] += s2; }      // straight into the gradient arena
    else { dbeta[c] = s1; dgamma[c] = s2; }
    const float invM = 1.f / (float)M, is = invstd[c], ga = gamma[c], mu = mean[c];
    const float A = ga * is;
    const float Bc = -ga * is * is * s2 * invM;
    coef[c] = A; coef[C + c] = Bc; coef[2 * C + c] = -A * s1 * invM - Bc * mu;
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                    const float* __restrict__ coef, float* __restrict__ dx, long n4, int C4, int relu,
                                    const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr) {
    const f32x4* A = reinterpret_cast<const f32x4*>(coef);
    const f32x4* Bc = A + C4;
    const f32x4* Cc = A + 2 * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4;
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        if (relu) {
            if (y) {
                const f32x4 o = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
            } else {
                const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c], sh = reinterpret_cast<const f32x4*>(shift)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = fmaf(xv[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
            }
        }
        reinterpret_cast<f32x4*>(dx)[i] = A[c] * g + Bc[c] * xv + Cc[c];
    }
}

// =====================================================================================================
// InstanceNorm over (H, W, C) jointly per sample:  xn = (x - mean_b) / (std_b + eps)
// fused SPADE modulation + LeakyReLU:  y = leaky( xn * (1 + gamma) + beta )      (layers/spade.py:7-33,51-54)
// =====================================================================================================
// per-sample sums: grid (nchunk, B): part[b][chunk][2] = (sum(x - x0), sum((x - x0)^2))
__global__ void in_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long per_sample, int nchunk) {
    __shared__ float red[17];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * per_sample;
    const float sh = xb[0];
    const long per = (per_sample + nchunk - 1) / nchunk;
    const long i0 = (long)blockIdx.x * per, i1 = min(per_sample, i0 + per);
    float s1 = 0.f, s2 = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const float d = xb[i] - sh; s1 += d; s2 += d * d; }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { part[((size_t)b * nchunk + blockIdx.x) * 2] = s1; part[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = s2; }
}
__global__ void in_final_kernel(const float* __restrict__ part, const float* __restrict__ x, float* __restrict__ stat /* [B][2]: mean, 1/(std+eps) */,
                                long per_sample, int nchunk, int B, float eps) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) { s1 += part[((size_t)b * nchunk + k) * 2]; s2 += part[((size_t)b * nchunk + k) * 2 + 1]; }
    const float inv = 1.f / (float)per_sample;
    const float d = s1 * inv;
    float var = s2 * inv - d * d;
    var = var > 0.f ? var : 0.f;
    stat[2 * b] = x[(size_t)b * per_sample] + d;
    stat[2 * b + 1] = 1.f / (sqrtf(var) + eps);
}
// y = act( (x - mean)*rstd * (1 + gamma) + beta ); gamma/beta nullptr -> plain normalisation. act_alpha<0: no activation
__global__ void in_apply_kernel(const float* __restrict__ x, const float* __restrict__ stat, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long per_sample, long n, float act_alpha) {
    // grid (chunks, B): one sample per blockIdx.y, 16-byte accesses (per_sample = H*W*C is a multiple of 4)
    const int b = blockIdx.y;
    const float mu = stat[2 * b], rs = stat[2 * b + 1];
    const size_t off4 = (size_t)b * (per_sample >> 2);
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + off4;
    const f32x4* G = gamma ? reinterpret_cast<const f32x4*>(gamma) + off4 : nullptr;
    const f32x4* Bt = gamma ? reinterpret_cast<const f32x4*>(beta) + off4 : nullptr;
    f32x4* Y = reinterpret_cast<f32x4*>(y) + off4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (per_sample >> 2); i += (long)gridDim.x * blockDim.x) {
        f32x4 v = (X[i] - mu) * rs;
        if (G) v = v * (G[i] + 1.f) + Bt[i];
        if (act_alpha >= 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * act_alpha;
        }
        Y[i] = v;
    }
    (void)n;
}
// backward of the fused op.  With xn = (x-mean)*rstd, u = xn*(1+gamma)+beta, y = leaky(u):
//   g = dy*leaky'(u); dgamma = g*xn; dbeta = g; dxn = g*(1+gamma)
//   dx = rstd * ( dxn - mean_b(dxn) - xn * rstd*sqrt(var)... )  -- see in_bwd_final for the (std+eps) form
// pass 1: elementwise dgamma/dbeta + per-sample sums S1 = sum dxn, S2 = sum dxn*xn
__global__ void in_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stat,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, float* __restrict__ dxn_out, float* __restrict__ part,
                                      long per_sample, int nchunk, float act_alpha) {
    __shared__ float red[17];
    const int b = blockIdx.y;
    const size_t off = (size_t)b * per_sample;
    const float mu = stat[2 * b], rs = stat[2 * b + 1];
    const long per = (per_sample + nchunk - 1) / nchunk;
    const long i0 = (long)blockIdx.x * per, i1 = min(per_sample, i0 + per);
    float s1 = 0.f, s2 = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float xn = (x[off + i] - mu) * rs;
        float g = dy[off + i];
        float dxn;
        if (gamma) {
            const float ga = gamma[off + i];
            const float u = xn * (1.f + ga) + beta[off + i];
            if (act_alpha >= 0.f && u < 0.f) g *= act_alpha;
            dgamma[off + i] = g * xn; dbeta[off + i] = g;
            dxn = g * (1.f + ga);
        } else {
            if (act_alpha >= 0.f && xn < 0.f) g *= act_alpha;
            dxn = g;
        }
        dxn_out[off + i] = dxn;
        s1 += dxn; s2 += dxn * xn;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { part[((size_t)b * nchunk + blockIdx.x) * 2] = s1; part[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = s2; }
}
// xn = (x-mu)/(sd+eps).  d xn_i / d x_j = [delta_ij - 1/N]/(sd+eps) - (x_i-mu)/(sd+eps)^2 * (x_j-mu)/(N sd)
//  => dx_j = rs*(dxn_j - S1/N) - rs^2 * (x_j - mu)/(N sd) * sum_i dxn_i (x_i-mu)
//          = rs*(dxn_j - S1/N) - xn_j * rs * S2 / (N * sd) * ... with S2 = sum dxn*xn, (x_i-mu) = xn_i/rs:
//     dx_j = rs*(dxn_j - S1/N) - xn_j * S2 / (N * sd)
__global__ void in_bwd_apply_kernel(const float* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ stat,
                                    const float* __restrict__ part, float* __restrict__ dx, long per_sample, long n, int nchunk, float eps) {
    // grid (chunks, B).  The per-sample sums are reduced ONCE per block (they used to be re-summed by every element).
    __shared__ float red[17];
    const int b = blockIdx.y;
    float p1 = 0.f, p2 = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += blockDim.x) { p1 += part[((size_t)b * nchunk + k) * 2]; p2 += part[((size_t)b * nchunk + k) * 2 + 1]; }
    const float s1 = block_sum(p1, red);
    const float s2 = block_sum(p2, red);
    const float mu = stat[2 * b], rs = stat[2 * b + 1];
    const float sd = 1.f / rs - eps;
    const float invN = 1.f / (float)per_sample;
    const float c1 = s1 * invN, c2 = sd > 0.f ? s2 * invN / sd : 0.f;
    const size_t off4 = (size_t)b * (per_sample >> 2);
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + off4;
    const f32x4* D = reinterpret_cast<const f32x4*>(dxn) + off4;
    f32x4* O = reinterpret_cast<f32x4*>(dx) + off4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (per_sample >> 2); i += (long)gridDim.x * blockDim.x) {
        const f32x4 xn = (X[i] - mu) * rs;
        O[i] = (D[i] - c1) * rs - xn * c2;
    }
    (void)n;
}

// ---- the same with gamma and beta as the two halves of ONE tensor gb [B*H*W][2C] (gamma = channels [0, C), beta = [C, 2C)): the
//      output of the fused gamma + beta convolution of a SPADE unit (ops.conv2d_pair).  16-byte accesses; C % 4 == 0. ----
template <int H, int HY>       // element codes of gb and of the output y (16-bit activation storage: both live in HBM in the 16-bit type)
__global__ void in_apply_gb_kernel(const float* __restrict__ x, const float* __restrict__ stat, const void* __restrict__ gb,
                                   void* __restrict__ y, long per_sample, int C4, float act_alpha) {
    const int b = blockIdx.y;
    const float mu = stat[2 * b], rs = stat[2 * b + 1];
    const long n4 = per_sample >> 2;
    const size_t off4 = (size_t)b * n4;
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + off4;
    const long gb0 = 2 * (long)off4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const f32x4 ga = ld4<H>(gb, gb0 + pix * 2 * C4 + c4), be = ld4<H>(gb, gb0 + pix * 2 * C4 + C4 + c4);
        f32x4 v = (X[i] - mu) * rs;
        v = v * (ga + 1.f) + be;
        if (act_alpha >= 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * act_alpha;
        }
        st4<HY>(y, (long)off4 + i, v);
    }
}
// pass 1 of the backward: dgb (dgamma | dbeta), dxn and the per-sample sums; expressions as in_bwd_partial_kernel
template <int H, int HY>       // element codes of gb / dgb and of dy (a gradient is stored like its tensor)
__global__ void in_bwd_partial_gb_kernel(const void* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stat,
                                         const void* __restrict__ gb, void* __restrict__ dgb, float* __restrict__ dxn_out,
                                         float* __restrict__ part, long per_sample, int nchunk, int C4, float act_alpha) {
    __shared__ float red[17];
    const int b = blockIdx.y;
    const long n4 = per_sample >> 2;
    const size_t off4 = (size_t)b * n4;
    const float mu = stat[2 * b], rs = stat[2 * b + 1];
    const long per = (n4 + nchunk - 1) / nchunk;
    const long i0 = (long)blockIdx.x * per, i1 = min(n4, i0 + per);
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + off4;
    const long gb0 = 2 * (long)off4;
    f32x4* DXN = reinterpret_cast<f32x4*>(dxn_out) + off4;
    float s1 = 0.f, s2 = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const f32x4 xn = (X[i] - mu) * rs;
        f32x4 g = ld4<HY>(dy, (long)off4 + i);
        const f32x4 ga = ld4<H>(gb, gb0 + pix * 2 * C4 + c4), be = ld4<H>(gb, gb0 + pix * 2 * C4 + C4 + c4);
        f32x4 dxn, dga;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = xn[e] * (1.f + ga[e]) + be[e];
            if (act_alpha >= 0.f && u < 0.f) g[e] *= act_alpha;
            dga[e] = g[e] * xn[e];
            dxn[e] = g[e] * (1.f + ga[e]);
            s1 += dxn[e]; s2 += dxn[e] * xn[e];
        }
        st4<H>(dgb, gb0 + pix * 2 * C4 + c4, dga);
        st4<H>(dgb, gb0 + pix * 2 * C4 + C4 + c4, g);
        DXN[i] = dxn;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { part[((size_t)b * nchunk + blockIdx.x) * 2] = s1; part[((size_t)b * nchunk + blockIdx.x) * 2 + 1] = s2; }
}

// row blocks of the float4 fast path: ~1024 blocks in total, at most 512 row blocks, at least 64 rows per block
static inline int v4_row_blocks(long M, int C) {
    long nb = 1024 / (C / 64);
    if (nb > 512) nb = 512;
    const long maxb = (M + 63) / 64;
    if (nb > maxb) nb = maxb;
    if (nb < 1) nb = 1;
    return (int)nb;
}

static inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" {

int mmseg_norm_workspace_floats(int C) { return NORM_MAX_BLOCKS * 2 * C; }

// training-mode statistics: writes mean, invstd, scale, shift [C each]; updates moving stats when non-null
int mmseg_bn_stats(const float* x, const float* gamma, const float* beta, float* mean, float* invstd, float* scale, float* shift,
                   float* mov_mean, float* mov_var, float* ws, long M, int C, float eps, float momentum, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int nblk;
    if ((C & 63) == 0) {
        nblk = v4_row_blocks(M, C);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_v4_kernel<0>, dim3(nblk, C / 64), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, ws, M, C, rpb, 0);
    } else {
        nblk = norm_blocks(M);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblk), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, ws, M, C, rpb, 0);
    }
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, x, gamma, beta, mean, invstd,
                       scale, shift, mov_mean, mov_var, nblk, C, M, eps, momentum);
    return MMSEG_CHECK_LAUNCH();
}
// ---- synchronised BatchNorm (data parallel): the two halves of mmseg_bn_stats / mmseg_bn_bwd with the cross-rank exchange
//      (all-gather of [2][C] statistics, all-reduce of [2][C] sums -- done by the caller) in between -------------------------
static int bn_launch_partial(int mode, const float* x, const float* dy, const float* y, const float* mean, const float* invstd,
                             float* ws, long M, int C, int relu, hipStream_t st) {
    int nblk;
    if ((C & 63) == 0) {
        nblk = v4_row_blocks(M, C);
        const long rpb = (M + nblk - 1) / nblk;
        if (mode == 0) hipLaunchKernelGGL(bn_partial_v4_kernel<0>, dim3(nblk, C / 64), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
        else hipLaunchKernelGGL(bn_partial_v4_kernel<1>, dim3(nblk, C / 64), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
    } else {
        nblk = norm_blocks(M);
        const long rpb = (M + nblk - 1) / nblk;
        if (mode == 0) hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblk), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
        else hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblk), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
    }
    return nblk;
}
int mmseg_bn_stats_local(const float* x, float* stat2, float* ws, long M, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nblk = bn_launch_partial(0, x, nullptr, nullptr, nullptr, nullptr, ws, M, C, 0, st);
    hipLaunchKernelGGL(bn_stats_local_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, x, stat2, nblk, C, M);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_stats_combine(const float* gathered, int R, const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                           float* shift, float* mov_mean, float* mov_var, long M_total, int C, float eps, float momentum, void* stream) {
    if (R < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_stats_combine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gathered, R, gamma, beta, mean,
                       invstd, scale, shift, mov_mean, mov_var, M_total, C, eps, momentum);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_bwd_sums(const float* dy, const float* y, const float* x, const float* mean, const float* invstd, float* sums, float* ws,
                      long M, int C, int relu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nblk = bn_launch_partial(1, x, dy, y, mean, invstd, ws, M, C, relu, st);
    hipLaunchKernelGGL(bn_bwd_sums_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, sums, nblk, C);
    return MMSEG_CHECK_LAUNCH();
}
// dgamma / dbeta (may be NULL) <- local sums; coef [3][C] <- global sums and M_total; then dx = mmseg_bn_bwd_apply
int mmseg_bn_bwd_finish(const float* sums_local, const float* sums_global, const float* gamma, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* coef, int C, long M_total, int accumulate, void* stream) {
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums_local, sums_global, gamma, mean,
                       invstd, dgamma, dbeta, coef, C, M_total, accumulate);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* coef, float* dx, long M, int C, int relu, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, dy, y, x, coef, dx, n4, C / 4, relu);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_bn_infer_prep(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, float* scale, float* shift,
                        int C, float eps, void* stream) {
    hipLaunchKernelGGL(bn_infer_prep_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mov_mean, mov_var, scale, shift, C, eps);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_infer_fold(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, const float* conv_bias,
                        float* scale, float* shift, int C, float eps, void* stream) {
    hipLaunchKernelGGL(bn_infer_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mov_mean, mov_var,
                       conv_bias, scale, shift, C, eps);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_bn_apply(const float* x, const float* scale, const float* shift, float* y, long M, int C, int relu, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, n4, C / 4, relu);
    return MMSEG_CHECK_LAUNCH();
}
// coef: 3*C floats of scratch
int mmseg_bn_bwd(const float* dy, const float* y, const float* x, const float* gamma, const float* mean, const float* invstd,
                 float* dx, float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu, int accumulate, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    int nblk;
    if ((C & 63) == 0) {
        nblk = v4_row_blocks(M, C);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_v4_kernel<1>, dim3(nblk, C / 64), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
    } else {
        nblk = norm_blocks(M);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblk), dim3(256), 0, st, x, dy, y, mean, invstd, ws, M, C, rpb, relu);
    }
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, gamma, mean, invstd, dgamma, dbeta, coef, nblk, C, M, accumulate);
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, dy, y, x, (const float*)coef, dx, n4, C / 4, relu);
    return MMSEG_CHECK_LAUNCH();
}
// The same without the saved output: the ReLU mask is recomputed from x with the forward pass's scale / shift (mmseg_bn_stats
// outputs; bit-identical to the mask of mmseg_bn_apply's output) -- 5 tensor passes instead of 7
int mmseg_bn_bwd_x(const float* dy, const float* x, const float* scale, const float* shift, const float* gamma, const float* mean,
                   const float* invstd, float* dx, float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu,
                   int accumulate, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const float* none = nullptr;
    int nblk;
    if ((C & 63) == 0) {
        nblk = v4_row_blocks(M, C);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_v4_kernel<1>, dim3(nblk, C / 64), dim3(256), 0, st, x, dy, none, mean, invstd, ws, M, C, rpb, relu, scale, shift);
    } else {
        nblk = norm_blocks(M);
        const long rpb = (M + nblk - 1) / nblk;
        hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblk), dim3(256), 0, st, x, dy, none, mean, invstd, ws, M, C, rpb, relu, scale, shift);
    }
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(FIN_NT), 0, st, (const float*)ws, gamma, mean, invstd, dgamma, dbeta, coef, nblk, C, M, accumulate);
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, dy, none, x, (const float*)coef, dx, n4, C / 4, relu, scale, shift);
    return MMSEG_CHECK_LAUNCH();
}

#define IN_CHUNKS 64
// blocks per sample of the element-wise InstanceNorm passes: ~4 float4 per thread, at most 512
static inline int in_apply_chunks(long per_sample) {
    long c = (per_sample / 4 + 1023) / 1024;
    if (c > 512) c = 512;
    if (c < 1) c = 1;
    return (int)c;
}
int mmseg_in_workspace_floats(int B) { return B * IN_CHUNKS * 2; }
// stat: [B][2] output (mean, 1/(std+eps)); y = act((x-mean)*rstd*(1+gamma)+beta)
int mmseg_instnorm_spade_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, float* ws, int B,
                             long per_sample, float eps, float act_alpha, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(in_partial_kernel, dim3(IN_CHUNKS, B), dim3(256), 0, st, x, ws, per_sample, IN_CHUNKS);
    hipLaunchKernelGGL(in_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, x, stat, per_sample, IN_CHUNKS, B, eps);
    const long n = (long)B * per_sample;
    if (per_sample & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(in_apply_kernel, dim3(in_apply_chunks(per_sample), B), dim3(256), 0, st, x, (const float*)stat, gamma, beta, y, per_sample, n, act_alpha);
    return MMSEG_CHECK_LAUNCH();
}
// dxn: scratch [B*per_sample]; dgamma/dbeta may be nullptr when gamma is nullptr
int mmseg_instnorm_spade_bwd(const float* dy, const float* x, const float* stat, const float* gamma, const float* beta, float* dx,
                             float* dgamma, float* dbeta, float* dxn, float* ws, int B, long per_sample, float eps, float act_alpha,
                             void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(in_bwd_partial_kernel, dim3(IN_CHUNKS, B), dim3(256), 0, st, dy, x, stat, gamma, beta, dgamma, dbeta, dxn, ws,
                       per_sample, IN_CHUNKS, act_alpha);
    const long n = (long)B * per_sample;
    if (per_sample & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(in_bwd_apply_kernel, dim3(in_apply_chunks(per_sample), B), dim3(256), 0, st, (const float*)dxn, x, stat, (const float*)ws, dx, per_sample, n, IN_CHUNKS, eps);
    return MMSEG_CHECK_LAUNCH();
}

// gamma and beta as the halves of one tensor gb [B * H * W][2C] (see in_apply_gb_kernel); per_sample = H * W * C; h: element code of
// gb / dgb, hy: element code of the output y / of its gradient dy (0 fp32, 1 bf16, 2 fp16) -- x and dx stay fp32
int mmseg_instnorm_spade_fwd_gb_t(const float* x, const void* gb, void* y, float* stat, float* ws, int B, long per_sample, int C, float eps,
                                  float act_alpha, int h, int hy, void* stream) {
    if ((C & 3) || C <= 0 || per_sample % C != 0 || h < 0 || h > 2 || hy < 0 || hy > 2 || (h && hy && h != hy)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(in_partial_kernel, dim3(IN_CHUNKS, B), dim3(256), 0, st, x, ws, per_sample, IN_CHUNKS);
    hipLaunchKernelGGL(in_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, x, stat, per_sample, IN_CHUNKS, B, eps);
    const dim3 grid(in_apply_chunks(per_sample), B);
#define L(HH, HYY) hipLaunchKernelGGL((in_apply_gb_kernel<HH, HYY>), grid, dim3(256), 0, st, x, (const float*)stat, gb, y, per_sample, C / 4, act_alpha)
    if (h == 0 && hy == 0) L(0, 0); else if (h == 1 && hy == 0) L(1, 0); else if (h == 2 && hy == 0) L(2, 0);
    else if (h == 0 && hy == 1) L(0, 1); else if (h == 0 && hy == 2) L(0, 2); else if (h == 1) L(1, 1); else L(2, 2);
#undef L
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_instnorm_spade_bwd_gb_t(const void* dy, const float* x, const float* stat, const void* gb, float* dx, void* dgb, float* dxn,
                                  float* ws, int B, long per_sample, int C, float eps, float act_alpha, int h, int hy, void* stream) {
    if ((C & 3) || C <= 0 || per_sample % C != 0 || h < 0 || h > 2 || hy < 0 || hy > 2 || (h && hy && h != hy)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(IN_CHUNKS, B);
#define L(HH, HYY) hipLaunchKernelGGL((in_bwd_partial_gb_kernel<HH, HYY>), grid, dim3(256), 0, st, dy, x, stat, gb, dgb, dxn, ws, per_sample, IN_CHUNKS, C / 4, act_alpha)
    if (h == 0 && hy == 0) L(0, 0); else if (h == 1 && hy == 0) L(1, 0); else if (h == 2 && hy == 0) L(2, 0);
    else if (h == 0 && hy == 1) L(0, 1); else if (h == 0 && hy == 2) L(0, 2); else if (h == 1) L(1, 1); else L(2, 2);
#undef L
    const long n = (long)B * per_sample;
    hipLaunchKernelGGL(in_bwd_apply_kernel, dim3(in_apply_chunks(per_sample), B), dim3(256), 0, st, (const float*)dxn, x, stat, (const float*)ws, dx, per_sample, n, IN_CHUNKS, eps);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_instnorm_spade_fwd_gb(const float* x, const float* gb, float* y, float* stat, float* ws, int B, long per_sample, int C, float eps,
                                float act_alpha, void* stream) {
    return mmseg_instnorm_spade_fwd_gb_t(x, gb, y, stat, ws, B, per_sample, C, eps, act_alpha, 0, 0, stream);
}
int mmseg_instnorm_spade_bwd_gb(const float* dy, const float* x, const float* stat, const float* gb, float* dx, float* dgb, float* dxn,
                                float* ws, int B, long per_sample, int C, float eps, float act_alpha, void* stream) {
    return mmseg_instnorm_spade_bwd_gb_t(dy, x, stat, gb, dx, dgb, dxn, ws, B, per_sample, C, eps, act_alpha, 0, 0, stream);
}

}  // extern "C"
