// HBM-bound elementwise / small-reduction kernels of the DAFNet/MMSDNet step (gfx950).
// All tensors NHWC fp32, contiguous.  Grid-stride loops, 16-byte accesses where the shape allows.
// Reference ops replaced (file:line in /root/reference): keras Activation/LeakyReLU, Add, MaxPooling2D,
// UpSampling2D grad, softmax + layers/rounding.py:23-42, layers/film.py:26-36, keras Maximum
// (model_components/anatomy_fuser.py:33), channel slicing Lambda (models/dafnet.py:187),
// utils/sdnet_utils.py:9-21 + costs.py:186-189.
#include "common.hpp"

static inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = act_apply(x[i], act, alpha);
}

// dx = dy * act'(y)   (y = activation OUTPUT)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               long n4, long n, int act, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i], o = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = g[e] * act_grad_from_out(o[e], act, alpha);
        reinterpret_cast<f32x4*>(dx)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = n4 * 4; i < n; ++i) dx[i] = dy[i] * act_grad_from_out(y[i], act, alpha);
}

// out = a * sa + b * sb
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                             long n4, long n, float sa, float sb) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 u = reinterpret_cast<const f32x4*>(a)[i], v = reinterpret_cast<const f32x4*>(b)[i];
        reinterpret_cast<f32x4*>(out)[i] = u * sa + v * sb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = n4 * 4; i < n; ++i) out[i] = a[i] * sa + b[i] * sb;
}

__global__ void fill_kernel(float* __restrict__ x, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}

// ---- column sums: out[c] = sum_m x[m][c]; two deterministic stages -------------------------------
// stage 1: block (bx over row chunks) accumulates its rows; threads own (row lane, channel) pairs
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long M, int C,
                                      long rows_per_block) {
    // blockDim.x = 256; channel = tid % cw, row lane = tid / cw with cw = min(C, 256)
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const int cw = min(C, 256);
    const int rl = 256 / cw;             // row lanes (>= 1)
    const int c_in = tid % cw, r_in = tid / cw;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int cb = 0; cb < C; cb += cw) {
        const int c = cb + c_in;
        float s = 0.f;
        if (c < C && r_in < rl)
            for (long r = r0 + r_in; r < r1; r += rl) s += x[r * C + c];
        sm[tid] = s;
        __syncthreads();
        if (tid < cw && c < C) {
            float t = 0.f;
            for (int k = 0; k < rl; ++k) t += sm[k * cw + tid];
            part[(size_t)blockIdx.x * C + c] = t;
        }
        __syncthreads();
    }
}

// float4 fast path: x viewed as [M][C] with C % 64 == 0; grid (row blocks, C/64), block = 16 float4 columns x 16 row lanes
__global__ __launch_bounds__(256) void colsum_v4_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long M, int C,
                                                                long rows_per_block) {
    __shared__ f32x4 sm[16][16];
    const int tid = threadIdx.x, c4 = tid & 15, rl = tid >> 4;
    const int C4 = C >> 2, col = blockIdx.y * 16 + c4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const f32x4* X = reinterpret_cast<const f32x4*>(x);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long r = r0 + rl; r < r1; r += 16) s += X[r * C4 + col];
    sm[rl][c4] = s;
    __syncthreads();
    if (tid < 16) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][tid];
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (size_t)(blockIdx.y * 16 + tid) * 4) = t;
    }
}
// out[c] = scale * sum over blocks and over the pseudo-channels p = j*C + c of a [.][CP] partial table (CP = 64 when
// a small-C tensor was re-viewed as 64 pseudo-channels, CP = C otherwise).  one block per output channel.
__global__ void colsum_fold_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C, int CP,
                                         float scale, int accumulate) {
    __shared__ float red[17];
    const int c = blockIdx.x, groups = CP / C;
    const int total = groups * nblk;
    float a = 0.f;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int j = i % groups, b = i / groups;
        a += part[(size_t)b * CP + j * C + c];
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + a * scale : a * scale;
}

__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C,
                                    float scale, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int b = 0; b < nblk; ++b) t += part[(size_t)b * C + c];
    t *= scale;
    out[c] = accumulate ? out[c] + t : t;
}

// ---- 2x2 max pooling ------------------------------------------------------------------------------
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const long n = (long)B * Ho * Wo * C4;
    const f32x4* X = reinterpret_cast<const f32x4*>(x);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int wo = r % Wo; r /= Wo;
        const int ho = r % Ho; const int b = r / Ho;
        const long base = (((long)b * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const f32x4 v00 = X[base], v01 = X[base + C4], v10 = X[base + (long)W * C4], v11 = X[base + (long)W * C4 + C4];
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
        reinterpret_cast<f32x4*>(y)[i] = m;
    }
}

// gradient goes to the FIRST maximum in row-major window order (TF MaxPoolGrad / torch CPU rule)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                    float* __restrict__ dx, int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const long n = (long)B * Ho * Wo * C4;
    const f32x4* X = reinterpret_cast<const f32x4*>(x);
    f32x4* DX = reinterpret_cast<f32x4*>(dx);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int wo = r % Wo; r /= Wo;
        const int ho = r % Ho; const int b = r / Ho;
        const long base = (((long)b * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const long o01 = C4, o10 = (long)W * C4, o11 = (long)W * C4 + C4;
        const f32x4 v00 = X[base], v01 = X[base + o01], v10 = X[base + o10], v11 = X[base + o11];
        const f32x4 m = reinterpret_cast<const f32x4*>(y)[i], g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 g00, g01, g10, g11;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = v00[e] == m[e];
            const bool bb = !a && v01[e] == m[e];
            const bool cc = !a && !bb && v10[e] == m[e];
            const bool dd = !a && !bb && !cc;
            g00[e] = a ? g[e] : 0.f; g01[e] = bb ? g[e] : 0.f; g10[e] = cc ? g[e] : 0.f; g11[e] = dd ? g[e] : 0.f;
        }
        DX[base] = g00; DX[base + o01] = g01; DX[base + o10] = g10; DX[base + o11] = g11;
    }
    // odd trailing rows/cols (not pooled) are zeroed by the launcher when H or W is odd
}

// dx[b, h, w, c] = sum of the 2x2 block of dy (gradient of nearest x2 up-sampling)
__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C4) {
    // H, W: low-res dims; dy is [B, 2H, 2W, C]
    const long n = (long)B * H * W * C4;
    const f32x4* DY = reinterpret_cast<const f32x4*>(dy);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int w = r % W; r /= W;
        const int h = r % H; const int b = r / H;
        const long base = (((long)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
        reinterpret_cast<f32x4*>(dx)[i] = DY[base] + DY[base + C4] + DY[base + 2L * W * C4] + DY[base + 2L * W * C4 + C4];
    }
}

// ---- channel softmax (+ round half-to-even) ---------------------------------------------------------
template <int C>
__global__ void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ p, float* __restrict__ s, long npix) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        float v[C];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { v[c] = x[i * C + c]; mx = fmaxf(mx, v[c]); }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - mx); sum += v[c]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float q = v[c] * inv;
            p[i * C + c] = q;
            if (s) s[i * C + c] = rintf(q);
        }
    }
}

// dx = p * (dy - sum_c dy*p)
template <int C>
__global__ void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ p, float* __restrict__ dx, long npix) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        float g[C], q[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { g[c] = dy[i * C + c]; q[c] = p[i * C + c]; dot += g[c] * q[c]; }
#pragma unroll
        for (int c = 0; c < C; ++c) dx[i * C + c] = q[c] * (g[c] - dot);
    }
}

// ---- FiLM: y = leaky(x * gamma[b,c] + beta[b,c]) (+ optional residual: y += res) ---------------------
__global__ void film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ res, float* __restrict__ y, int B, long HW, int C, float alpha) {
    const long n = (long)B * HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C; const int b = i / (HW * C);
        float v = x[i] * gamma[b * C + c] + beta[b * C + c];
        v = v >= 0.f ? v : v * alpha;
        y[i] = res ? v + res[i] : v;
    }
}

// C % 4 == 0, 16-byte aligned tensors: one float4 of channels per thread step, grid.y = sample (no 64-bit divisions)
__global__ __launch_bounds__(256) void film_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ res, float* __restrict__ y, int n4_per_sample, int C4, float alpha) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * n4_per_sample;
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + base;
    const f32x4* R = res ? reinterpret_cast<const f32x4*>(res) + base : nullptr;
    f32x4* Y = reinterpret_cast<f32x4*>(y) + base;
    const f32x4* G = reinterpret_cast<const f32x4*>(gamma) + (size_t)b * C4;
    const f32x4* Bt = reinterpret_cast<const f32x4*>(beta) + (size_t)b * C4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4_per_sample; i += gridDim.x * 256) {
        const int c4 = i % C4;
        const f32x4 xv = X[i], ga = G[c4], be = Bt[c4];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = xv[e] * ga[e] + be[e]; v[e] = t >= 0.f ? t : t * alpha; }
        if (R) v += R[i];
        Y[i] = v;
    }
}

// backward of u = leaky(x*gamma+beta): given du, x, gamma, beta ->
//   dx = g*gamma,  part_dgamma[blk][b][c] = sum g*x,  part_dbeta = sum g   with g = du * leaky'(x*gamma+beta)
__global__ void film_bwd_kernel(const float* __restrict__ du, const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ dx, float* __restrict__ part,
                                int B, long HW, int C, float alpha, int nchunk) {
    // grid = (nchunk, B); block 256 threads; C must divide 256
    extern __shared__ float sm[];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % C, pl = tid / C, npl = blockDim.x / C;
    const long per = (HW + nchunk - 1) / nchunk;
    const long p0 = (long)blockIdx.x * per, p1 = min(HW, p0 + per);
    const float ga = gamma[b * C + c], be = beta[b * C + c];
    float sg = 0.f, sb = 0.f;
    for (long px = p0 + pl; px < p1; px += npl) {
        const long i = ((long)b * HW + px) * C + c;
        const float xv = x[i];
        const float pre = xv * ga + be;
        const float g = du[i] * (pre >= 0.f ? 1.f : alpha);
        dx[i] = g * ga;
        sg += g * xv; sb += g;
    }
    sm[tid] = sg; sm[blockDim.x + tid] = sb;
    __syncthreads();
    if (tid < C) {
        float tg = 0.f, tb = 0.f;
        for (int k = 0; k < npl; ++k) { tg += sm[k * C + tid]; tb += sm[blockDim.x + k * C + tid]; }
        float* o = part + ((size_t)blockIdx.x * B + b) * 2 * C;
        o[tid] = tg; o[C + tid] = tb;
    }
}

// the same with one float4 of channels per thread (C % 4 == 0, 256 % (C / 4) == 0, 16-byte aligned tensors); the block's partial sums
// meet in LDS and are folded by a fixed-order tree over the pixel lanes
__global__ __launch_bounds__(256) void film_bwd_v4_kernel(const float* __restrict__ du, const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ dx, float* __restrict__ part,
                                                          int B, int HW, int C, float alpha, int nchunk) {
    __shared__ f32x4 sm[2][256];
    const int b = blockIdx.y, tid = threadIdx.x, C4 = C >> 2;
    const int c4 = tid % C4, pl = tid / C4, npl = 256 / C4;
    const int per = (HW + nchunk - 1) / nchunk;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[b * C4 + c4], be = reinterpret_cast<const f32x4*>(beta)[b * C4 + c4];
    const size_t base = (size_t)b * HW * C4;
    const f32x4* X = reinterpret_cast<const f32x4*>(x) + base;
    const f32x4* DU = reinterpret_cast<const f32x4*>(du) + base;
    f32x4* DX = reinterpret_cast<f32x4*>(dx) + base;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int px = p0 + pl; px < p1; px += npl) {
        const int i = px * C4 + c4;
        const f32x4 xv = X[i], d = DU[i];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float pre = xv[e] * ga[e] + be[e]; g[e] = d[e] * (pre >= 0.f ? 1.f : alpha); }
        DX[i] = g * ga;
        sg += g * xv; sb += g;
    }
    sm[0][tid] = sg; sm[1][tid] = sb;
    __syncthreads();
    for (int st = npl >> 1; st >= 1; st >>= 1) {             // npl is a power of two (256 / C4 with C4 a power of two) or handled below
        if (pl < st) { sm[0][tid] += sm[0][tid + st * C4]; sm[1][tid] += sm[1][tid + st * C4]; }
        __syncthreads();
    }
    if (tid < C4) {
        float* o = part + ((size_t)blockIdx.x * B + b) * 2 * C;
        *reinterpret_cast<f32x4*>(o + 4 * tid) = sm[0][tid];
        *reinterpret_cast<f32x4*>(o + C + 4 * tid) = sm[1][tid];
    }
}

// 16 lanes per (b, c) walk the chunk partials (8 each at 128 chunks), then a fixed-order sum over the lanes; block 256 = 16 pairs
__global__ __launch_bounds__(256) void film_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int B, int C, int nchunk) {
    __shared__ float sm[2][16][17];
    const int pl = threadIdx.x & 15, ln = threadIdx.x >> 4;          // consecutive threads = consecutive channels: coalesced reads
    const int i = blockIdx.x * 16 + pl;
    float tg = 0.f, tb = 0.f;
    if (i < B * C) {
        const int b = i / C, c = i % C;
#pragma unroll 4
        for (int k = ln; k < nchunk; k += 16) {
            const float* o = part + ((size_t)k * B + b) * 2 * C;
            tg += o[c]; tb += o[C + c];
        }
    }
    sm[0][ln][pl] = tg; sm[1][ln][pl] = tb;
    __syncthreads();
    if (ln == 0 && i < B * C) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) { g += sm[0][l][pl]; bsum += sm[1][l][pl]; }
        dgamma[i] = g; dbeta[i] = bsum;
    }
}

// ---- element-wise maximum (tf.maximum gradient rule: ties go to the FIRST argument) -----------------------
__global__ void maximum_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = fmaxf(a[i], b[i]);
}
__global__ void maximum_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dy,
                                   float* __restrict__ da, float* __restrict__ db, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const bool first = a[i] >= b[i];
        const float g = dy[i];
        if (da) da[i] = first ? g : 0.f;
        if (db) db[i] = first ? 0.f : g;
    }
}

// ---- channel slice copy: y[m, 0:Cs] = x[m, c0:c0+Cs]; backward scatters into a zeroed tensor ---------------
__global__ void slice_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long M, int C, int c0, int Cs) {
    const long n = M * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Cs; const int c = i % Cs;
        y[i] = x[m * C + c0 + c];
    }
}
__global__ void slice_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long M, int C, int c0, int Cs) {
    const long n = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C; const int c = i % C;
        dx[i] = (c >= c0 && c < c0 + Cs) ? dy[m * Cs + (c - c0)] : 0.f;
    }
}

// ---- VAE sampling + KL (tiny: B x Z) ------------------------------------------------------------------------
__global__ void sampling_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                       float* __restrict__ z, float* __restrict__ kl, int B, int Z) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc = 0.f;
    for (int j = 0; j < Z; ++j) {
        const float m = mu[b * Z + j], l = lv[b * Z + j];
        z[b * Z + j] = m + expf(0.5f * l) * eps[b * Z + j];
        acc += 1.f + l - m * m - expf(l);
    }
    kl[b] = -0.5f * acc;
}
// dmu = dz + dkl*mu ; dlv = dz*0.5*exp(0.5 lv)*eps + dkl*(-0.5)(1 - exp(lv))
__global__ void sampling_kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                       const float* __restrict__ dz, const float* __restrict__ dkl, float* __restrict__ dmu,
                                       float* __restrict__ dlv, int B, int Z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Z) return;
    const int b = i / Z;
    const float m = mu[i], l = lv[i];
    const float gz = dz ? dz[i] : 0.f, gk = dkl ? dkl[b] : 0.f;
    dmu[i] = gz + gk * m;
    dlv[i] = gz * 0.5f * expf(0.5f * l) * eps[i] + gk * (-0.5f) * (1.f - expf(l));
}

// ---- nearest resampling ------------------------------------------------------------------------------------
// keras UpSampling2D(2) as a standalone op (SPADE decoder, decoder.py:70-80): y[B,2H,2W,C]
__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4) {
    const long n = (long)B * 2 * H * 2 * W * C4;
    const f32x4* X = reinterpret_cast<const f32x4*>(x);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C4; long r = i / C4;
        const int w = r % (2 * W); r /= (2 * W);
        const int h = r % (2 * H); const int b = r / (2 * H);
        reinterpret_cast<f32x4*>(y)[i] = X[(((long)b * H + (h >> 1)) * W + (w >> 1)) * C4 + c];
    }
}
// tf.image.resize_nearest_neighbor(align_corners=False) for an integer down-sampling factor f (layers/spade.py:36-38):
// y[b, i, j, :] = x[b, i*f, j*f, :]
__global__ void subsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Ho, int Wo, int C, int f) {
    const long n = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C; long r = i / C;
        const int w = r % Wo; r /= Wo;
        const int h = r % Ho; const int b = r / Ho;
        y[i] = x[(((long)b * Ho * f + (long)h * f) * Wo * f + (long)w * f) * C + c];
    }
}
// dx[B, Ho*f, Wo*f, C]: dy at the sampled positions, 0 elsewhere
__global__ void subsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int Ho, int Wo, int C, int f) {
    const int H = Ho * f, W = Wo * f;
    const long n = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C; long r = i / C;
        const int w = r % W; r /= W;
        const int h = r % H; const int b = r / H;
        dx[i] = (h % f == 0 && w % f == 0) ? dy[(((long)b * Ho + h / f) * Wo + w / f) * C + c] : 0.f;
    }
}

// ---- tap sum: second half of the data gradient of a stride-1 convolution with FEW input channels ------------------------------
// dx[b, h, w, ci] = sum_{kh, kw} T[b, h + ph - kh, w + pw - kw, (kh*KW + kw)*Cin + ci]   (terms outside the Ho x Wo plane are zero)
// where T[q, tap*Cin + ci] = sum_co dy[q, co] * W[tap, ci, co] is ONE 1x1 convolution (a plain GEMM with K = Cout, N = taps*Cin)
// over the incoming gradient.  Run as an N = Cin implicit GEMM the same data gradient pads N to the 32-wide MFMA tile and
// re-reads dy once per tap (segmentor c0, 64 -> 8 channels at 256 x 256: 0.55 ms); this way dy is read once and T (taps*Cin
// floats per pixel) is written and read once.  One thread per (pixel, channel quad).
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ T, float* __restrict__ dx, int B, int H, int W, int Ho, int Wo,
                                                     int Cin, int KH, int KW, int ph, int pw) {
    const int CQ = Cin >> 2, NTC = KH * KW * Cin;
    const long n = (long)B * H * W * CQ;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int cq = i % CQ; long r = i / CQ;
        const int w = r % W; r /= W;
        const int h = r % H; const int b = r / H;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < KH; ++kh) {
            const int ho = h + ph - kh;
            if ((unsigned)ho >= (unsigned)Ho) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int wo = w + pw - kw;
                if ((unsigned)wo >= (unsigned)Wo) continue;
                acc += *reinterpret_cast<const f32x4*>(T + (((size_t)b * Ho + ho) * Wo + wo) * NTC + (kh * KW + kw) * Cin + 4 * cq);
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}

// out[m][0:Ca] = a[m][:], out[m][Ca:Ca+Cb] = b[m][:]  -- two Keras kernels / biases side by side as ONE operand with Ca + Cb output
// channels (the gamma and beta convolutions of a SPADE unit run as one convolution, layers/spade.py:30-31 of the reference)
__global__ void concat_cols_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long M, int Ca, int Cb) {
    const int C = Ca + Cb;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        out[i] = c < Ca ? a[m * Ca + c] : b[m * Cb + (c - Ca)];
    }
}
// da[m][:] += src[m][0:Ca], db[m][:] += src[m][Ca:Ca+Cb]  -- the fused weight / bias gradient back into the two gradient-arena views
__global__ void split_cols_acc_kernel(const float* __restrict__ src, float* __restrict__ da, float* __restrict__ db, long M, int Ca, int Cb) {
    const int C = Ca + Cb;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        if (c < Ca) da[m * Ca + c] += src[i]; else db[m * Cb + (c - Ca)] += src[i];
    }
}

extern "C" {

int mmseg_act_fwd(const float* x, float* y, long n, int act, float alpha, void* stream) {
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, act, alpha);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float alpha, void* stream) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n / 4, n, act, alpha);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_concat_cols(const float* a, const float* b, float* out, long M, int Ca, int Cb, void* stream) {
    if (M <= 0 || Ca <= 0 || Cb <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(concat_cols_kernel, dim3(grid_for(M * (Ca + Cb), 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, M, Ca, Cb);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_split_cols_acc(const float* src, float* da, float* db, long M, int Ca, int Cb, void* stream) {
    if (M <= 0 || Ca <= 0 || Cb <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(split_cols_acc_kernel, dim3(grid_for(M * (Ca + Cb), 256)), dim3(256), 0, (hipStream_t)stream, src, da, db, M, Ca, Cb);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_axpby(const float* a, const float* b, float* out, long n, float sa, float sb, void* stream) {
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, n, sa, sb);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_fill(float* x, long n, float v, void* stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, v);
    return MMSEG_CHECK_LAUNCH();
}

// workspace: nblk * C floats, nblk = mmseg_colsum_blocks(M)
int mmseg_colsum_blocks(long M) {
    long nb = (M + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}
// workspace floats for any path of mmseg_colsum: <= 512 row blocks x max(C, 64) pseudo-channels, or 1024 x C
long mmseg_colsum_workspace_floats(long M, int C) { (void)M; const long a = 1024L * C; return a > 65536 ? a : 65536; }
int mmseg_colsum(const float* x, float* out, float* ws, long M, int C, float scale, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    long M2 = 0; int C2 = 0;
    if (aligned && (C & 63) == 0) { M2 = M; C2 = C; }
    else if (aligned && C <= 32 && 64 % C == 0 && (M * C) % 64 == 0) { M2 = M * C / 64; C2 = 64; }
    if (C2) {
        long nb = 1024 / (C2 / 64);
        if (nb > 512) nb = 512;
        const long maxb = (M2 + 63) / 64;
        if (nb > maxb) nb = maxb;
        if (nb < 1) nb = 1;
        const long rpb = (M2 + nb - 1) / nb;
        hipLaunchKernelGGL(colsum_v4_partial_kernel, dim3((unsigned)nb, C2 / 64), dim3(256), 0, st, x, ws, M2, C2, rpb);
        hipLaunchKernelGGL(colsum_fold_final_kernel, dim3(C), dim3(256), 0, st, (const float*)ws, out, (int)nb, C, C2, scale, accumulate);
        return MMSEG_CHECK_LAUNCH();
    }
    const int nblk = mmseg_colsum_blocks(M);
    const long rpb = (M + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 256 * sizeof(float), st, x, ws, M, C, rpb);
    hipLaunchKernelGGL(colsum_fold_final_kernel, dim3(C), dim3(256), 0, st, (const float*)ws, out, nblk, C, C, scale, accumulate);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
    const long n = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, B, H, W, C / 4);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_upsample2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C / 4);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_upsample2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if (C & 3) return (int)hipErrorInvalidValue;
    const long n = (long)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_subsample_fwd(const float* x, float* y, int B, int Ho, int Wo, int C, int f, void* stream) {
    if (f < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(subsample_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, Ho, Wo, C, f);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_subsample_bwd(const float* dy, float* dx, int B, int Ho, int Wo, int C, int f, void* stream) {
    if (f < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(subsample_bwd_kernel, dim3(grid_for((long)B * Ho * f * Wo * f * C, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, Ho, Wo, C, f);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_softmax_fwd(const float* x, float* p, float* s, long npix, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid_for(npix, 256)), b(256);
    switch (C) {
        case 3: hipLaunchKernelGGL(softmax_fwd_kernel<3>, g, b, 0, st, x, p, s, npix); break;
        case 5: hipLaunchKernelGGL(softmax_fwd_kernel<5>, g, b, 0, st, x, p, s, npix); break;
        case 8: hipLaunchKernelGGL(softmax_fwd_kernel<8>, g, b, 0, st, x, p, s, npix); break;
        default: return (int)hipErrorInvalidValue;
    }
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_softmax_bwd(const float* dy, const float* p, float* dx, long npix, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid_for(npix, 256)), b(256);
    switch (C) {
        case 3: hipLaunchKernelGGL(softmax_bwd_kernel<3>, g, b, 0, st, dy, p, dx, npix); break;
        case 5: hipLaunchKernelGGL(softmax_bwd_kernel<5>, g, b, 0, st, dy, p, dx, npix); break;
        case 8: hipLaunchKernelGGL(softmax_bwd_kernel<8>, g, b, 0, st, dy, p, dx, npix); break;
        default: return (int)hipErrorInvalidValue;
    }
    return MMSEG_CHECK_LAUNCH();
}

static inline bool film_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int mmseg_film_fwd(const float* x, const float* gamma, const float* beta, const float* res, float* y, int B, long HW, int C,
                   float alpha, void* stream) {
    const long n = (long)B * HW * C;
    const long n4s = HW * (C / 4);
    if (C % 4 == 0 && n4s < (1L << 30) && B <= 65535 && film_al16(x) && film_al16(y) && film_al16(gamma) && film_al16(beta) && (res == nullptr || film_al16(res))) {
        long gx = (n4s + 255) / 256; if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(film_fwd_v4_kernel, dim3((unsigned)gx, B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, res, y, (int)n4s, C / 4, alpha);
        return MMSEG_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(film_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, res, y, B, HW, C, alpha);
    return MMSEG_CHECK_LAUNCH();
}
#define FILM_CHUNKS 128
int mmseg_film_bwd_workspace(int B, int C) { return FILM_CHUNKS * B * 2 * C; }
int mmseg_film_bwd(const float* du, const float* x, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                   float* ws, int B, long HW, int C, float alpha, void* stream) {
    if (C > 256 || 256 % C != 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int C4 = C / 4;
    if (C % 4 == 0 && (C4 & (C4 - 1)) == 0 && HW * (long)C4 < (1L << 30) && film_al16(du) && film_al16(x) && film_al16(dx) && film_al16(gamma) &&
        film_al16(beta) && film_al16(ws))
        hipLaunchKernelGGL(film_bwd_v4_kernel, dim3(FILM_CHUNKS, B), dim3(256), 0, st, du, x, gamma, beta, dx, ws, B, (int)HW, C, alpha, FILM_CHUNKS);
    else
        hipLaunchKernelGGL(film_bwd_kernel, dim3(FILM_CHUNKS, B), dim3(256), 2 * 256 * sizeof(float), st, du, x, gamma, beta, dx, ws, B, HW, C, alpha, FILM_CHUNKS);
    hipLaunchKernelGGL(film_bwd_final_kernel, dim3((B * C + 15) / 16), dim3(256), 0, st, (const float*)ws, dgamma, dbeta, B, C, FILM_CHUNKS);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_maximum_fwd(const float* a, const float* b, float* y, long n, void* stream) {
    hipLaunchKernelGGL(maximum_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_maximum_bwd(const float* a, const float* b, const float* dy, float* da, float* db, long n, void* stream) {
    hipLaunchKernelGGL(maximum_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, dy, da, db, n);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_slice_fwd(const float* x, float* y, long M, int C, int c0, int Cs, void* stream) {
    if (c0 < 0 || c0 + Cs > C) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slice_fwd_kernel, dim3(grid_for(M * Cs, 256)), dim3(256), 0, (hipStream_t)stream, x, y, M, C, c0, Cs);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_slice_bwd(const float* dy, float* dx, long M, int C, int c0, int Cs, void* stream) {
    if (c0 < 0 || c0 + Cs > C) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(slice_bwd_kernel, dim3(grid_for(M * C, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, M, C, c0, Cs);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_sampling_kl_fwd(const float* mu, const float* lv, const float* eps, float* z, float* kl, int B, int Z, void* stream) {
    hipLaunchKernelGGL(sampling_kl_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, mu, lv, eps, z, kl, B, Z);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_sampling_kl_bwd(const float* mu, const float* lv, const float* eps, const float* dz, const float* dkl, float* dmu,
                          float* dlv, int B, int Z, void* stream) {
    hipLaunchKernelGGL(sampling_kl_bwd_kernel, dim3((B * Z + 63) / 64), dim3(64), 0, (hipStream_t)stream, mu, lv, eps, dz, dkl, dmu, dlv, B, Z);
    return MMSEG_CHECK_LAUNCH();
}


// dx [B,H,W,Cin] from T [B,Ho,Wo,KH*KW*Cin] (see tapsum_kernel); Cin % 4 == 0, both 16-byte aligned
int mmseg_conv2d_dgrad_tapsum(const float* T, float* dx, int B, int H, int W, int Ho, int Wo, int Cin, int KH, int KW, int ph, int pw,
                              void* stream) {
    if ((Cin & 3) || B <= 0 || H <= 0 || W <= 0 || (reinterpret_cast<uintptr_t>(T) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15))
        return (int)hipErrorInvalidValue;
    const long n = (long)B * H * W * (Cin / 4);
    hipLaunchKernelGGL(tapsum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, T, dx, B, H, W, Ho, Wo, Cin, KH, KW, ph, pw);
    return MMSEG_CHECK_LAUNCH();
}
}  // extern "C"
