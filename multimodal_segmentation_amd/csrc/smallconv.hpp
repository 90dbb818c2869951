// HBM-bound convolutions with a tiny channel count on one side (included by conv.hip; fp32 tensors only).
//
// An MFMA tile is >= 32 wide in N and 32 deep in K, so a 1x1 head with 5 or 8 (or 1) output channels, or a first layer with ONE
// input channel, runs the matrix cores on 75-97 % padding -- the generic implicit-GEMM kernel reached 0.2-10 TFLOP/s and 8 % of
// the HBM rate on them (round-2 review: conv_fwd_kernel<128, 32, 4, 1, true, 0> at 0.63 TB/s).  These layers move 17-134 MB per
// launch and do <= 36 FMAs per 16 bytes: they are bandwidth kernels, written as such here -- coalesced 16-byte accesses along the
// channel axis of the wide side, the narrow side and the weights in registers, lane-group reductions by DPP.
//
//   "reduce" type  (Cin = 4 * LANES wide, Cout <= 8):  pw_reduce_kernel        forward of the 1x1 heads (64 -> 8 anatomy head,
//                   64 -> 5 segmentor head, 8 -> 1 / 16 -> 1 decoder heads)     model_components/anatomy_encoder.py:23,
//                                                      pw_reduce_wgrad_kernel   segmentor.py:22, decoder.py:28 of the reference
//   "expand" type  (K = KS*KS*Cin <= 16, Cout % 4 == 0): smallk_conv_kernel    their data gradients (5 -> 64, 8 -> 64, 1 -> 8 as 1x1
//                                                                               convolutions with the transposed kernel) and the
//                                                                               FIRST layers with one input channel: UNet d_l0.a
//                                                                               3x3 (models/unet.py:40), D_Image 4x4 stride 2
//                                                      smallk_wgrad_kernel      (models/discriminator.py:24)
// Every reduction runs in a fixed order (deterministic, like the rest of the library).
#pragma once

// reduced-precision modes (mmseg_set_conv_precision): these layers multiply in fp32, but on operands rounded to the 16-bit type exactly
// where the MFMA kernels they replace rounded them (input channels a multiple of 4) -- same products, fp32 accumulation
__device__ __forceinline__ float round16(float v, int prec) {
    return prec == 1 ? (float)(__bf16)v : (prec == 2 ? (float)(_Float16)v : v);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the LANES consecutive lanes of a group (LANES in {2, 4, 8, 16}, groups aligned to LANES); every lane of the group gets the sum
template <int LANES>
__device__ __forceinline__ float group_allsum(float v) {
    if (LANES >= 2 && LANES <= 8) v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    if (LANES >= 4 && LANES <= 8) v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
    if (LANES == 8) v += dpp_mov<0x141>(v);                     // row_half_mirror: lane i <-> 7 - i, i.e. the other quad of the 8
    if (LANES == 16) {                                           // rotations inside the row of 16 lanes
        v += dpp_mov<0x128>(v);
        v += dpp_mov<0x124>(v);
        v += dpp_mov<0x122>(v);
        v += dpp_mov<0x121>(v);
    }
    return v;
}

// y[p, j] = act( (sum_c x[p, c] * w[c, j]) * oscale[j] + bias[j] ),  Cin = 4 * LANES * VPL, j < COUT <= LANES.
// LANES lanes share a pixel (VPL 16-byte loads each, 16 * LANES consecutive bytes per load instruction and pixel), lane q < COUT of
// the group stores output channel q (consecutive pixels -> consecutive addresses).
template <int LANES, int COUT, int VPL = 1>
__global__ __launch_bounds__(256) void pw_reduce_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ oscale,
                                                        float* __restrict__ y, long M, int act, float alpha, int prec) {
    static_assert(COUT <= LANES && COUT <= 8, "one output channel per lane of the group");
    constexpr int CIN = 4 * LANES * VPL, PPW = 64 / LANES;
    const int lane = threadIdx.x & 63, q = lane % LANES, g = lane / LANES;
    float wr[VPL][4][COUT];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* row = w + (4 * (q + v * LANES) + e) * COUT;          // COUT consecutive floats
            if (COUT % 4 == 0) {
#pragma unroll
                for (int j = 0; j < COUT; j += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(row + j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) wr[v][e][j + i] = round16(t[i], prec);
                }
            } else {
#pragma unroll
                for (int j = 0; j < COUT; ++j) wr[v][e][j] = round16(row[j], prec);
            }
        }
    const float bj = (bias != nullptr && q < COUT) ? bias[q] : 0.f;
    const float sj = (oscale != nullptr && q < COUT) ? oscale[q] : 1.f;
    const long nw = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int UN = 4 / VPL;                               // pixel groups per iteration: 4 independent 16-byte loads in flight per lane
    for (long p0 = w0 * PPW * UN; p0 < M; p0 += nw * PPW * UN) {
        f32x4 xv[UN][VPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long pix = p0 + u * PPW + g;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                xv[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (pix < M) xv[u][v] = *reinterpret_cast<const f32x4*>(x + pix * CIN + 4 * (q + v * LANES));
            }
        }
        if (prec) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int v = 0; v < VPL; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[u][v][e] = round16(xv[u][v][e], prec);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long pix = p0 + u * PPW + g;
            float out = 0.f;
#pragma unroll
            for (int j = 0; j < COUT; ++j) {
                float a = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fmaf(xv[u][v][e], wr[v][e][j], a);
                a = group_allsum<LANES>(a);
                out = (q == j) ? a : out;
            }
            if (pix < M && q < COUT) y[pix * COUT + q] = act_apply(fmaf(out, sj, bj), act, alpha);
        }
    }
}

// slab[blockIdx.x][c, j] = sum over this block's pixels of x[p, c] * dy[p, j]   (Cin = 4 * LANES * VPL, j < COUT); the caller reduces the slabs.
// 4 pixel groups per iteration (4 * VPL independent 16-byte loads of x in flight per lane); the block's 256 / LANES partial sums of
// every (c, j) meet in LDS and are added in a fixed order.
template <int LANES, int COUT, int VPL = 1>
__global__ __launch_bounds__(256) void pw_reduce_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ ws, long M, int prec) {
    constexpr int CIN = 4 * LANES * VPL, PPW = 64 / LANES, KN = CIN * COUT, NG = 256 / LANES;   // NG pixel groups per block
    extern __shared__ float smw[];                           // [NG][KN]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane % LANES, g = lane / LANES;
    float acc[VPL][4][COUT];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < COUT; ++j) acc[v][e][j] = 0.f;
    const long nw = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + wave;
    constexpr int UN = 4;
    for (long p0 = w0 * PPW * UN; p0 < M; p0 += nw * PPW * UN) {
        f32x4 xv[UN][VPL];
        float d[UN][COUT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long pix = p0 + u * PPW + g;
#pragma unroll
            for (int v = 0; v < VPL; ++v) xv[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < COUT; ++j) d[u][j] = 0.f;
            if (pix < M) {
#pragma unroll
                for (int v = 0; v < VPL; ++v) xv[u][v] = *reinterpret_cast<const f32x4*>(x + pix * CIN + 4 * (q + v * LANES));
                if (COUT % 4 == 0) {
#pragma unroll
                    for (int j = 0; j < COUT; j += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(dy + pix * COUT + j);
#pragma unroll
                        for (int i = 0; i < 4; ++i) d[u][j + i] = t[i];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < COUT; ++j) d[u][j] = dy[pix * COUT + j];
                }
            }
        }
        if (prec) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
#pragma unroll
                for (int v = 0; v < VPL; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[u][v][e] = round16(xv[u][v][e], prec);
#pragma unroll
                for (int j = 0; j < COUT; ++j) d[u][j] = round16(d[u][j], prec);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < COUT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[v][e][j] = fmaf(xv[u][v][e], d[u][j], acc[v][e][j]);
    }
    const int grp = wave * PPW + g;                          // this lane's pixel group inside the block
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < COUT; ++j) smw[(size_t)grp * KN + (4 * (q + v * LANES) + e) * COUT + j] = acc[v][e][j];
    __syncthreads();
    for (int i = threadIdx.x; i < KN; i += 256) {
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < NG; ++k) a += smw[(size_t)k * KN + i];
        ws[(size_t)blockIdx.x * KN + i] = a;
    }
}

// The K = KS*KS*CIN input values of output pixel `pix` for a group of L lanes (L = Cout / 4) that share the pixel: value k is fetched
// ONCE, by lane k % L of the group (ceil(K / L) loads per lane instead of K), and handed to the other lanes through the LDS crossbar
// (ds_bpermute_b32, no LDS memory).  Padding taps read as zeros.  All lanes of a group call this together.
template <int KS, int CIN, int L>
__device__ __forceinline__ void smallk_gather(const ConvParams& p, long pix, int q, int lane, int prec, float (&xv)[KS * KS * CIN]) {
    constexpr int K = KS * KS * CIN, NL = (K + L - 1) / L;
    const int HoWo = p.Ho * p.Wo;
    const int b = (int)(pix / HoWo), r = (int)(pix - (long)b * HoWo);
    const int ho = r / p.Wo, wo = r - ho * p.Wo;
    const int hb = ho * p.stride - p.pad_h, wb = wo * p.stride - p.pad_w;
    float mine[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int k = q + i * L;
        const int tap = k / CIN, c = k - tap * CIN, kh = tap / KS, kw = tap - kh * KS;
        const int hi = hb + kh, wi = wb + kw;
        const bool ok = k < K && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        mine[i] = ok ? round16(p.x1[(((size_t)b * p.H + hi) * p.W + wi) * CIN + c], prec) : 0.f;
    }
    const int base = (lane & ~(L - 1)) * 4;
#pragma unroll
    for (int k = 0; k < K; ++k)
        xv[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base + (k % L) * 4, __builtin_bit_cast(int, mine[k / L])));
}

// KS x KS convolution with K = KS*KS*CIN <= 16 gathered values per pixel and Cout = 4 * L: L lanes share an output pixel, each
// owning 4 consecutive output channels (its K x 4 weights stay in registers; one 16-byte store per lane, 16 * L consecutive bytes
// per pixel).  Strided / padded geometry as the generic kernel (no fused up-sampling, one input tensor).  y = act(conv * oscale + bias).
template <int KS, int CIN, int L>
__global__ __launch_bounds__(256) void smallk_conv_kernel(ConvParams p, int prec) {
    constexpr int K = KS * KS * CIN;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, q = lane % L;
    f32x4 wr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wr[k] = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.Cout + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[k][e] = round16(wr[k][e], prec);
    }
    f32x4 bj = {0.f, 0.f, 0.f, 0.f}, sj = {1.f, 1.f, 1.f, 1.f};
    if (p.bias) bj = *reinterpret_cast<const f32x4*>(p.bias + 4 * q);
    if (p.oscale) sj = *reinterpret_cast<const f32x4*>(p.oscale + 4 * q);
    const long step = ((long)gridDim.x * 256) / L;
    for (long pix = tid / L; pix < p.M; pix += step) {
        float xv[K];
        smallk_gather<KS, CIN, L>(p, pix, q, lane, prec, xv);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) a += xv[k] * wr[k];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_apply(fmaf(a[e], sj[e], bj[e]), p.act, p.alpha);
        *reinterpret_cast<f32x4*>(p.y + pix * p.Cout + 4 * q) = o;
    }
}

// slab[blockIdx.x][k, n] = sum over this block's pixels of xgather[p, k] * dy[p, n]   (same geometry as smallk_conv_kernel)
template <int KS, int CIN, int L>
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(ConvParams p, const float* __restrict__ dy, float* __restrict__ ws) {
    constexpr int K = KS * KS * CIN;
    extern __shared__ float smk[];                           // [4][K * Cout]
    const int KN = K * p.Cout;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane % L;
    f32x4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long step = ((long)gridDim.x * 256) / L;
    for (long pix = tid / L; pix < p.M; pix += step) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + pix * p.Cout + 4 * q);
        float xv[K];
        smallk_gather<KS, CIN, L>(p, pix, q, lane, 0, xv);
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += xv[k] * d;
    }
    // lanes q, q + L, ... of a wave hold different pixels of the same channel quad, then the 4 waves of the block -- fixed order
#pragma unroll
    for (int k = 0; k < K; ++k) {
        f32x4 a = acc[k];
#pragma unroll
        for (int o = L; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], o, 64);
        if (lane < L) *reinterpret_cast<f32x4*>(smk + (size_t)wave * KN + (size_t)k * p.Cout + 4 * q) = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KN; i += 256)
        ws[(size_t)blockIdx.x * KN + i] = (smk[i] + smk[KN + i]) + (smk[2 * KN + i] + smk[3 * KN + i]);
}
