// NHWC convolution family for gfx950 as implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//   forward : y[m, n]  = act( sum_k A[m, k] * W[k, n] + bias[n] )      m = (b, ho, wo), k = (kh, kw, ci)
//   dgrad   : the same kernel on flipped/transposed weights (mmseg_conv2d_wflip), with the
//             fractionally-strided gather (`transposed`) for stride > 1
//   wgrad   : dW[k, n] = sum_m A[m, k] * dy[m, n]  split over pixel chunks into slabs, then a
//             deterministic slab reduction (no float atomics -> bitwise reproducible)
//
// The im2col matrix A is never materialised: each workgroup gathers its [BM x 32] tile from the
// NHWC activation straight into LDS (16-byte loads along the channel axis).  The gather also
// folds nearest x2 up-sampling of the input (keras UpSampling2D before a conv) and the channel
// concatenation of two inputs (keras Concatenate before a conv), so neither is ever written
// to HBM.  Reference ops replaced: keras Conv2D / UpSampling2D / Concatenate as used in
// models/unet.py:37-101, utils/model_utils.py:15-22, model_components/*.py,
// models/discriminator.py:16-41, layers/stn_spline.py:94-120 of the reference.
//
// LDS tile layout: As[m][k] (k contiguous, row stride 36 floats -> conflict-free ds_read_b128),
// Bs[k][n].  The K index inside an 8-wide chunk is permuted (lane half h owns k = 8q+4h+t) so one
// ds_read_b128 feeds four consecutive MFMAs; A and B use the same permutation so the sum is
// unchanged.
#include "common.hpp"
#include <stdio.h>
#include <stdlib.h>

// Kernel-selection switches for A/B measurements (tools/*_bench.py).  The product library compiles the defaults
// in: the environment is only consulted in a build with -DMMSEG_AB (MMSEG_AB_BUILD=1 python -c "... _native.build(force=True)").
static inline int ab_int(const char* name, int dflt) {
#ifdef MMSEG_AB
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

struct ConvParams {
    const float* x1;
    const float* x2;
    const float* w;
    const float* wt;    // fast path only: the same kernel as [Cout][K] (K-contiguous rows), see mmseg_conv2d_wprep
    const float* bias;
    const float* oscale;   // optional per-output-channel scale applied BEFORE the bias (inference BatchNorm folded in)
    float* y;
    float* y2;      // second output for channel-split epilogue (dgrad of a concat conv) or nullptr
    int B, H, W;    // logical input spatial size (after optional up-sampling)
    int C1, C2;     // channels taken from x1 / x2 (C2 = 0: single input)
    int H1, W1;     // physical spatial size of x1 (H >> ups, W >> ups)
    int Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w;
    int ups;        // x1 is stored at half resolution and nearest-upsampled on the fly
    int transposed; // fractionally strided gather: tap valid iff (ho + kh - pad) % stride == 0
    int act;
    float alpha;
    int M, K;
    int nsplit1;    // channel split point of the epilogue (Cout1) when y2 != nullptr
    // output pixel mapping (fast path): launch pixel (b, ho, wo) is stored at (b, ho*osh + ooh, wo*osw + oow) of a
    // [B, oH, oW, Cout] tensor.  Identity unless the launch is one parity class of a strided data gradient.
    int oH, oW, osh, osw, ooh, oow;
    // 16-bit tensors in HBM (reduced-precision modes only; the `_t` entry points): bit 0 = x1, bit 1 = x2, bit 2 = y / y2 are stored in
    // the 16-bit type of the active precision mode, bit 3 = that type is fp16 (else bf16).  0 = all fp32.
    int io;
    int qepi;       // fast path: the epilogue may use the quad-transposed vector stores (set by the host: alignment, MMSEG_QUAD_EPI)
};

#define BK 32
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// 16-bit MFMA operand types of the reduced-precision modes (mmseg_set_conv_precision): PREC 1 = bf16, 2 = fp16
template <int PREC> struct LowPrec;
template <> struct LowPrec<1> {
    typedef __bf16 T; typedef bf16x4 V4; typedef bf16x8 V8;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct LowPrec<2> {
    typedef _Float16 T; typedef f16x4 V4; typedef f16x8 V8;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct LowPrec<0> { typedef float T; typedef f32x4 V4; typedef f32x4 V8; };   // unused placeholder
#define AS_LD (BK + 4)

// 4 x 4 transpose inside a quad of lanes (tq = lane & 3): a[k] of lane t becomes a[t] of lane k.  Two DPP exchanges: with lane ^ 1 on
// the register pairs (0,1),(2,3), then with lane ^ 2 on (0,2),(1,3).  Used by the epilogues: the MFMA accumulator layout gives a lane
// one output channel of 4 consecutive rows; transposed, the lane owns 4 consecutive channels of one row = one vector store.
__device__ __forceinline__ void quad_transpose4(float (&a)[4], const int tq) {
#pragma unroll
    for (int pp = 0; pp < 4; pp += 2) {
        const float send = (tq & 1) ? a[pp] : a[pp + 1];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
        if (tq & 1) a[pp] = recv; else a[pp + 1] = recv;
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const float send = (tq & 2) ? a[pp] : a[pp + 2];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
        if (tq & 2) a[pp] = recv; else a[pp + 2] = recv;
    }
}
// 4 consecutive output channels of one pixel row, as fp32 or as the 16-bit type selected by ConvParams::io (bit 2: 16-bit, bit 3: fp16)
__device__ __forceinline__ void store4_out(void* base, size_t o, const float (&a)[4], int io) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    if (!(io & 4)) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + o) = f32x4{a[0], a[1], a[2], a[3]};
    else if (io & 8) *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(base) + o) = h4{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3]};
    else *reinterpret_cast<b4*>(reinterpret_cast<__bf16*>(base) + o) = b4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
}

__device__ __forceinline__ f32x4 gather_tap4(const ConvParams& p, int b, int hb, int wb, int kh, int kw, int c) {
    // returns 4 consecutive channels c..c+3 of the (possibly virtual) input at the tap, or zeros
    int hi = hb + kh, wi = wb + kw;
    if (p.transposed) {
        if (hi < 0 || wi < 0) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int s = p.stride;
        const int hq = hi / s, wq = wi / s;
        if (hq * s != hi || wq * s != wi) return f32x4{0.f, 0.f, 0.f, 0.f};
        hi = hq; wi = wq;
    }
    if ((unsigned)hi >= (unsigned)p.H || (unsigned)wi >= (unsigned)p.W) return f32x4{0.f, 0.f, 0.f, 0.f};
    const float* src;
    if (c < p.C1) {
        const int h1 = p.ups ? (hi >> 1) : hi, w1 = p.ups ? (wi >> 1) : wi;
        src = p.x1 + (((size_t)b * p.H1 + h1) * p.W1 + w1) * p.C1 + c;
    } else {
        src = p.x2 + (((size_t)b * p.H + hi) * p.W + wi) * p.C2 + (c - p.C1);
    }
    return *reinterpret_cast<const f32x4*>(src);
}

__device__ __forceinline__ float gather_tap1(const ConvParams& p, int b, int hb, int wb, int kh, int kw, int c) {
    int hi = hb + kh, wi = wb + kw;
    if (p.transposed) {
        if (hi < 0 || wi < 0) return 0.f;
        const int s = p.stride;
        const int hq = hi / s, wq = wi / s;
        if (hq * s != hi || wq * s != wi) return 0.f;
        hi = hq; wi = wq;
    }
    if ((unsigned)hi >= (unsigned)p.H || (unsigned)wi >= (unsigned)p.W) return 0.f;
    if (c < p.C1) {
        const int h1 = p.ups ? (hi >> 1) : hi, w1 = p.ups ? (wi >> 1) : wi;
        return p.x1[(((size_t)b * p.H1 + h1) * p.W1 + w1) * p.C1 + c];
    }
    return p.x2[(((size_t)b * p.H + hi) * p.W + wi) * p.C2 + (c - p.C1)];
}

// gather 4 consecutive k (k..k+3) of im2col row (b, hb, wb)
template <bool VEC>
__device__ __forceinline__ f32x4 gather_a(const ConvParams& p, int b, int hb, int wb, int k, int tap, int c) {
    if (VEC) {
        if (k >= p.K) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        return gather_tap4(p, b, hb, wb, kh, kw, c);
    } else {
        f32x4 v;
        const int Cin = p.C1 + p.C2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ke = k + e;
            float t = 0.f;
            if (ke < p.K) {
                const int tp = ke / Cin, ce = ke - tp * Cin;
                const int kh = tp / p.KW, kw = tp - kh * p.KW;
                t = gather_tap1(p, b, hb, wb, kh, kw, ce);
            }
            v[e] = t;
        }
        return v;
    }
}

// PREC != 0 (reduced-precision modes, forward launches with 4-channel gathers): the fp32 LDS tiles are unchanged, the operands are
// rounded to the 16-bit type as they are read and multiplied on v_mfma_f32_32x32x16_* -- a 32-deep K tile is 2 MFMAs of 32 cycles
// per output tile instead of 16 of 64, which is what the small-K layers of the SPADE units (8 -> 128, K = 72) are bound by in fp32
template <int BM, int BN, int WM, int WN, bool VEC, int PREC = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_fwd_kernel(ConvParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS_PER_PASS = NT / 8;
    constexpr int A_F4 = BM / A_ROWS_PER_PASS;
    constexpr int BF4_PER_ROW = BN / 4;
    constexpr int B_ROWS_PER_PASS = NT / BF4_PER_ROW;
    constexpr int B_F4 = (BK + B_ROWS_PER_PASS - 1) / B_ROWS_PER_PASS;
    constexpr int BS_LD = BN + 4;
    static_assert(A_F4 >= 1 && BM % A_ROWS_PER_PASS == 0, "A staging");

    __shared__ __attribute__((aligned(16))) float As[BM * AS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;

    // block -> (m tile, n tile); n fastest so that the blocks of one XCD share the A panel in L2
    const int ntn = (p.Cout + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, nwg);
    const int m0 = (lb / ntn) * BM, n0 = (lb % ntn) * BN;

    // ---- per-thread A gather state -------------------------------------------------------
    const int kc = tid & 7;              // which float4 of the 32-wide K tile
    const int ar0 = tid >> 3;            // first row
    int a_b[A_F4], a_hb[A_F4], a_wb[A_F4];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        const int m = m0 + ar0 + j * A_ROWS_PER_PASS;
        if (m < p.M) {
            const int b = m / HoWo, r = m - b * HoWo;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            a_b[j] = b;
            a_hb[j] = (p.transposed ? ho : ho * p.stride) - p.pad_h;
            a_wb[j] = (p.transposed ? wo : wo * p.stride) - p.pad_w;
        } else {
            a_b[j] = 0; a_hb[j] = -(1 << 28); a_wb[j] = -(1 << 28);  // every tap out of range -> zeros
        }
    }
    const int Cin = p.C1 + p.C2;
    int a_k = 4 * kc, a_tap = 0, a_c = 4 * kc;
    if (VEC) { a_tap = a_k / Cin; a_c = a_k - a_tap * Cin; }

    // ---- per-thread B (weight) load state -------------------------------------------------
    const int b_nc = tid % BF4_PER_ROW, b_r0 = tid / BF4_PER_ROW;
    const bool w_vec = (p.Cout & 3) == 0;

    f32x4 ra[A_F4], rb[B_F4];

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) ra[j] = gather_a<VEC>(p, a_b[j], a_hb[j], a_wb[j], a_k, a_tap, a_c);
        a_k += BK;
        if (VEC) { a_c += BK; while (a_c >= Cin) { a_c -= Cin; ++a_tap; } }
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int kr = b_r0 + j * B_ROWS_PER_PASS;
            const int k = kt * BK + kr, n = n0 + 4 * b_nc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kr < BK && k < p.K) {
                const float* src = p.w + (size_t)k * p.Cout + n;
                if (w_vec) {
                    if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(src);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.Cout) v[e] = src[e];
                }
            }
            rb[j] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_F4; ++j)
            *reinterpret_cast<f32x4*>(&As[(ar0 + j * A_ROWS_PER_PASS) * AS_LD + 4 * kc]) = ra[j];
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            const int kr = b_r0 + j * B_ROWS_PER_PASS;
            if (kr < BK) *reinterpret_cast<f32x4*>(&Bs[kr * BS_LD + 4 * b_nc]) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();

    const int a_row = wm * (BM / WM) + li;
    const int b_col = wn * (BN / WN) + li;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);   // global loads stay in flight under the MFMAs
        if constexpr (PREC != 0) {
            typedef typename LowPrec<PREC>::T LT;
            typedef typename LowPrec<PREC>::V8 LV8;
#pragma unroll
            for (int q = 0; q < 2; ++q) {      // two k-steps of 16; lane half lh supplies k = 16 q + 8 lh + [0, 8)
                const int kb = 16 * q + 8 * lh;
                LV8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(&As[(a_row + i * 32) * AS_LD + kb]);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(&As[(a_row + i * 32) * AS_LD + kb + 4]);
                    a[i] = LV8{(LT)lo[0], (LT)lo[1], (LT)lo[2], (LT)lo[3], (LT)hi[0], (LT)hi[1], (LT)hi[2], (LT)hi[3]};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[j][e] = (LT)Bs[(kb + e) * BS_LD + b_col + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = LowPrec<PREC>::mfma(a[i], b[j], acc[i][j]);
            }
        } else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&As[(a_row + i * 32) * AS_LD + 8 * q + 4 * lh]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(8 * q + 4 * lh + t) * BS_LD + b_col + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + activation, rows of 32 consecutive channels per half-wave ---------
    if (p.qepi && p.Cout % 4 == 0 && (p.y2 == nullptr || p.nsplit1 % 4 == 0)) {      // quad-transposed vector stores (see conv_fast_body)
        const int tq = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + li;
            const bool nok = n < p.Cout;
            const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
            const float sv = (nok && p.oscale) ? p.oscale[n] : 1.f;
            const int nq = n - tq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = act_apply(acc[i][j][4 * g + e] * sv + bv, p.act, p.alpha);
                    quad_transpose4(a, tq);
                    const int m = m0 + wm * (BM / WM) + i * 32 + 8 * g + 4 * lh + tq;
                    if (!nok || m >= p.M) continue;
                    const bool first = p.y2 == nullptr || nq < p.nsplit1;
                    const size_t o = p.y2 == nullptr ? (size_t)m * p.Cout + nq
                                     : (first ? (size_t)m * p.nsplit1 + nq : (size_t)m * (p.Cout - p.nsplit1) + (nq - p.nsplit1));
                    store4_out(first ? (void*)p.y : (void*)p.y2, o, a, p.io);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + li;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        const float sv = p.oscale ? p.oscale[n] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                const float v = act_apply(acc[i][j][r] * sv + bv, p.act, p.alpha);
                if (p.io & 4) {                  // 16-bit output tensor (fp32 arithmetic; the generic kernel only WRITES 16-bit)
                    const size_t o = p.y2 == nullptr ? (size_t)m * p.Cout + n
                                     : (n < p.nsplit1 ? (size_t)m * p.nsplit1 + n : (size_t)m * (p.Cout - p.nsplit1) + (n - p.nsplit1));
                    void* base = (p.y2 == nullptr || n < p.nsplit1) ? (void*)p.y : (void*)p.y2;
                    if (p.io & 8) reinterpret_cast<_Float16*>(base)[o] = (_Float16)v;
                    else reinterpret_cast<__bf16*>(base)[o] = (__bf16)v;
                    continue;
                }
                if (p.y2 == nullptr) p.y[(size_t)m * p.Cout + n] = v;
                else if (n < p.nsplit1) p.y[(size_t)m * p.nsplit1 + n] = v;
                else p.y2[(size_t)m * (p.Cout - p.nsplit1) + (n - p.nsplit1)] = v;
            }
        }
    }
}

// =====================================================================================
// Fast path of the same implicit GEMM for the layers that carry the FLOPs (every conv with Cin % 32 == 0 that is
// not a fractionally-strided data gradient: the UNet, segmentor c1, discriminator blocks, SPADE convs and all
// stride-1 data gradients).  Differences to conv_fwd_kernel:
//   * a 32-wide K tile lies inside ONE filter tap and ONE input tensor, so tap / channel / source bookkeeping is
//     wave-uniform scalar arithmetic done once per K tile instead of per lane and row;
//   * the gather uses buffer loads with 32-bit offsets: a padding tap simply gets an out-of-range offset and the
//     hardware returns zeros -- no divergent branches, ~10 VALU per 16-byte load instead of ~100;
//   * LDS is double buffered: the next tile is written while the current one feeds the MFMAs, one barrier per K tile.
// =====================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define BUF_OOB 0x7ffffff0

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// vector stores of the quad-transposed epilogue need 16-byte (fp32) / 8-byte (16-bit) aligned rows: the host checks the
// base pointers (ConvParams::qepi)
__device__ __forceinline__ bool quad_epilogue_enabled(const ConvParams& p) { return p.qepi != 0; }

template <int BM, int BN, int WM, int WN, int PREC = 0, bool IN16 = false, int KT = BK>
__device__ __forceinline__ void conv_fast_body(const ConvParams& p, const int bid, const int nblk) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(KT == 32 || (KT == 64 && PREC != 0), "K tile: 32 channels, or 64 in the 16-bit modes");
    // LDS tiles: 32 floats per row, NO padding; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7, which makes
    // both the ds_write_b128 (8 lanes = 8 chunks of one row) and the ds_read_b128 (16-lane groups over distinct rows)
    // conflict-free and lets 3 blocks of the 128x64 tile share a CU's 160 KB
    // BF16 variant (mmseg_set_conv_precision(1)): the fp32 operands are rounded to bf16 (RNE) on their way into LDS, the tiles
    // are rows of 32 bf16 = 64 bytes (4 chunks of 16 bytes, chunk index XOR-swizzled with (row >> 2) & 3: conflict-free for
    // the stores and -- over ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} -- the 16-byte operand reads
    // alike; (row >> 1) & 3 left the reads 2-way conflicted), products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
    // KT = 64 (16-bit modes, layers whose channel counts are multiples of 64): rows of 64 elements = 128 bytes, i.e. byte for
    // byte the fp32 tile's geometry and swizzle; a stage then carries twice the MFMAs per barrier, load wait and LDS round trip
    // -- the 16-bit MFMAs finish a 32-deep stage in a quarter of the fp32 time and were waiting on exactly those.
    // LD counts 4-byte words per row.
    constexpr bool BF16 = PREC != 0;         // any 16-bit operand mode (bf16 or fp16): same tile geometry
    typedef typename LowPrec<PREC>::T LT;
    typedef typename LowPrec<PREC>::V4 LV4;
    typedef typename LowPrec<PREC>::V8 LV8;
    constexpr int LD = BF16 ? KT / 2 : KT;                // words per LDS row (16 or 32)
    constexpr int NCH = LD / 4;                           // 16-byte chunks per LDS row (4 or 8)
    constexpr int A_SZ = BM * LD, B_SZ = BN * LD;
    // staging geometry: a lane moves 16 bytes of SOURCE; LPA / LPB lanes cover one row of the activation / weight tile
    constexpr int LPA = KT * (IN16 ? 2 : 4) / 16, LPB = KT * (BF16 ? 2 : 4) / 16;
    constexpr int A_RPP = NT / 8;                         // (rows per pass of the 8-lanes-per-row mapping; kept for the weight tile of the fp32 kernel)

    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];
    float* As = smem;
    float* Bs = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int lb = xcd_remap(bid, nblk);
    // Consecutive logical ids land on one XCD (one L2).  Let them share the BIGGER operand: for the shallow layers that
    // is the pixel panel (all N tiles of one pixel tile run back to back); for the deep layers the kernel matrix is
    // larger than the activations (16x16x1024 -> 1024: 37.7 MB of weights vs 8.4 MB of pixels), so there the pixel tiles
    // of one N tile run back to back and every weight slice is fetched by one XCD instead of by all eight.
    const int ntm = nblk / ntn;
    const bool w_major = (long)p.K * p.Cout > (long)p.M * (p.C1 + p.C2);
    const int mt = w_major ? lb % ntm : lb / ntn;
    const int n0 = (w_major ? lb / ntm : lb % ntn) * BN;
    // 2-D pixel tiles (TH x 16) when the output plane divides evenly: the 9 taps of a tile then re-read an L1-sized
    // halo patch instead of 9 disjoint row segments; otherwise BM consecutive pixels in raster order
    constexpr int TH = BM / 16;
    const bool tile2d = (p.Wo % 16 == 0) && (p.Ho % TH == 0);
    const int tpr = tile2d ? p.Wo / 16 : 1, tpi = tile2d ? (p.Ho / TH) * tpr : 1;
    const int t_b = tile2d ? mt / tpi : 0, t_r = tile2d ? mt - t_b * tpi : 0;
    const int t_y0 = tile2d ? (t_r / tpr) * TH : 0, t_x0 = tile2d ? (t_r % tpr) * 16 : 0;
    const int m0 = mt * BM;
    auto row_to_m = [&](int row) -> int {
        return tile2d ? (t_b * p.Ho + t_y0 + (row >> 4)) * p.Wo + t_x0 + (row & 15) : m0 + row;
    };

    // 16-bit input tensors (reduced-precision modes, ConvParams::io bit 0 -> IN16; x1 and x2 together): elements are 2 bytes and a
    // lane's 16-byte load covers 8 channels, so 4 lanes (not 8) stage a 32-channel row and a pass of the block covers 64 rows (not
    // 32) -- half the load and LDS-store instructions of the fp32-input tile
    constexpr int ES = IN16 ? 2 : 4;                       // bytes per input element
    constexpr int A_RPA = NT / LPA;                        // rows per staging pass of the activation tile ...
    constexpr int A_NP = (BM + A_RPA - 1) / A_RPA;         // ... and passes
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * ES : 0, 0x00020000);
    // the weight image holds MFMA-operand-type elements in the reduced-precision modes (mmseg_conv2d_wprep rounds once)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.K * p.Cout * (BF16 ? 2 : 4), 0x00020000);

    // ---- per-thread row state: element offsets of (b, hb, wb, first channel of the lane's chunk) in x1 / x2 ----------------
    const int kcA = tid % LPA;                         // activation-tile staging: 16-byte source piece of the lane ...
    const int arA = tid / LPA;                         // ... its first row ...
    const int ecA = (IN16 ? 8 : 4) * kcA;              // ... and first channel of its piece inside the K tile
    // a_msk: bit (kh * KW + kw) = that tap of the row lies inside the image (the host keeps KH * KW <= 32 on this path), so the
    // inner loop tests one bit instead of re-deriving (hi, wi) and comparing; a_o1 / a_o2 are BYTE offsets (x1 without up-sampling, x2)
    int a_hb[A_NP], a_wb[A_NP], a_o1[A_NP], a_o2[A_NP];
    unsigned a_msk[A_NP];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_NP; ++j) {
        const int row = arA + j * A_RPA;
        const int m = (BM % A_RPA == 0 || row < BM) ? row_to_m(row) : p.M;
        a_msk[j] = 0u;
        if (m < p.M) {
            // (b, ho, wo) of the row: known without divisions for the 2-D pixel tiles
            int b, ho, wo;
            if (tile2d) { b = t_b; ho = t_y0 + (row >> 4); wo = t_x0 + (row & 15); }
            else { b = m / HoWo; const int r = m - b * HoWo; ho = r / p.Wo; wo = r - ho * p.Wo; }
            const int hb = ho * p.stride - p.pad_h, wb = wo * p.stride - p.pad_w;
            a_hb[j] = hb; a_wb[j] = wb;
            a_o1[j] = p.ups ? b * p.H1 : (((b * p.H + hb) * p.W + wb) * p.C1 + ecA) * ES;
            a_o2[j] = (((b * p.H + hb) * p.W + wb) * p.C2 + ecA) * ES;
            // validity mask = (rows kh inside the image) x (columns kw inside): KH + KW tests instead of KH * KW
            unsigned wm_ = 0u;
            for (int kw = 0; kw < p.KW; ++kw)
                if ((unsigned)(wb + kw) < (unsigned)p.W) wm_ |= 1u << kw;
            for (int kh = 0; kh < p.KH; ++kh)
                if ((unsigned)(hb + kh) < (unsigned)p.H) a_msk[j] |= wm_ << (kh * p.KW);
        } else {
            a_hb[j] = 0; a_wb[j] = 0; a_o1[j] = 0; a_o2[j] = 0;
        }
    }
    const int Cin = p.C1 + p.C2;
    const bool omap = p.osh != 1 || p.osw != 1 || p.ooh != 0 || p.oow != 0 || p.oH != p.Ho || p.oW != p.Wo;
    int s_c0 = 0, s_kh = 0, s_kw = 0;          // wave-uniform position of the next K tile: channel base, tap
    // weight tile: fp32 image = 4 floats per lane; 16-bit image = 8 elements per lane
    constexpr int B_RPP = NT / LPB, B_NP = (BN + B_RPP - 1) / B_RPP;
    const int kcB = tid % LPB, arB = tid / LPB;
    constexpr int EB = BF16 ? 2 : 4;           // bytes per element of the weight image
    constexpr int FAR = 0x40000000;            // added to a byte offset it puts the load out of range (the images stay below 2^30 bytes)
    int b_o[B_NP];                             // byte offset of (n, first k of the lane's chunk) in wt, or FAR for rows beyond Cout / BN
#pragma unroll
    for (int j = 0; j < B_NP; ++j) {
        const int row = arB + j * B_RPP, n = n0 + row;
        b_o[j] = ((BN % B_RPP == 0 || row < BN) && n < p.Cout) ? (n * p.K + (BF16 ? 8 : 4) * kcB) * EB : FAR;
    }

    // a tile in flight as raw bits: 4 floats, or 8 16-bit elements per lane
    struct Stage { u32x4 a[A_NP]; u32x4 b[B_NP]; };
    // 16 bytes at byte offset `off` (4 floats or 8 16-bit elements); out of range -> zeros
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int off, bool ok) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : BUF_OOB, 0, 0);
    };
    // `live` = the tile exists (the two-tile prefetch of the 16-bit kernels issues its loads unconditionally, so that the
    // outstanding-load count the compiler waits on is exact; loads of a tile beyond K get out-of-range offsets = no traffic)
    auto load_tile = [&](Stage& t, const bool live) {
        const int kh = s_kh, kw = s_kw;
        const int tap = kh * p.KW + kw;
        const unsigned tbit = live ? 1u << tap : 0u;                       // this tap's bit of the rows' validity masks
        const int b_koff = live ? (tap * Cin + s_c0) * EB : FAR;           // first K column of the tile, in bytes
        if (s_c0 < p.C1) {
            if (p.ups) {
#pragma unroll
                for (int j = 0; j < A_NP; ++j) {
                    const int hi = a_hb[j] + kh, wi = a_wb[j] + kw;
                    const int off = ((a_o1[j] + (hi >> 1)) * p.W1 + (wi >> 1)) * p.C1 + s_c0 + ecA;
                    t.a[j] = ld4(r1, off * ES, (a_msk[j] & tbit) != 0u);
                }
            } else {
                const int toff = ((kh * p.W + kw) * p.C1 + s_c0) * ES;
#pragma unroll
                for (int j = 0; j < A_NP; ++j) t.a[j] = ld4(r1, a_o1[j] + toff, (a_msk[j] & tbit) != 0u);
            }
        } else {
            const int toff = ((kh * p.W + kw) * p.C2 + (s_c0 - p.C1)) * ES;
#pragma unroll
            for (int j = 0; j < A_NP; ++j) t.a[j] = ld4(r2, a_o2[j] + toff, (a_msk[j] & tbit) != 0u);
        }
#pragma unroll
        for (int j = 0; j < B_NP; ++j) t.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_o[j] + b_koff, 0, 0);
        // advance the uniform K position by one tile: taps fastest inside a 32-channel chunk, so that consecutive
        // tiles re-read (shifted) the same cache lines
        if (++s_kw == p.KW) { s_kw = 0; if (++s_kh == p.KH) { s_kh = 0; s_c0 += KT; } }
    };
    // chunk swizzle of an LDS row: 8-chunk rows (128 bytes) (row >> 1) & 7, 4-chunk rows (64 bytes) (row >> 2) & 3
    auto swz = [&](int row) -> int { return NCH == 8 ? (row >> 1) & 7 : (row >> 2) & 3; };
    auto store_tile = [&](int buf, const Stage& t) {
        float* A = As + buf * A_SZ;
        float* Bt = Bs + buf * B_SZ;
        if constexpr (BF16) {
            if constexpr (IN16) {                // already the MFMA operand type: the lane's 16 bytes are one chunk of its row
#pragma unroll
                for (int j = 0; j < A_NP; ++j) {
                    const int row = arA + j * A_RPA;
                    if (BM % A_RPA == 0 || row < BM) *reinterpret_cast<u32x4*>(&A[row * LD + 4 * (kcA ^ swz(row))]) = t.a[j];
                }
            } else {                             // fp32 input: this thread's 4 consecutive k = half of the 16-byte chunk kcA >> 1
#pragma unroll
                for (int j = 0; j < A_NP; ++j) {
                    const int row = arA + j * A_RPA;
                    const f32x4 f = __builtin_bit_cast(f32x4, t.a[j]);
                    LV4 v = {(LT)f[0], (LT)f[1], (LT)f[2], (LT)f[3]};
                    if (BM % A_RPA == 0 || row < BM) *reinterpret_cast<LV4*>(&A[row * LD + 4 * ((kcA >> 1) ^ swz(row)) + 2 * (kcA & 1)]) = v;
                }
            }
#pragma unroll
            for (int j = 0; j < B_NP; ++j) {
                const int row = arB + j * B_RPP;
                if (BN % B_RPP == 0 || row < BN) *reinterpret_cast<u32x4*>(&Bt[row * LD + 4 * (kcB ^ swz(row))]) = t.b[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_NP; ++j)
                *reinterpret_cast<u32x4*>(&A[(arA + j * A_RPA) * LD + 4 * (kcA ^ swz(arA + j * A_RPA))]) = t.a[j];
#pragma unroll
            for (int j = 0; j < B_NP; ++j) {
                const int row = arB + j * B_RPP;
                if (BN % B_RPP == 0 || row < BN) *reinterpret_cast<u32x4*>(&Bt[row * LD + 4 * (kcB ^ swz(row))]) = t.b[j];
            }
        }
    };

    // A 32x32x2 MFMA takes 16 passes; back-to-back MFMAs into the SAME accumulator stall on the previous result.  Waves
    // with fewer than 4 output tiles therefore keep NACC accumulator sets and rotate over them along K (summed once at the
    // end), so that every wave always has >= 4 independent MFMA chains in flight.
    constexpr int NACC = (TM * TN >= 4) ? 1 : 4 / (TM * TN);
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

    const int nkt = p.K / KT;
    const int a_row = wm * (BM / WM) + li;
    const int b_col = wn * (BN / WN) + li;
    auto mma_tile = [&](int cur) {
        const float* A = As + cur * A_SZ;
        const float* Bt = Bs + cur * B_SZ;
        if constexpr (BF16) {
#pragma unroll
            for (int q = 0; q < KT / 16; ++q) {      // k-steps of 16; lane half lh supplies k = 16 q + 8 lh + [0, 8)
                LV8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const LV8*>(&A[(a_row + i * 32) * LD + 4 * ((2 * q + lh) ^ swz(a_row + i * 32))]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const LV8*>(&Bt[(b_col + j * 32) * LD + 4 * ((2 * q + lh) ^ swz(b_col + j * 32))]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[q % NACC][i][j] = LowPrec<PREC>::mfma(a[i], b[j], acc[q % NACC][i][j]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(&A[(a_row + i * 32) * LD + 4 * ((2 * q + lh) ^ swz(a_row + i * 32))]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(&Bt[(b_col + j * 32) * LD + 4 * ((2 * q + lh) ^ swz(b_col + j * 32))]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[t % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[t % NACC][i][j], 0, 0, 0);
            }
        }
    };

    if constexpr (BF16) {
        // 16-bit MFMAs finish a 32-deep tile in a quarter of the fp32 time, far less than a trip to L2 / HBM: the loads run TWO
        // tiles ahead (two register stages, the loop unrolled by two so that their names are static), LDS stays double buffered
        Stage t0, t1;
        load_tile(t0, true);
        load_tile(t1, 1 < nkt);
        store_tile(0, t0);
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nkt; kt += 2) {              // straight-line body: no early exit between the two halves
            load_tile(t0, kt + 2 < nkt);             // LDS[0] = tile kt, t1 = tile kt + 1 (in flight), t0 <- tile kt + 2
            mma_tile(0);
            store_tile(1, t1);
            __syncthreads();
            load_tile(t1, kt + 3 < nkt);             // LDS[1] = tile kt + 1, t0 = tile kt + 2 (in flight), t1 <- tile kt + 3
            mma_tile(1);
            store_tile(0, t0);                       // (zeros when tile kt + 2 does not exist)
            __syncthreads();
        }
        if (kt < nkt) mma_tile(0);                   // odd tile count: the last tile sits in LDS[0]
    } else {
        Stage t0;
        load_tile(t0, true);
        store_tile(0, t0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) load_tile(t0, true);   // buffer loads in flight under the MFMAs
            mma_tile(cur);
            if (kt + 1 < nkt) store_tile(cur ^ 1, t0);   // the other buffer was last read one iteration ago (barrier below)
            __syncthreads();
        }
    }
#pragma unroll
    for (int s = 1; s < NACC; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][i][j] += acc[s][i][j];

    // ---- epilogue.  The accumulator layout gives a lane ONE output channel (column li) of 16 rows: stored as it is, every store
    // instruction moves one element per lane.  Where the channel count allows, each group of 4 rows x 4 lanes (a quad = 4
    // adjacent channels) is transposed inside the quad with two DPP exchanges, after which a lane owns 4 CONSECUTIVE channels of
    // one row: one 16-byte (fp32) or 8-byte (16-bit) store instead of four -- 4 store instructions per 32 x 32 tile instead of
    // 16, same 128-byte segments.  The 16-bit convolutions with few K tiles were bound by exactly those stores.
    if (p.Cout % 4 == 0 && (p.y2 == nullptr || p.nsplit1 % 4 == 0) && quad_epilogue_enabled(p)) {
        const int tq = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + li;
            const bool nok = n < p.Cout;                 // uniform inside a quad (Cout % 4 == 0)
            const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
            const float sv = (nok && p.oscale) ? p.oscale[n] : 1.f;
            const int nq = n - tq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = act_apply(acc[0][i][j][4 * g + e] * sv + bv, p.act, p.alpha);
                    quad_transpose4(a, tq);
                    // this lane now holds channels nq .. nq + 3 of row (8 g + 4 lh + tq) of the tile
                    int m = row_to_m(wm * (BM / WM) + i * 32 + 8 * g + 4 * lh + tq);
                    if (!nok || m >= p.M) continue;
                    if (omap) {
                        const int ob = m / HoWo, orr = m - ob * HoWo;
                        const int oh = orr / p.Wo, ow = orr - oh * p.Wo;
                        m = (ob * p.oH + oh * p.osh + p.ooh) * p.oW + ow * p.osw + p.oow;
                    }
                    const bool first = p.y2 == nullptr || nq < p.nsplit1;
                    const size_t o = p.y2 == nullptr ? (size_t)m * p.Cout + nq
                                     : (first ? (size_t)m * p.nsplit1 + nq : (size_t)m * (p.Cout - p.nsplit1) + (nq - p.nsplit1));
                    store4_out(first ? (void*)p.y : (void*)p.y2, o, a, BF16 ? p.io : 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + li;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        const float sv = p.oscale ? p.oscale[n] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = row_to_m(wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                if (m >= p.M) continue;
                if (omap) {
                    const int ob = m / HoWo, orr = m - ob * HoWo;
                    const int oh = orr / p.Wo, ow = orr - oh * p.Wo;
                    m = (ob * p.oH + oh * p.osh + p.ooh) * p.oW + ow * p.osw + p.oow;
                }
                const float v = act_apply(acc[0][i][j][r] * sv + bv, p.act, p.alpha);
                if (BF16 && (p.io & 4)) {        // 16-bit output tensor(s)
                    LT* y = reinterpret_cast<LT*>(p.y);
                    LT* y2 = reinterpret_cast<LT*>(p.y2);
                    if (p.y2 == nullptr) y[(size_t)m * p.Cout + n] = (LT)v;
                    else if (n < p.nsplit1) y[(size_t)m * p.nsplit1 + n] = (LT)v;
                    else y2[(size_t)m * (p.Cout - p.nsplit1) + (n - p.nsplit1)] = (LT)v;
                    continue;
                }
                if (p.y2 == nullptr) p.y[(size_t)m * p.Cout + n] = v;
                else if (n < p.nsplit1) p.y[(size_t)m * p.nsplit1 + n] = v;
                else p.y2[(size_t)m * (p.Cout - p.nsplit1) + (n - p.nsplit1)] = v;
            }
        }
    }
}

// (the 128 x 64 tile with 64-channel K tiles needs 48 KB of LDS: three blocks fit a CU if the registers allow three waves per SIMD)
template <int BM, int BN, int WM, int WN, int PREC = 0, bool IN16 = false, int KT = BK>
__global__ __launch_bounds__(WM * WN * 64, (KT == 64 && BM * BN == 128 * 64) ? 3 : 1) void conv_fast_kernel(ConvParams p) {
    conv_fast_body<BM, BN, WM, WN, PREC, IN16, KT>(p, blockIdx.x, gridDim.x);
}
// up to 4 independent convolutions of one tile configuration in a single launch (blockIdx.y selects the problem): the
// parity classes of a strided convolution's data gradient are each too small to fill 256 CUs
struct ConvBatch { ConvParams p[4]; int nblk[4]; };
// 0: fp32 MFMA (default); 1: the fast-path forward / data-gradient convolutions round their operands to bf16 and use the bf16
// MFMA with fp32 accumulation (activations, weights and weight gradients stay fp32 in HBM) -- mmseg_set_conv_precision
static int g_conv_bf16 = 0;     // 0 fp32, 1 bf16, 2 fp16
// which kernel template the last convolution entry point launched (mmseg_conv2d_last_kernel): family * 1000000 + 500000 * flag +
// M tile * 1000 + N tile (flag: 16-byte gather of the generic kernels / 16-bit input of the fast kernels / two inputs of conv_wgrad_tr_kernel); families: 1 conv_fast_kernel, 2 conv_fwd_kernel (generic), 3 conv_direct_kernel, 4 conv_fast_batched_kernel,
// 5 conv_dgrad_s2k4_smallc_kernel, 6 conv_wgrad_tr_kernel, 7 conv_wgrad_fast_kernel, 8 conv_wgrad_kernel, 9 conv_wgrad_c8_kernel,
// 10 pw_reduce_kernel<LANES, COUT, VPL>, 12 pw_reduce_wgrad_kernel<...> (M tile field = LANES, N tile field = COUT);
// 11 smallk_conv_kernel<KS, CIN, L>, 13 smallk_wgrad_kernel<...> (M tile field = KS * 20 + L, N tile field = CIN); 14 conv_wgrad_tr_anyw_kernel; 16 conv16_kernel, 17 conv16h_kernel (M tile field = pixels per block), 18 wgrad32h_kernel<NCI, NCO>, 19 wgrad16h_kernel, 20 conv8h_kernel (M tile field = 16-bit x | 2 * 16-bit y | 4 * ReLU); 21 s2k3c9_fwd_kernel (N field 16) / s2k3c9_dgrad_kernel (N field 9), 22 s2k3c9_wgrad_kernel, 23 locnet5_fwd_kernel / locnet5_f32_kernel, 24 locnet5_wgrad_kernel
static int g_last_kernel = 0;
#define MMSEG_SET_LAST(fam, bm, bn) (g_last_kernel = (fam) * 1000000 + (bm) * 1000 + (bn))
template <int BM, int BN, int WM, int WN, int PREC = 0, bool IN16 = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_fast_batched_kernel(ConvBatch pb) {
    const int z = blockIdx.y;
    const int nb = pb.nblk[z];
    if ((int)blockIdx.x >= nb) return;
    conv_fast_body<BM, BN, WM, WN, PREC, IN16>(pb.p[z], blockIdx.x, nb);
}
template <int BM, int BN, int WM, int WN>
static int launch_fast_batched(ConvBatch& pb, int n, hipStream_t st) {
    int mx = 0;
    MMSEG_SET_LAST(4, BM, BN);
    for (int z = 0; z < n; ++z) {
        const ConvParams& p = pb.p[z];
        pb.nblk[z] = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
        if (pb.nblk[z] > mx) mx = pb.nblk[z];
    }
    for (int z = n; z < 4; ++z) pb.nblk[z] = 0;
    const bool in16 = g_conv_bf16 != 0 && (pb.p[0].io & 1);      // every problem of a batch shares the storage type of its input
    for (int z = 1; z < n; ++z)
        if ((pb.p[z].io & 1) != (pb.p[0].io & 1)) return (int)hipErrorInvalidValue;
    if (in16) g_last_kernel += 500000;
    const dim3 grid(mx, n), block(WM * WN * 64);
    if (g_conv_bf16 == 1 && in16) hipLaunchKernelGGL((conv_fast_batched_kernel<BM, BN, WM, WN, 1, true>), grid, block, 0, st, pb);
    else if (g_conv_bf16 == 2 && in16) hipLaunchKernelGGL((conv_fast_batched_kernel<BM, BN, WM, WN, 2, true>), grid, block, 0, st, pb);
    else if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_fast_batched_kernel<BM, BN, WM, WN, 1>), dim3(mx, n), dim3(WM * WN * 64), 0, st, pb);
    else if (g_conv_bf16 == 2) hipLaunchKernelGGL((conv_fast_batched_kernel<BM, BN, WM, WN, 2>), dim3(mx, n), dim3(WM * WN * 64), 0, st, pb);
    else hipLaunchKernelGGL((conv_fast_batched_kernel<BM, BN, WM, WN>), dim3(mx, n), dim3(WM * WN * 64), 0, st, pb);
    return MMSEG_CHECK_LAUNCH();
}

// MMSEG_FAST_K64=0 keeps the 16-bit kernels on 32-channel K tiles (A/B measurements)
static bool fast_k64_enabled() {
    static const bool on = ab_int("MMSEG_FAST_K64", 1) != 0;
    return on;
}
template <int BM, int BN, int WM, int WN>
static int launch_fast(const ConvParams& p, hipStream_t st) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Cout + BN - 1) / BN;
    MMSEG_SET_LAST(1, BM, BN);
    const bool in16 = g_conv_bf16 != 0 && (p.io & 1);
    if (in16) g_last_kernel += 500000;                  // the 16-bit-input instantiation
    // 16-bit tensors: 64-channel K tiles where a tile still lies inside one tap of one input tensor (with fp32 sources the
    // doubled staging registers cost occupancy: measured slower on the 64-output-channel layers)
    const bool k64 = in16 && BM * BN >= 128 * 64 && p.C1 % 64 == 0 && p.C2 % 64 == 0 && fast_k64_enabled();
    if (k64) {
        g_last_kernel += 250000;
        const dim3 grid(ntm * ntn), block(WM * WN * 64);
        if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 1, true, 64>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 2, true, 64>), grid, block, 0, st, p);
        return MMSEG_CHECK_LAUNCH();
    }
    if (g_conv_bf16 == 1 && in16) hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 1, true>), dim3(ntm * ntn), dim3(WM * WN * 64), 0, st, p);
    else if (g_conv_bf16 == 2 && in16) hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 2, true>), dim3(ntm * ntn), dim3(WM * WN * 64), 0, st, p);
    else if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 1>), dim3(ntm * ntn), dim3(WM * WN * 64), 0, st, p);
    else if (g_conv_bf16 == 2) hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN, 2>), dim3(ntm * ntn), dim3(WM * WN * 64), 0, st, p);
    else hipLaunchKernelGGL((conv_fast_kernel<BM, BN, WM, WN>), dim3(ntm * ntn), dim3(WM * WN * 64), 0, st, p);
    return MMSEG_CHECK_LAUNCH();
}

template <int BM, int BN, int WM, int WN>
static int launch_fwd(const ConvParams& p, bool vec, hipStream_t st) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Cout + BN - 1) / BN;
    MMSEG_SET_LAST(2, BM, BN);
    if (vec) g_last_kernel += 500000;                   // the 16-byte-gather instantiation
    dim3 grid(ntm * ntn), block(WM * WN * 64);
    // reduced-precision modes: forward launches with 4-channel gathers multiply 16-bit operands (MMSEG_GENERIC_LP=0: fp32, for A/B runs)
    static const bool lp_on = ab_int("MMSEG_GENERIC_LP", 1) != 0;
    if (vec && !p.transposed && g_conv_bf16 != 0 && lp_on) {
        g_last_kernel += 250000;
        if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true, 2>), grid, block, 0, st, p);
        return MMSEG_CHECK_LAUNCH();
    }
    if (vec) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, false>), grid, block, 0, st, p);
    return MMSEG_CHECK_LAUNCH();
}

// =====================================================================================
// Direct convolution for the tiny-channel layers of the FiLM decoder (3x3, 8 -> 8, stride 1, 'same'): arithmetic
// intensity ~18 FLOP/B, i.e. HBM-bound -- an MFMA tile would be >= 75 % padding.  One thread per output pixel, the 576
// weights come through scalar loads (wave-uniform addresses), inputs as 16-byte loads of the 9 taps (L1/L2 hits).
// Replaces Conv2D(8, 3, padding='same') of model_components/decoder.py:45-48,58 and its data gradient.
// =====================================================================================
typedef unsigned int u32x4d __attribute__((ext_vector_type(4)));
template <int CIN, int COUT, int KS>
__global__ __launch_bounds__(256) void conv_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int B, int H, int W, int pad, int act, float alpha) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npix = (long)B * H * W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(npix * CIN * 4), 0x00020000);
    const bool live = idx < npix;
    const int wq = idx % W; long r = idx / W;
    const int hq = r % H; const int b = r / H;
    // accumulators as pairs of output channels: the compiler issues v_pk_fma_f32 (two fp32 FMAs per lane and instruction, the pixel's
    // input value broadcast to both halves by op_sel, the weight pair straight from an SGPR pair); same FMAs in the same order:
    // bit-identical results.  Measured (round 3): half the VALU instructions, the same 70 us per launch -- see the note below.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(COUT % 2 == 0, "output channels in pairs");
    f32x2 acc[COUT / 2];
#pragma unroll
    for (int co = 0; co < COUT / 2; ++co) acc[co] = bias ? f32x2{bias[2 * co], bias[2 * co + 1]} : f32x2{0.f, 0.f};
    // one filter tap per (NOT unrolled) iteration: its CIN*COUT weights fit the scalar register file (a fully unrolled
    // body would need 576 SGPRs and spill them through v_readlane); the tap's inputs are fetched branch-free (padding
    // taps get an out-of-range offset and read as zeros).  Round 3: the inputs of tap t + 1 are requested BEFORE the FMAs of
    // tap t (two register sets, the loop advances two taps per trip).  Measured: neither this prefetch nor the packed FMAs move
    // the kernel (1.91 -> 1.88 ms per iteration, 2.9 TB/s of algorithmic bytes = 52 TFLOP/s, i.e. 75 % of the UNPACKED fp32 VALU
    // rate): PMC shows 485 VALU instructions per wave and a launch time of exactly 288 x 8 + 197 x 4 cycles per wave -- the packed
    // FMA issues at half rate here, the kernel is VALU-bound.  The weights as VGPR pairs (staged once per block
    // in LDS, broadcast reads per tap) instead of SGPR pairs: measured 1.95 ms, no better -- it is not the scalar operand.
    auto fetch = [&](int t, f32x4 (&xv)[CIN / 4]) {
        const int kh = t / KS, kw = t - kh * KS;
        const int hi = hq + kh - pad, wi = wq + kw - pad;
        const bool ok = live && t < KS * KS && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const int off = (((b * H + hi) * W + wi) * CIN) * 4;
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q)
            xv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off + 16 * q : 0x7ffffff0, 0, 0));
    };
    auto mac = [&](int t, const f32x4 (&xv)[CIN / 4]) {
        const f32x2* wt = reinterpret_cast<const f32x2*>(w + t * CIN * COUT);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float xs = xv[ci >> 2][ci & 3];
            const f32x2 x2 = {xs, xs};
#pragma unroll
            for (int co = 0; co < COUT / 2; ++co) acc[co] = __builtin_elementwise_fma(x2, wt[ci * (COUT / 2) + co], acc[co]);
        }
    };
    f32x4 xa[CIN / 4], xb[CIN / 4];
    fetch(0, xa);
#pragma unroll 1
    for (int t = 0; t < KS * KS; t += 2) {
        fetch(t + 1, xb);                 // (t + 1 == KS * KS on the last trip of an odd tap count: an out-of-range offset, zeros)
        mac(t, xa);
        if (t + 1 < KS * KS) {
            fetch(t + 2, xa);
            mac(t + 1, xb);
        }
    }
    if (!live) return;
    f32x4* dst = reinterpret_cast<f32x4*>(y + idx * COUT);
#pragma unroll
    for (int q = 0; q < COUT / 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(acc[2 * q + (e >> 1)][e & 1], act, alpha);
        dst[q] = v;
    }
}

// The same layer on the 4 x 4 x 1 MFMA (16 independent 4 x 4 outer products per instruction).  A lane still owns one output pixel and
// requests its 9 taps up front (18 16-byte loads in flight); per (tap, input channel) it feeds the pixel's value as the A operand of two
// MFMAs whose B operands are the weights of output channels (lane & 3) and 4 + (lane & 3) -- a block of the instruction is 4 consecutive
// pixels x 4 channels, the accumulators hold (row = pixel of the quad, lane = channel).  Weights sit in LDS as [k][channel & 3][channel >> 2]:
// one 8-byte broadcast read serves both MFMAs.  Same products in the same order per output as the VALU kernel above, no VALU arithmetic
// left in the loop.  Blocks are renumbered so that consecutive image rows run on ONE XCD: the halo rows of a block are L2 hits instead
// of a second and third fetch through the fabric -- PMC (`tools/pmc_direct_ab.sh`, batch 48): 294 -> 98 MB of reads per launch (1.0 x the
// algorithmic bytes, was 3.0 x), 67.4 -> 61.8 us; the MFMA pipe is busy 40 % of the launch.  Measured and rejected: the
// taps staged through LDS as one contiguous pixel range (79.9 us: 5 blocks per CU and a barrier between the load and MFMA phases), a
// persistent block walking several tiles with the next tile's taps in flight (103.7 us: 2-3 waves per SIMD do not cover the latency).
template <int KS>
__global__ __launch_bounds__(256) void conv_direct_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int B, int H, int W, int pad, int act, float alpha) {
    constexpr int CIN = 8, COUT = 8, NT = KS * KS;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float wl[NT * CIN * COUT];
    for (int i = threadIdx.x; i < NT * CIN * COUT; i += 256) {
        const int k = i >> 3, co = i & 7;
        wl[k * 8 + (co & 3) * 2 + (co >> 2)] = w[i];
    }
    __syncthreads();
    const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const long npix = (long)B * H * W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(npix * CIN * 4), 0x00020000);
    const bool live = idx < npix;
    const int wq = idx % W; long r = idx / W;
    const int hq = r % H; const int b = r / H;
    const int j = threadIdx.x & 3;
    const float b0 = bias ? bias[j] : 0.f, b1 = bias ? bias[4 + j] : 0.f;
    f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
    f32x4 xt[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int kh = t / KS, kw = t - kh * KS;
        const int hi = hq + kh - pad, wi = wq + kw - pad;
        const bool ok = live && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const int off = (((b * H + hi) * W + wi) * CIN) * 4;
        xt[t][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0x7ffffff0, 0, 0));
        xt[t][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off + 16 : 0x7ffffff0, 0, 0));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* wt = wl + t * CIN * COUT + j * 2;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const f32x2 wv = *reinterpret_cast<const f32x2*>(wt + ci * 8);
            const float xs = xt[t][ci >> 2][ci & 3];
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xs, wv[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xs, wv[1], acc1, 0, 0, 0);
        }
    }
    // accumulator register q of lane (quad, j) = output channel j (acc0) / 4 + j (acc1) of the quad's pixel q: transposed inside the quad
    // a lane holds channels 0..3 / 4..7 of ITS pixel -- two 16-byte stores
    float a0[4], a1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a0[e] = act_apply(acc0[e], act, alpha); a1[e] = act_apply(acc1[e], act, alpha); }
    quad_transpose4(a0, j);
    quad_transpose4(a1, j);
    if (!live) return;
    f32x4* dst = reinterpret_cast<f32x4*>(y + idx * COUT);
    dst[0] = f32x4{a0[0], a0[1], a0[2], a0[3]};
    dst[1] = f32x4{a1[0], a1[1], a1[2], a1[3]};
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

#include "smallconv.hpp"

// HBM-bound small-channel layers (smallconv.hpp): -1 = not one of them, otherwise the launch's return code
static int smallconv_dispatch(const ConvParams& p, hipStream_t st) {
    static const int mask = ab_int("MMSEG_SMALLCONV", 15);     // A/B: 1 reduce fwd, 2 expand fwd
    const bool on = mask != 0;
    if (!on || p.io != 0 || p.C2 != 0 || p.ups || p.transposed || p.y2 != nullptr || p.w == nullptr || p.osh != 1 || p.osw != 1 || p.ooh != 0 ||
        p.oow != 0 || p.oH != p.Ho || p.oW != p.Wo || !aligned16(p.x1) || !aligned16(p.y) || !aligned16(p.w)) return -1;
    const long M = p.M;
    const bool one = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0;
    // "reduce" type: wide input, <= 8 outputs
    if (one && (mask & 1)) {
#define PWR(LANES, COUT, VPL)                                                                                                         \
        do {                                                                                                                           \
            long nb = (M + 64 * (64 / LANES) - 1) / (64 * (64 / LANES));     /* >= 4 iterations per wave: the weight prologue */       \
            if (nb > 2048) nb = 2048;                                                                                                  \
            hipLaunchKernelGGL((pw_reduce_kernel<LANES, COUT, VPL>), dim3((unsigned)nb), dim3(256), 0, st, p.x1, p.w, p.bias, p.oscale, p.y, M, \
                               p.act, p.alpha, g_conv_bf16);                                                                           \
            MMSEG_SET_LAST(10, LANES, COUT);                                                                                           \
            return MMSEG_CHECK_LAUNCH();                                                                                               \
        } while (0)
        if (p.C1 == 64 && p.Cout == 5) PWR(16, 5, 1);
        if (p.C1 == 64 && p.Cout == 8) PWR(8, 8, 2);
        if (p.C1 == 16 && p.Cout == 1) PWR(4, 1, 1);
        if (p.C1 == 8 && p.Cout == 1) PWR(2, 1, 1);
#undef PWR
    }
    // "expand" type: <= 16 gathered values per pixel, L = Cout / 4 lanes per pixel
    const int L = p.Cout / 4;
    if (!(mask & 2)) return -1;
    static const int only_ks = ab_int("MMSEG_SMALLCONV_KS", 0);
    if (only_ks && p.KH != only_ks) return -1;
    if (p.Cout % 4 != 0 || (L != 2 && L != 4 && L != 16) || p.KH != p.KW || (p.bias && !aligned16(p.bias)) ||
        (p.oscale && !aligned16(p.oscale))) return -1;
#define SMK1(KS, CIN, LL)                                                                                                              \
    do {                                                                                                                               \
        long nb = (M * LL + 1023) / 1024;                                                                                              \
        if (nb > 2048) nb = 2048;                                                                                                      \
        hipLaunchKernelGGL((smallk_conv_kernel<KS, CIN, LL>), dim3((unsigned)nb), dim3(256), 0, st, p, (CIN % 4 == 0) ? g_conv_bf16 : 0); \
        MMSEG_SET_LAST(11, KS * 20 + LL, CIN);                                                                                         \
        return MMSEG_CHECK_LAUNCH();                                                                                                   \
    } while (0)
#define SMK(KS, CIN)                                                                                                                   \
    do {                                                                                                                               \
        if (L == 16) SMK1(KS, CIN, 16);                                                                                                \
        if (L == 4) SMK1(KS, CIN, 4);                                                                                                  \
        SMK1(KS, CIN, 2);                                                                                                              \
    } while (0)
    if (one && p.C1 == 1) SMK(1, 1);
    if (one && p.C1 == 5) SMK(1, 5);
    if (one && p.C1 == 8) SMK(1, 8);
    if (p.KH == 3 && p.C1 == 1) SMK(3, 1);
    if (p.KH == 4 && p.C1 == 1) SMK(4, 1);
#undef SMK1
#undef SMK
    return -1;
}

#include "conv16.hpp"
#include "s2conv.hpp"

// the quad-transposed epilogue stores 4 channels per lane: 16-byte (fp32) / 8-byte (16-bit) aligned output rows
static int quad_epilogue_ok(const ConvParams& p) {
    static const bool on = ab_int("MMSEG_QUAD_EPI", 1) != 0;
    const uintptr_t al = (p.io & 4) ? 7 : 15;
    return (on && (reinterpret_cast<uintptr_t>(p.y) & al) == 0 && (p.y2 == nullptr || (reinterpret_cast<uintptr_t>(p.y2) & al) == 0)) ? 1 : 0;
}

static int conv_dispatch(ConvParams& p, hipStream_t st) {
    if (p.M <= 0 || p.Cout <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
    p.qepi = quad_epilogue_ok(p);
    if (p.C1 == 8 && p.C2 == 0 && p.Cout == 8 && p.KH == 3 && p.KW == 3 && p.stride == 1 && !p.transposed && !p.ups &&
        p.Ho == p.H && p.Wo == p.W && p.pad_h == 1 && p.pad_w == 1 && p.y2 == nullptr && p.w != nullptr && p.osh == 1 &&
        p.oscale == nullptr && aligned16(p.x1) && aligned16(p.y) && (long)p.M * 8 * 4 < (1L << 31) - 64 && p.io == 0) {
        static const int direct_mfma = ab_int("MMSEG_DIRECT_MFMA", 1);       // 0 (measurement builds): the VALU kernel
        if (direct_mfma)
            hipLaunchKernelGGL((conv_direct_mfma_kernel<3>), dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st,
                               p.x1, p.w, p.bias, p.y, p.B, p.H, p.W, 1, p.act, p.alpha);
        else
            hipLaunchKernelGGL((conv_direct_kernel<8, 8, 3>), dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st,
                               p.x1, p.w, p.bias, p.y, p.B, p.H, p.W, 1, p.act, p.alpha);
        MMSEG_SET_LAST(3, 8, 8);
        return MMSEG_CHECK_LAUNCH();
    }
    {
        const int rc = smallconv_dispatch(p, st);
        if (rc >= 0) return rc;
    }
    {
        const int rc = s2conv_dispatch(p, st);       // the modality encoder's first layer (9 -> 16, 3x3, stride 2) and its data gradient
        if (rc >= 0) return rc;
    }
    const bool vec = (p.C1 % 4 == 0) && (p.C2 % 4 == 0) && aligned16(p.x1) && (p.C2 == 0 || aligned16(p.x2));
    const long tiles_big = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    const long lim = (1L << 31) - 64;
    const bool fast = p.wt != nullptr && aligned16(p.wt) && vec && !p.transposed && (p.C1 % 32 == 0) && (p.C2 % 32 == 0) && (p.Cout % 4 == 0) && aligned16(p.w) &&
                      p.KH * p.KW <= 32 &&                  // one validity bit per tap and row
                      (long)p.B * p.H1 * p.W1 * p.C1 * ((p.io & 1) ? 2 : 4) < lim && (long)p.B * p.H * p.W * p.C2 * ((p.io & 2) ? 2 : 4) < lim &&
                      (long)p.K * p.Cout * 4 < lim;        // 32-bit buffer offsets (bytes of the tensors as they are stored)
    const bool omap = p.osh != 1 || p.osw != 1 || p.ooh != 0 || p.oow != 0 || p.oH != p.Ho || p.oW != p.Wo;
    if (omap && !fast) return (int)hipErrorInvalidValue;     // strided output mapping exists on the fast path only
    if (!fast && (p.io & 3)) return (int)hipErrorInvalidValue;       // only the fast path reads 16-bit input tensors
    if (!fast && p.w == nullptr) return (int)hipErrorInvalidValue;   // the generic kernels read the Keras-layout weights: a caller
                                                                     // that only prepared `wt` must not fall through to them
    if (fast && g_conv_bf16 != 0) {
        // 16-bit tensors, channel counts multiples of 64, enough pixels: the large-tile direct-to-LDS kernel (conv16.hpp)
        const int bnh = conv16h_tile(p);                     // 3x3 'same': the activation patch stays in LDS across the nine taps
        const int bn = bnh ? 0 : conv16_tile(p);
        if (bnh) {
            MMSEG_SET_LAST(17, conv16h_rows(p, bnh) * 32, bnh);
            return g_conv_bf16 == 1 ? launch_conv16h_prec<1>(p, bnh, st) : launch_conv16h_prec<2>(p, bnh, st);
        }
        if (bn) {
            MMSEG_SET_LAST(16, 256, bn);
            return g_conv_bf16 == 1 ? launch_conv16_prec<1>(p, bn, st) : launch_conv16_prec<2>(p, bn, st);
        }
    }
    if (fast && g_conv_bf16 == 0) {
        // fp32, 3x3 'same', enough pixels: the patch-resident large-tile kernel on the fp32 MFMA (conv16.hpp, PREC 0)
        const int bnh = conv16h_tile(p, true);
        if (bnh) {
            MMSEG_SET_LAST(17, conv16h_rows(p, bnh) * 32, bnh);
            return launch_conv16h_prec<0>(p, bnh, st);
        }
    }
    if (fast) {
        static const int force_tile = ab_int("MMSEG_FAST_TILE", 0);   // tile A/B measurements
        if (force_tile == 1) return launch_fast<128, 128, 2, 2>(p, st);
        if (force_tile == 2) return launch_fast<128, 64, 2, 2>(p, st);
        if (force_tile == 3) return launch_fast<64, 64, 2, 2>(p, st);
        if (force_tile == 4) return launch_fast<128, 32, 4, 1>(p, st);
        if (p.Cout > 64 && tiles_big >= 384) return launch_fast<128, 128, 2, 2>(p, st);
        if (p.Cout > 32) {
            const long tiles_mid = (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64);
            if (tiles_mid >= 384) return launch_fast<128, 64, 2, 2>(p, st);
            return launch_fast<64, 64, 2, 2>(p, st);
        }
        return launch_fast<128, 32, 4, 1>(p, st);
    }
    if (p.Cout > 64 && tiles_big >= 384) return launch_fwd<128, 128, 2, 2>(p, vec, st);
    if (p.Cout > 32) {
        const long tiles_mid = (long)((p.M + 127) / 128) * ((p.Cout + 63) / 64);
        if (tiles_mid >= 384) return launch_fwd<128, 64, 2, 2>(p, vec, st);
        return launch_fwd<64, 64, 2, 2>(p, vec, st);
    }
    return launch_fwd<128, 32, 4, 1>(p, vec, st);
}

// =====================================================================================
// Data gradient of the discriminators' FIRST layer (4x4, stride 2, valid, Cin = 1 image channel or 4 mask channels,
// Cout = 64): dx[y, x, ci] = sum_{kh = y mod 2 (+2), kw = x mod 2 (+2)} sum_co dy[(y-kh)/2, (x-kw)/2, co] * w[kh, kw, ci, co].
// An N = Cin GEMM is hopeless on 32-wide MFMA tiles (0.5 TFLOP/s through the generic path); the op is a 64-term dot
// product per tap, HBM/L2-bound on reading dy.  Mapping: 16 lanes own the 64 channels of a dy pixel (one coalesced 256-byte
// read); such a group handles index (i, j) = the 2x2 input pixels (2i+ph, 2j+pw), which all read the SAME four dy pixels
// (i-di, j-dj) with different taps (kh = ph + 2di, kw = pw + 2dj) -> every dy pixel is fetched 4 times instead of 16;
// partial dots are reduced over the 16 lanes with 4 shuffles.  Weights (16 taps x Cin x 64) sit in LDS.
// =====================================================================================
template <int CIN>
__global__ __launch_bounds__(256) void conv_dgrad_s2k4_smallc_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                     float* __restrict__ dx, int B, int H, int W, int Ho, int Wo) {
    constexpr int CO = 64;
    __shared__ __attribute__((aligned(16))) float wl[16 * CIN * CO];
    for (int i = threadIdx.x; i < 16 * CIN * CO / 4; i += 256)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w)[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int Hi = (H + 1) >> 1, Wi = (W + 1) >> 1;
    const long total = (long)B * Hi * Wi;
    const long gidx = (long)blockIdx.x * 16 + grp;
    const bool live = gidx < total;
    const long gi = live ? gidx : 0;
    const int b = (int)(gi / ((long)Hi * Wi));
    const int r = (int)(gi - (long)b * Hi * Wi);
    const int i = r / Wi, j = r - i * Wi;
    float acc[2][2][CIN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc[a][c][ci] = 0.f;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int oy = i - di, ox = j - dj;
            if (live && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * Ho + oy) * Wo + ox) * CO + 4 * l16);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw) {
                        const int tap = (ph + 2 * di) * 4 + (pw + 2 * dj);
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[(tap * CIN + ci) * CO + 4 * l16]);
                            acc[ph][pw][ci] += g[0] * wv[0] + g[1] * wv[1] + g[2] * wv[2] + g[3] * wv[3];
                        }
                    }
            }
        }
    }
    // reduce over the 16 lanes of the group; afterwards lane q < 4*CIN writes value q = (ph, pw, ci)
    float mine = 0.f;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                float v = acc[ph][pw][ci];
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
                if (l16 == (ph * 2 + pw) * CIN + ci) mine = v;
            }
    if (live && l16 < 4 * CIN) {
        const int q = l16 / CIN, ci = l16 - q * CIN;
        const int y = 2 * i + (q >> 1), x = 2 * j + (q & 1);
        if (y < H && x < W) dx[(((size_t)b * H + y) * W + x) * CIN + ci] = mine;
    }
}

// =====================================================================================
// Direct weight gradient of the 3x3, 8 -> 8 layers (FiLM decoder): dW[tap][ci][co] = sum_pix x[pix + tap][ci] * dy[pix][co].
// 576 outputs only, so the work is a long reduction over pixels: a block stages an (8+2) x (64+2) input patch and the
// 8 x 64 gradient tile in LDS in CHANNEL-MAJOR rows (so a lane's 4 consecutive pixels are one 16-byte LDS read); lane
// (ci, co) of every wave keeps the 9 taps of its (ci, co) pair in registers; blocks write 576-float slabs that the
// deterministic slab reduction sums.  VALU/LDS-bound, ~10x faster than pushing a 72 x 8 GEMM through 32x32 MFMA tiles.
// =====================================================================================
#define WG8_ROWS 8
#define WG8_COLS 64
#define WG8_LD 68
__global__ __launch_bounds__(256) void conv_wgrad_c8_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ ws, int B, int H, int W, int ntiles) {
    __shared__ __attribute__((aligned(16))) float xs[(WG8_ROWS + 2) * 8 * WG8_LD];
    __shared__ __attribute__((aligned(16))) float dys[WG8_ROWS * 8 * WG8_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ci = lane >> 3, co = lane & 7;
    const int tpr = W / WG8_COLS, tpi = (H / WG8_ROWS) * tpr;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tpi, tr = tile - b * tpi;
        const int r0 = (tr / tpr) * WG8_ROWS, c0 = (tr % tpr) * WG8_COLS;
        __syncthreads();                                   // previous tile fully consumed
        // stage the input patch (zero padded) and the gradient tile, channel-major rows
        for (int i = tid; i < (WG8_ROWS + 2) * (WG8_COLS + 2) * 2; i += 256) {
            const int half = i & 1, pc = (i >> 1) % (WG8_COLS + 2), pr = (i >> 1) / (WG8_COLS + 2);
            const int hi = r0 + pr - 1, wi = c0 + pc - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + hi) * W + wi) * 8 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[(pr * 8 + 4 * half + e) * WG8_LD + pc] = v[e];
        }
        for (int i = tid; i < WG8_ROWS * WG8_COLS * 2; i += 256) {
            const int half = i & 1, pc = (i >> 1) % WG8_COLS, pr = (i >> 1) / WG8_COLS;
            const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * H + r0 + pr) * W + c0 + pc) * 8 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) dys[(pr * 8 + 4 * half + e) * WG8_LD + pc] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < WG8_ROWS / 4; ++rr) {
            const int row = wid * (WG8_ROWS / 4) + rr;
#pragma unroll 4
            for (int c4 = 0; c4 < WG8_COLS / 4; ++c4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(&dys[(row * 8 + co) * WG8_LD + 4 * c4]);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* xr = &xs[((row + kh) * 8 + ci) * WG8_LD + 4 * c4];
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(xr);
                    const float x4 = xr[4], x5 = xr[5];
                    acc[kh * 3 + 0] += xa[0] * g[0] + xa[1] * g[1] + xa[2] * g[2] + xa[3] * g[3];
                    acc[kh * 3 + 1] += xa[1] * g[0] + xa[2] * g[1] + xa[3] * g[2] + x4 * g[3];
                    acc[kh * 3 + 2] += xa[2] * g[0] + xa[3] * g[1] + x4 * g[2] + x5 * g[3];
                }
            }
        }
    }
    __syncthreads();
    float* red = xs;                                       // 4 waves x 576 partials
#pragma unroll
    for (int t = 0; t < 9; ++t) red[wid * 576 + t * 64 + lane] = acc[t];
    __syncthreads();
    for (int o = tid; o < 576; o += 256)
        ws[(size_t)blockIdx.x * 576 + o] = red[o] + red[576 + o] + red[1152 + o] + red[1728 + o];
}

// The same weight gradient on v_mfma_f32_4x4x1_16B_f32 (round 4): the instruction's 16 blocks are 16 consecutive pixels, block rows = 4 input
// channels, block columns = 4 output channels -- each block accumulates its own pixel stream; the 16 streams (and the waves) meet once per
// launch in a fixed order.  Wave (h, q) owns input-channel half h and output-channel half q: 9 tap accumulators (36 registers), per 16 pixels
// one ds_read_b32 of dy, nine of x and 9 MFMAs, no VALU arithmetic.  The tile (same 8 x 64 pixels + halo as above) is staged AS IT LIES, two
// 16-byte writes per pixel; the two halves of a pixel swap places when bit 3 of its column is set, so that the 16 pixels of a read (32-byte
// stride: p and p + 8 would meet in a bank) cover the 64 banks once; the tile loop is fully unrolled: an operand address is one of four per-lane
// offsets plus an immediate (the first version computed the swizzle per read: 5 VALU per MFMA, slower than the VALU kernel).  The VALU kernel above spent 36 FMAs x 4 cycles per 4 pixels and a transposing LDS stage
// (4 ds_write_b32 per loaded vector).
__global__ __launch_bounds__(256) void conv_wgrad_c8m_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ ws, int B, int H, int W, int ntiles) {
    constexpr int PW = 80;                                             // patch row pitch in pixels (66 used): a multiple of 16, so that bit 3 of a
                                                                       // pixel's index -- the swizzle key -- is bit 3 of its COLUMN
    __shared__ __attribute__((aligned(16))) float xs[(WG8_ROWS + 2) * PW * 8];
    __shared__ __attribute__((aligned(16))) float dys[WG8_ROWS * WG8_COLS * 8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int blk = lane >> 2, i4 = lane & 3;
    const int h = wid >> 1, q = wid & 1;
    const int tpr = W / WG8_COLS, tpi = (H / WG8_ROWS) * tpr;
    // per-lane operand offsets (floats): everything else of an address is a compile-time constant of the unrolled tile loop
    int offx[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) offx[kw] = blk * 8 + 4 * (h ^ (((blk + kw) >> 3) & 1)) + i4;
    const int offd = blk * 8 + 4 * (q ^ (blk >> 3)) + i4;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, B * H * W * 8 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, B * H * W * 8 * 4, 0x00020000);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tpi, tr = tile - b * tpi;
        const int r0 = (tr / tpr) * WG8_ROWS, c0 = (tr % tpr) * WG8_COLS;
        // all of the tile's loads are requested before the first LDS write (branch-free: halo pixels outside the image get an out-of-range
        // offset and read as zeros) -- one exposed memory latency per tile, not one per staging iteration
        constexpr int NXP = (WG8_ROWS + 2) * (WG8_COLS + 2), XIT = (NXP + 255) / 256, DIT = WG8_ROWS * WG8_COLS / 256;
        f32x4 xv[XIT][2], dv[DIT][2];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = tid + 256 * it;
            const int pr = i / (WG8_COLS + 2), pc = i - pr * (WG8_COLS + 2);
            const int hi = r0 + pr - 1, wi = c0 + pc - 1;
            const bool ok = i < NXP && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            const int off = (((b * H + hi) * W + wi) * 8) * 4;
            xv[it][0] = buf_load4(rx, ok ? off : BUF_OOB);
            xv[it][1] = buf_load4(rx, ok ? off + 16 : BUF_OOB);
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int i = tid + 256 * it;
            const int pr = i / WG8_COLS, pc = i - pr * WG8_COLS;
            const int off = (((b * H + r0 + pr) * W + c0 + pc) * 8) * 4;
            dv[it][0] = buf_load4(rd, off);
            dv[it][1] = buf_load4(rd, off + 16);
        }
        __syncthreads();                                   // previous tile fully consumed (its loads above were already in flight)
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = tid + 256 * it;
            const int pr = i / (WG8_COLS + 2), pc = i - pr * (WG8_COLS + 2);
            const int sw = (pc >> 3) & 1;
            float* dst = xs + (pr * PW + pc) * 8;
            if (i < NXP) {
                *reinterpret_cast<f32x4*>(dst + 4 * sw) = xv[it][0];
                *reinterpret_cast<f32x4*>(dst + 4 * (sw ^ 1)) = xv[it][1];
            }
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int i = tid + 256 * it;
            const int sw = ((i % WG8_COLS) >> 3) & 1;
            *reinterpret_cast<f32x4*>(dys + i * 8 + 4 * sw) = dv[it][0];
            *reinterpret_cast<f32x4*>(dys + i * 8 + 4 * (sw ^ 1)) = dv[it][1];
        }
        __syncthreads();
#pragma unroll
        for (int row = 0; row < WG8_ROWS; ++row)
#pragma unroll
            for (int cg = 0; cg < WG8_COLS / 16; ++cg) {
                const float g = dys[(row * WG8_COLS + 16 * cg) * 8 + offd];                   // dy[pixel][co = 4 q + i4]
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float a = xs[((row + kh) * PW + 16 * cg + kw) * 8 + offx[kw]];  // x[pixel + tap][ci = 4 h + i4]
                        acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, g, acc[kh * 3 + kw], 0, 0, 0);
                    }
            }
    }
    // the 16 pixel streams of a wave: a fixed butterfly over the lane groups; register r of lane (block, j) = D[block][ci = 4 h + r][co = 4 q + j]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[t][r] = v;
        }
    if (lane < 4) {
        float* slab = ws + (size_t)blockIdx.x * 576;        // [tap][ci][co]: every element has ONE owner (wave = channel halves, lane = column)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[t * 64 + (4 * h + r) * 8 + 4 * q + lane] = acc[t][r];
    }
}

// =====================================================================================
// wgrad: dW[k, n] = sum_m A[m, k] dy[m, n]
// =====================================================================================
struct WgradParams {
    ConvParams c;       // geometry + x1/x2 (gather source); c.y unused
    const float* dy;    // [M, Cout]
    float* ws;          // [S, K, Cout] slabs (or dW itself when S == 1)
    int chunk;          // pixels per split (multiple of 32)
};

template <int BKT, int BNT, int WM, int WN, bool VEC>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_kernel(WgradParams q) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BKT / WM / 32, TN = BNT / WN / 32;
    constexpr int PT = 32;                       // pixels per LDS stage
    constexpr int AF4_PER_ROW = BKT / 4, A_RPP = NT / AF4_PER_ROW, A_F4 = (PT + A_RPP - 1) / A_RPP;
    constexpr int DF4_PER_ROW = BNT / 4, D_RPP = NT / DF4_PER_ROW, D_F4 = (PT + D_RPP - 1) / D_RPP;
    const ConvParams& p = q.c;

    __shared__ __attribute__((aligned(16))) float Ap[PT * BKT];
    __shared__ __attribute__((aligned(16))) float Dp[PT * BNT];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = blockIdx.x * BKT, n0 = blockIdx.y * BNT;
    const int pbeg = blockIdx.z * q.chunk;
    const int pend = min(p.M, pbeg + q.chunk);

    // A gather: this thread always fetches the same 4 k's (fixed tap / channel), rows advance
    const int a_kc = tid % AF4_PER_ROW, a_r0 = tid / AF4_PER_ROW;
    const int a_k = k0 + 4 * a_kc;
    const int Cin = p.C1 + p.C2;
    int a_tap = 0, a_c = 0;
    if (VEC) { a_tap = a_k / Cin; a_c = a_k - a_tap * Cin; }
    int r_b[A_F4], r_ho[A_F4], r_wo[A_F4];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        const int m = pbeg + a_r0 + j * A_RPP;
        const int b = m / HoWo, r = m - b * HoWo;
        r_b[j] = b; r_ho[j] = r / p.Wo; r_wo[j] = r - r_ho[j] * p.Wo;
    }
    const int d_nc = tid % DF4_PER_ROW, d_r0 = tid / DF4_PER_ROW;
    const bool dy_vec = (p.Cout & 3) == 0;

    f32x4 ra[A_F4], rd[D_F4];
    auto load_stage = [&](int ps) {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int pr = a_r0 + j * A_RPP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pr < PT && ps + pr < pend) {
                const int hb = r_ho[j] * p.stride - p.pad_h, wb = r_wo[j] * p.stride - p.pad_w;
                v = gather_a<VEC>(p, r_b[j], hb, wb, a_k, a_tap, a_c);
            }
            ra[j] = v;
            // advance this row's pixel by PT
            r_wo[j] += PT;
            while (r_wo[j] >= p.Wo) { r_wo[j] -= p.Wo; if (++r_ho[j] >= p.Ho) { r_ho[j] = 0; ++r_b[j]; } }
        }
#pragma unroll
        for (int j = 0; j < D_F4; ++j) {
            const int pr = d_r0 + j * D_RPP;
            const int m = ps + pr, n = n0 + 4 * d_nc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pr < PT && m < pend) {
                const float* src = q.dy + (size_t)m * p.Cout + n;
                if (dy_vec) { if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(src); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.Cout) v[e] = src[e];
                }
            }
            rd[j] = v;
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int pr = a_r0 + j * A_RPP;
            if (pr < PT) *reinterpret_cast<f32x4*>(&Ap[pr * BKT + 4 * a_kc]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < D_F4; ++j) {
            const int pr = d_r0 + j * D_RPP;
            if (pr < PT) *reinterpret_cast<f32x4*>(&Dp[pr * BNT + 4 * d_nc]) = rd[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pbeg < pend) {
        load_stage(pbeg);
        store_stage();
        __syncthreads();
        const int a_col = wm * (BKT / WM) + li, b_col = wn * (BNT / WN) + li;
        for (int ps = pbeg; ps < pend; ps += PT) {
            const bool more = ps + PT < pend;
            if (more) load_stage(ps + PT);
#pragma unroll
            for (int s = 0; s < PT / 2; ++s) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ap[(2 * s + lh) * BKT + a_col + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Dp[(2 * s + lh) * BNT + b_col + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) { store_stage(); __syncthreads(); }
        }
    }

    float* out = q.ws + (size_t)blockIdx.z * p.K * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BNT / WN) + j * 32 + li;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * (BKT / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < p.K) out[(size_t)k * p.Cout + n] = acc[i][j][r];
            }
    }
}

// out[g][i] = sum_{s in group g} ws[s][i]   (fixed order -> deterministic).  block = 64 float4 columns x 4 slab lanes;
// grid.y = number of slab groups (1 for the final pass).  Many slabs of a SMALL matrix (e.g. 1024 x 576 floats for the
// 8 -> 8 layers) are reduced in two passes so that no thread walks more than ~8 slabs serially.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long n, int S,
                                                          int per_group, int accumulate) {
    __shared__ f32x4 sm[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i4 = ((long)blockIdx.x * 64 + cl) * 4;
    const int s0 = blockIdx.y * per_group, s1 = min(S, s0 + per_group);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i4 + 3 < n) {
#pragma unroll 4
        for (int s = s0 + sl; s < s1; s += 4) a += *reinterpret_cast<const f32x4*>(ws + (size_t)s * n + i4);
    } else {
        for (int s = s0 + sl; s < s1; s += 4)
            for (int e = 0; e < 4; ++e) if (i4 + e < n) a[e] += ws[(size_t)s * n + i4 + e];
    }
    sm[sl][cl] = a;
    __syncthreads();
    if (sl == 0) {
        f32x4 t = sm[0][cl] + sm[1][cl] + sm[2][cl] + sm[3][cl];
        float* o = out + (size_t)blockIdx.y * n;
        if (i4 + 3 < n) {
            if (accumulate) t += *reinterpret_cast<const f32x4*>(o + i4);
            *reinterpret_cast<f32x4*>(o + i4) = t;
        } else {
            for (int e = 0; e < 4; ++e) if (i4 + e < n) o[i4 + e] = accumulate ? o[i4 + e] + t[e] : t[e];
        }
    }
}

// dW = sum of S slabs of n floats (ws is clobbered: the first-level partial sums are written over its first slabs' tail)
static void launch_slab_reduce(const float* ws, float* tmp, float* dw, long n, int S, int accumulate, hipStream_t st) {
    const unsigned nb = (unsigned)((n / 4 + 1 + 63) / 64);
    if (S > 64 && tmp != nullptr) {
        const int G = (S + 31) / 32;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(nb, G), dim3(256), 0, st, ws, tmp, n, S, 32, 0);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(nb, 1), dim3(256), 0, st, (const float*)tmp, dw, n, G, G, accumulate);
    } else {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(nb, 1), dim3(256), 0, st, ws, dw, n, S, S, accumulate);
    }
}

template <int BKT, int BNT, int WM, int WN, int PREC = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_fast_kernel(WgradParams q) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BKT / WM / 32, TN = BNT / WN / 32;
    constexpr int PT = 16;   // pixels per LDS stage: 16 keeps 4-5 blocks per CU resident (LDS 32 KB), which the 4-byte LDS operand reads need
    constexpr int AF4_PER_ROW = BKT / 4, A_RPP = NT / AF4_PER_ROW, A_F4 = (PT + A_RPP - 1) / A_RPP;
    constexpr int DF4_PER_ROW = BNT / 4, D_RPP = NT / DF4_PER_ROW, D_F4 = (PT + D_RPP - 1) / D_RPP;
    constexpr int A_SZ = PT * BKT, D_SZ = PT * BNT;
    const ConvParams& p = q.c;

    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + D_SZ)];
    float* Ap = smem;
    float* Dp = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid, XCD-aware: the blocks that share a pixel chunk (all K tiles x N tiles of one split) are consecutive
    // logical ids and therefore land on ONE XCD, so the chunk of x / dy is fetched into one L2 instead of eight
    const int nkb = (p.K + BKT - 1) / BKT, nnb = (p.Cout + BNT - 1) / BNT;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int bz = lb / (nkb * nnb), brem = lb - bz * (nkb * nnb);
    const int k0 = (brem % nkb) * BKT, n0 = (brem / nkb) * BNT;
    const int pbeg = bz * q.chunk;
    const int pend = min(p.M, pbeg + q.chunk);

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)q.dy, 0, p.M * p.Cout * 4, 0x00020000);

    // this lane's 4 consecutive k: fixed tap and channel quad
    const int a_kc = tid % AF4_PER_ROW, a_r0 = tid / AF4_PER_ROW;
    const int a_k = k0 + 4 * a_kc;
    const int Cin = p.C1 + p.C2;
    const bool k_ok = a_k < p.K;
    const int a_tap = k_ok ? a_k / Cin : 0, a_c = k_ok ? a_k - a_tap * Cin : 0;
    const int a_kh = a_tap / p.KW, a_kw = a_tap - a_kh * p.KW;
    const bool from1 = a_c < p.C1;
    const int a_cs = from1 ? a_c : a_c - p.C1;
    int r_b[A_F4], r_ho[A_F4], r_wo[A_F4];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        const int m = pbeg + a_r0 + j * A_RPP;
        const int b = m / HoWo, r = m - b * HoWo;
        r_b[j] = b; r_ho[j] = r / p.Wo; r_wo[j] = r - r_ho[j] * p.Wo;
    }
    const int d_nc = tid % DF4_PER_ROW, d_r0 = tid / DF4_PER_ROW;
    const int d_n = n0 + 4 * d_nc;
    const bool d_ok = d_n < p.Cout;

    f32x4 ra[A_F4], rd4[D_F4];
    auto load_stage = [&](int ps) {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int pr = a_r0 + j * A_RPP;
            const int hi = r_ho[j] * p.stride - p.pad_h + a_kh, wi = r_wo[j] * p.stride - p.pad_w + a_kw;
            const bool ok = k_ok && pr < PT && ps + pr < pend && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            int off;
            if (from1) off = p.ups ? ((r_b[j] * p.H1 + (hi >> 1)) * p.W1 + (wi >> 1)) * p.C1 + a_cs
                                   : ((r_b[j] * p.H + hi) * p.W + wi) * p.C1 + a_cs;
            else off = ((r_b[j] * p.H + hi) * p.W + wi) * p.C2 + a_cs;
            // descriptors must stay wave-uniform: with two inputs issue both loads, the unselected one out of range (zeros)
            if (p.C2 == 0) ra[j] = buf_load4(r1, ok ? off * 4 : BUF_OOB);
            else ra[j] = buf_load4(r1, (ok && from1) ? off * 4 : BUF_OOB) + buf_load4(r2, (ok && !from1) ? off * 4 : BUF_OOB);
            r_wo[j] += PT;
            while (r_wo[j] >= p.Wo) { r_wo[j] -= p.Wo; if (++r_ho[j] >= p.Ho) { r_ho[j] = 0; ++r_b[j]; } }
        }
#pragma unroll
        for (int j = 0; j < D_F4; ++j) {
            const int pr = d_r0 + j * D_RPP;
            const int m = ps + pr;
            rd4[j] = buf_load4(rd, (d_ok && pr < PT && m < pend) ? (m * p.Cout + d_n) * 4 : BUF_OOB);
        }
    };
    auto store_stage = [&](int buf) {
        float* A = Ap + buf * A_SZ;
        float* D = Dp + buf * D_SZ;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int pr = a_r0 + j * A_RPP;
            if (pr < PT) *reinterpret_cast<f32x4*>(&A[pr * BKT + 4 * a_kc]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < D_F4; ++j) {
            const int pr = d_r0 + j * D_RPP;
            if (pr < PT) *reinterpret_cast<f32x4*>(&D[pr * BNT + 4 * d_nc]) = rd4[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (pbeg < pend) {
        load_stage(pbeg);
        store_stage(0);
        __syncthreads();
        const int a_col = wm * (BKT / WM) + li, b_col = wn * (BNT / WN) + li;
        int cur = 0;
        for (int ps = pbeg; ps < pend; ps += PT) {
            const bool more = ps + PT < pend;
            if (more) load_stage(ps + PT);
            const float* A = Ap + cur * A_SZ;
            const float* D = Dp + cur * D_SZ;
            if constexpr (PREC != 0) {
                typedef typename LowPrec<PREC>::T LT;
                typedef typename LowPrec<PREC>::V8 LV8;
                // bf16 mode: the reduction index of this GEMM is the pixel, so an MFMA operand is 8 consecutive PIXELS of one
                // column -- read down the fp32 [pixel][column] tile (8 conflict-free 4-byte reads), round to bf16 in registers
#pragma unroll
                for (int s = 0; s < PT / 16; ++s) {
                    LV8 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[i][e] = (LT)A[(16 * s + 8 * lh + e) * BKT + a_col + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) b[j][e] = (LT)D[(16 * s + 8 * lh + e) * BNT + b_col + j * 32];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = LowPrec<PREC>::mfma(a[i], b[j], acc[i][j]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < PT / 2; ++s) {
                    float a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = A[(2 * s + lh) * BKT + a_col + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = D[(2 * s + lh) * BNT + b_col + j * 32];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
            if (more) store_stage(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    float* out = q.ws + (size_t)bz * p.K * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BNT / WN) + j * 32 + li;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * (BKT / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < p.K) out[(size_t)k * p.Cout + n] = acc[i][j][r];
            }
    }
}

// =====================================================================================
// Weight gradient, transposed-staging variant (the layers that carry the FLOPs: stride 1, output width a multiple of 4, channel
// counts multiples of 4).  Same GEMM as conv_wgrad_fast_kernel (dW[k, n] = sum over pixels of A[p, k] * dy[p, n]) but the LDS
// tiles are laid out like the forward kernel's -- one row per GEMM row / column holding 32 consecutive PIXELS (the reduction
// index), 16-byte chunks XOR-swizzled -- so an MFMA operand is one conflict-free ds_read_b128 feeding four MFMAs instead of
// four ds_read_b32.  The transposition [pixel][channel] (HBM) -> [channel][pixel] (LDS) costs nothing: a thread fetches a
// 4-pixel x 4-channel block with four 16-byte buffer loads and stores its four channel rows with four 16-byte LDS writes (the
// register "transpose" is a renaming).  A 32-pixel stage carries 64 MFMAs per wave, twice the old kernel's, and the kernel
// runs at the forward kernel's occupancy (2-3 blocks per CU) -- which is what lets the host split the pixels into far fewer
// slabs (mmseg_conv2d_wgrad picks the split from a small cost model): the slab write + re-read, 2.2x the algorithmic HBM
// bytes in round 1, shrinks accordingly.  Deterministic as before: fixed-order slab reduction, no float atomics.
// =====================================================================================
// LDS chunk swizzle of the fp32 tiles: conflict-free for the MFMA operand reads (32 consecutive rows, one chunk) AND for the
// transposing stores (8 lanes = rows 4 apart, one chunk)
__device__ __forceinline__ int wg_swz(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }
// LDS image of a 16-bit tile (64-byte rows = 32 pixels = four 16-byte slots).  ds_write_b64 resolves banks modulo 128 bytes over
// 16 consecutive lanes and ds_read_b128 modulo 256 bytes over the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}
// (MI355X_MICROARCH.md, LDS), so with plain 64-byte row pitch the transposing stores -- rows 4 (or 8) apart in one instruction --
// pile 16 lanes onto 4 slots (measured: 80-89 % of the LDS cycles were conflict cycles).  Instead the rows are dealt into RPC
// planes by (row mod RPC), RPC = the rows one thread stores (4: fp32 source, 8: 16-bit source), the plane pitch is padded by one
// row and the slot is XORed with row bits chosen so that BOTH access patterns are conflict-free (tools/lds_layout_check.py
// enumerates every lane group of every tile size).  Returns the slot of chunk 0 of `row`; `f` is the XOR term of the chunk index.
template <int TILE, int RPC>
__device__ __forceinline__ int lp_row_slot(int row, int& f) {
    if (RPC == 4) { f = (row >> 3) & 3; return (row & 3) * (TILE + 4) + (row >> 2) * 4; }
    f = ((row >> 3) ^ (row >> 5)) & 3;
    return (row & 7) * (TILE / 2 + 4) + (row >> 3) * 4;
}
// staging block `pair` of a 16-bit tile -> (channel group, pixel group): 16 consecutive lanes take 8 channel groups x 2 pixel
// groups (the two 8-byte halves of one slot), which is what makes their ds_write_b64 cover all 32 banks
template <int TILE, int RPC>
__device__ __forceinline__ void lp_pair(int pair, int& cq, int& pg) {
    constexpr int NCQ = TILE / RPC, CQL = NCQ >= 8 ? 8 : NCQ, PGL = 16 / CQL, G = NCQ / CQL;
    const int rest = pair >> 4;
    cq = (rest % G) * CQL + (pair / PGL) % CQL;
    pg = (rest / G) * PGL + pair % PGL;
}

// ANYW: any output width >= 4 and stride 1 or 2 (the discriminators' 4x4 layers: Wo = 62, 30, 27): the 4 pixels of a thread's block may
// straddle a row (and sample) boundary, so each pixel carries its own (sample, row, column) -- 3 more VALU per 16-byte load.
template <int BKT, int BNT, int WM, int WN, int PREC, bool TWO, bool ANYW>
__device__ __forceinline__ void conv_wgrad_tr_body(const WgradParams& q) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BKT / WM / 32, TN = BNT / WN / 32;
    constexpr int PT = 32;                               // pixels per LDS stage = K depth of one tile pair
    constexpr bool LP = PREC != 0;
    typedef typename LowPrec<PREC>::T LT;
    typedef typename LowPrec<PREC>::V4 LV4;
    typedef typename LowPrec<PREC>::V8 LV8;
    constexpr int LD = PT;                               // 4-byte words per LDS row (fp32 tiles; the 16-bit image: lp_row_slot)
    constexpr int A_SZ = LP ? (4 * BKT + 32) * 4 : BKT * LD, D_SZ = LP ? (4 * BNT + 32) * 4 : BNT * LD;
    constexpr int A_PAIRS = (BKT / 4) * (PT / 4), D_PAIRS = (BNT / 4) * (PT / 4);   // (channel quad, pixel group) blocks
    constexpr int A_IT = (A_PAIRS + NT - 1) / NT, D_IT = (D_PAIRS + NT - 1) / NT;
    const ConvParams& p = q.c;

    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + D_SZ)];
    float* Ap = smem;
    float* Dp = smem + 2 * A_SZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nkb = (p.K + BKT - 1) / BKT, nnb = (p.Cout + BNT - 1) / BNT;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);     // the tiles of one pixel chunk are consecutive -> one XCD's L2
    const int bz = lb / (nkb * nnb), brem = lb - bz * (nkb * nnb);
    const int k0 = (brem % nkb) * BKT, n0 = (brem / nkb) * BNT;
    const int pbeg = bz * q.chunk;                       // multiple of 32
    const int pend = min(p.M, pbeg + q.chunk);

    // 16-bit tensors in HBM (reduced-precision modes, mmseg_conv2d_wgrad_t): io bit 0 = x1 AND x2, bit 2 = dy hold elements of the
    // MFMA operand type -- they go to LDS without conversion
    const bool x16 = LP && (p.io & 1), d16 = LP && (p.io & 4);
    const int esx = x16 ? 2 : 4, esd = d16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * esx, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(TWO ? p.x2 : p.x1), 0,
                                                                        TWO ? p.B * p.H * p.W * p.C2 * esx : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)q.dy, 0, p.M * p.Cout * esd, 0x00020000);
    // 16 bytes at element offset `off`: raw bits of 4 floats, or (h) of 8 16-bit elements
    auto ldraw = [&](const __amdgpu_buffer_rsrc_t& r, int off, bool ok, bool h) -> u32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off * ((LP && h) ? 2 : 4) : BUF_OOB, 0, 0);
    };
    // channel e of each of 4 raw pixel loads -> the 4 pixel values of one LDS row segment, as the 16-bit operand type
    // (h: the loads hold 8 16-bit channels each, e in 0..7; else 4 floats each, e in 0..3)
    auto row16 = [&](const u32x4 (&px)[4], int e, bool h) -> LV4 {
        if constexpr (LP) {
            if (h) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const int w = e >> 1;
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;      // v_perm_b32: the low (high) halves of two dwords
                return __builtin_bit_cast(LV4, u32x2{__builtin_amdgcn_perm(px[1][w], px[0][w], sel),
                                                     __builtin_amdgcn_perm(px[3][w], px[2][w], sel)});
            }
        }
        e &= 3;
        const f32x4 f0 = __builtin_bit_cast(f32x4, px[0]), f1 = __builtin_bit_cast(f32x4, px[1]);
        const f32x4 f2 = __builtin_bit_cast(f32x4, px[2]), f3 = __builtin_bit_cast(f32x4, px[3]);
        return LV4{(LT)f0[e], (LT)f1[e], (LT)f2[e], (LT)f3[e]};
    };

    // ---- A side: this thread's (k quad, pixel group) blocks.  Tap, channel and source tensor are fixed per thread, so the
    // gather offset is  b * sb + (hi >> sh) * sr + (wi >> sh) * sp + c  with per-thread strides -- no branches in the loop ----
    const int Cin = p.C1 + p.C2;
    const int HoWo = p.Ho * p.Wo;
    const int q32 = PT / p.Wo, r32 = PT - q32 * p.Wo;    // a stage advances the pixel by 32 = q32 rows + r32 columns
    int a_row[A_IT], a_pg[A_IT], a_dh[A_IT], a_dw[A_IT], a_c[A_IT], a_sb[A_IT], a_sr[A_IT], a_sp[A_IT], a_sh[A_IT];
    int a_b[A_IT], a_ho[A_IT], a_wo[A_IT];
    bool a_ok[A_IT], a_from1[A_IT], a_act[A_IT];
    // x stored 16-bit: a lane's 16-byte load is 8 channels, so a thread's block is 4 pixels x 8 channels and the A tile needs only
    // BKT of the block's threads (0 .. BKT-1); the dy tile, when 16-bit too, is staged by the LAST BNT threads
    const int a_cw = x16 ? 8 : 4;                        // channels per load
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int pair = x16 ? tid : tid + it * NT;
        int cq = pair % (BKT / a_cw), pg = pair / (BKT / a_cw);
        if constexpr (LP) {
            if (x16) lp_pair<BKT, 8>(pair, cq, pg); else lp_pair<BKT, 4>(pair, cq, pg);
        }
        const int k = k0 + a_cw * cq;
        a_act[it] = x16 ? (it == 0 && tid < BKT) : (pair < A_PAIRS);
        a_ok[it] = a_act[it] && k < p.K;
        const int tap = a_ok[it] ? k / Cin : 0, c = a_ok[it] ? k - tap * Cin : 0;
        a_row[it] = a_cw * cq; a_pg[it] = pg;
        const int kh = tap / p.KW;
        a_dh[it] = kh - p.pad_h; a_dw[it] = tap - kh * p.KW - p.pad_w;
        const bool f1 = !TWO || c < p.C1;
        a_from1[it] = f1;
        a_c[it] = f1 ? c : c - p.C1;
        a_sp[it] = f1 ? p.C1 : p.C2;
        a_sr[it] = f1 ? p.W1 * p.C1 : p.W * p.C2;
        a_sb[it] = f1 ? p.H1 * p.W1 * p.C1 : p.H * p.W * p.C2;
        a_sh[it] = f1 ? p.ups : 0;
        const int m = pbeg + 4 * pg;                     // first of the 4 consecutive pixels (same output row: Wo % 4 == 0)
        const int b = m / HoWo, r = m - b * HoWo;
        a_b[it] = b; a_ho[it] = r / p.Wo; a_wo[it] = r - a_ho[it] * p.Wo;
    }
    int d_row[D_IT], d_pg[D_IT], d_off[D_IT];
    bool d_ok[D_IT], d_act[D_IT];
    const int d_cw = d16 ? 8 : 4;
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
        const int pair = d16 ? max(tid - (NT - BNT), 0) : tid + it * NT;
        int cq = pair % (BNT / d_cw), pg = pair / (BNT / d_cw);
        if constexpr (LP) {
            if (d16) lp_pair<BNT, 8>(pair, cq, pg); else lp_pair<BNT, 4>(pair, cq, pg);
        }
        d_row[it] = d_cw * cq; d_pg[it] = pg;
        d_act[it] = d16 ? (it == 0 && tid >= NT - BNT) : (pair < D_PAIRS);
        d_ok[it] = d_act[it] && n0 + d_cw * cq < p.Cout;
        d_off[it] = (pbeg + 4 * pg) * p.Cout + n0 + d_cw * cq;
    }

    struct WStage { u32x4 a[A_IT][4]; u32x4 d[D_IT][4]; };       // a stage in flight as raw bits
    // (a stage at or beyond `pend` loads nothing: every pixel test fails, the offsets are out of range)
    auto load_stage = [&](WStage& t, int ps) {
        u32x4 (&ra)[A_IT][4] = t.a;
        u32x4 (&rdv)[D_IT][4] = t.d;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int hi = a_ho[it] * p.stride + a_dh[it];
            const bool rowok = a_ok[it] && (unsigned)hi < (unsigned)p.H && ps + 4 * a_pg[it] < pend;
            const int wi0 = a_wo[it] * p.stride + a_dw[it];
            const int rowoff = a_b[it] * a_sb[it] + (hi >> a_sh[it]) * a_sr[it] + a_c[it];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int wi = wi0 + j * p.stride, roff = rowoff;
                bool rok = rowok;
                if constexpr (ANYW) {
                    int wo = a_wo[it] + j, ho = a_ho[it], b = a_b[it];
                    if (wo >= p.Wo) { wo -= p.Wo; ++ho; if (ho >= p.Ho) { ho -= p.Ho; ++b; } }
                    const int hj = ho * p.stride + a_dh[it];
                    wi = wo * p.stride + a_dw[it];
                    rok = a_ok[it] && (unsigned)hj < (unsigned)p.H && ps + 4 * a_pg[it] + j < pend;
                    roff = b * a_sb[it] + (hj >> a_sh[it]) * a_sr[it] + a_c[it];
                }
                const bool ok = rok && (unsigned)wi < (unsigned)p.W;
                const int off = roff + (wi >> a_sh[it]) * a_sp[it];
                // descriptors stay wave-uniform: with two inputs both loads are issued, the unselected one out of range (zero bits)
                if constexpr (!TWO) ra[it][j] = ldraw(r1, off, ok, x16);
                else ra[it][j] = ldraw(r1, off, ok && a_from1[it], x16) | ldraw(r2, off, ok && !a_from1[it], x16);
            }
            // advance the pixel group by one stage (32 pixels): q32 rows + r32 columns, one carry; the host guarantees Ho > q32
            a_wo[it] += r32; a_ho[it] += q32;
            if (a_wo[it] >= p.Wo) { a_wo[it] -= p.Wo; ++a_ho[it]; }
            if (a_ho[it] >= p.Ho) { a_ho[it] -= p.Ho; ++a_b[it]; }
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            const int m0 = ps + 4 * d_pg[it];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rdv[it][j] = ldraw(rd, d_off[it] + j * p.Cout, d_ok[it] && m0 + j < pend, d16);
            d_off[it] += PT * p.Cout;
        }
    };
    auto store_stage = [&](int buf, const WStage& t) {
        const u32x4 (&ra)[A_IT][4] = t.a;
        const u32x4 (&rdv)[D_IT][4] = t.d;
        float* A = Ap + buf * A_SZ;
        float* D = Dp + buf * D_SZ;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            if (!a_act[it]) continue;
#pragma unroll
            for (int e = 0; e < (LP ? 8 : 4); ++e) {
                if (e >= a_cw) break;
                const int row = a_row[it] + e;
                if constexpr (LP) {
                    int f;
                    const int s0 = x16 ? lp_row_slot<BKT, 8>(row, f) : lp_row_slot<BKT, 4>(row, f);
                    *reinterpret_cast<LV4*>(&A[4 * (s0 + ((a_pg[it] >> 1) ^ f)) + 2 * (a_pg[it] & 1)]) = row16(ra[it], e, x16);
                } else {
                    const u32x4 v = {ra[it][0][e], ra[it][1][e], ra[it][2][e], ra[it][3][e]};
                    *reinterpret_cast<u32x4*>(&A[row * LD + 4 * (a_pg[it] ^ wg_swz(row))]) = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            if (!d_act[it]) continue;
#pragma unroll
            for (int e = 0; e < (LP ? 8 : 4); ++e) {
                if (e >= d_cw) break;
                const int row = d_row[it] + e;
                if constexpr (LP) {
                    int f;
                    const int s0 = d16 ? lp_row_slot<BNT, 8>(row, f) : lp_row_slot<BNT, 4>(row, f);
                    *reinterpret_cast<LV4*>(&D[4 * (s0 + ((d_pg[it] >> 1) ^ f)) + 2 * (d_pg[it] & 1)]) = row16(rdv[it], e, d16);
                } else {
                    const u32x4 v = {rdv[it][0][e], rdv[it][1][e], rdv[it][2][e], rdv[it][3][e]};
                    *reinterpret_cast<u32x4*>(&D[row * LD + 4 * (d_pg[it] ^ wg_swz(row))]) = v;
                }
            }
        }
    };

    constexpr int NACC = (TM * TN >= 4) ? 1 : 4 / (TM * TN);
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

    if (pbeg < pend) {
        const int ar = wm * (BKT / WM) + li, br = wn * (BNT / WN) + li;
        int a_s0[TM], a_f[TM], b_s0[TN], b_f[TN];        // 16-bit image: slot of chunk 0 + XOR term of this lane's operand rows
        if constexpr (LP) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a_s0[i] = x16 ? lp_row_slot<BKT, 8>(ar + i * 32, a_f[i]) : lp_row_slot<BKT, 4>(ar + i * 32, a_f[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b_s0[j] = d16 ? lp_row_slot<BNT, 8>(br + j * 32, b_f[j]) : lp_row_slot<BNT, 4>(br + j * 32, b_f[j]);
        }
        auto mma_stage = [&](int cur) {
            const float* A = Ap + cur * A_SZ;
            const float* D = Dp + cur * D_SZ;
            if constexpr (LP) {
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {         // two steps of 16 pixels; lane half lh supplies pixels 16 qq + 8 lh + [0, 8)
                    LV8 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i] = *reinterpret_cast<const LV8*>(&A[4 * (a_s0[i] + ((2 * qq + lh) ^ a_f[i]))]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[j] = *reinterpret_cast<const LV8*>(&D[4 * (b_s0[j] + ((2 * qq + lh) ^ b_f[j]))]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[qq % NACC][i][j] = LowPrec<PREC>::mfma(a[i], b[j], acc[qq % NACC][i][j]);
                }
            } else {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {         // lane half lh supplies pixels 8 qq + 4 lh + [0, 4)
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i] = *reinterpret_cast<const f32x4*>(&A[(ar + i * 32) * LD + 4 * ((2 * qq + lh) ^ wg_swz(ar + i * 32))]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[j] = *reinterpret_cast<const f32x4*>(&D[(br + j * 32) * LD + 4 * ((2 * qq + lh) ^ wg_swz(br + j * 32))]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[t % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[t % NACC][i][j], 0, 0, 0);
                }
            }
        };
        if constexpr (LP) {
            // 16-bit MFMAs finish a 32-pixel stage in a quarter of the fp32 time: the loads run TWO stages ahead (two register
            // sets, the loop unrolled by two so that their names are static and the compiler's vmcnt waits are exact)
            WStage t0, t1;
            load_stage(t0, pbeg);
            load_stage(t1, pbeg + PT);
            store_stage(0, t0);
            __syncthreads();
            int ps = pbeg;
            for (; ps + PT < pend; ps += 2 * PT) {
                load_stage(t0, ps + 2 * PT);             // LDS[0] = stage ps, t1 = stage ps + PT (in flight)
                mma_stage(0);
                store_stage(1, t1);
                __syncthreads();
                load_stage(t1, ps + 3 * PT);             // LDS[1] = stage ps + PT, t0 = stage ps + 2 PT (in flight)
                mma_stage(1);
                store_stage(0, t0);                      // (zeros when that stage lies beyond the chunk)
                __syncthreads();
            }
            if (ps < pend) mma_stage(0);                 // odd stage count: the last one sits in LDS[0]
        } else {
            WStage t0;
            load_stage(t0, pbeg);
            store_stage(0, t0);
            __syncthreads();
            int cur = 0;
            for (int ps = pbeg; ps < pend; ps += PT) {
                const bool more = ps + PT < pend;
                if (more) load_stage(t0, ps + PT);       // buffer loads in flight under the MFMAs
                mma_stage(cur);
                if (more) store_stage(cur ^ 1, t0);
                __syncthreads();
                cur ^= 1;
            }
        }
    }
#pragma unroll
    for (int s = 1; s < NACC; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][i][j] += acc[s][i][j];

    float* out = q.ws + (size_t)bz * p.K * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BNT / WN) + j * 32 + li;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * (BKT / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < p.K) out[(size_t)k * p.Cout + n] = acc[0][i][j][r];
            }
    }
}

template <int BKT, int BNT, int WM, int WN, int PREC = 0, bool TWO = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_tr_kernel(WgradParams q) {
    conv_wgrad_tr_body<BKT, BNT, WM, WN, PREC, TWO, false>(q);
}
template <int BKT, int BNT, int WM, int WN, int PREC = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_tr_anyw_kernel(WgradParams q) {
    conv_wgrad_tr_body<BKT, BNT, WM, WN, PREC, false, true>(q);
}

template <int BKT, int BNT, int WM, int WN, bool TWO>
static void launch_wgrad_tr_prec(const WgradParams& q, dim3 grid, dim3 block, hipStream_t st) {
    if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_wgrad_tr_kernel<BKT, BNT, WM, WN, 1, TWO>), grid, block, 0, st, q);
    else if (g_conv_bf16 == 2) hipLaunchKernelGGL((conv_wgrad_tr_kernel<BKT, BNT, WM, WN, 2, TWO>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_wgrad_tr_kernel<BKT, BNT, WM, WN, 0, TWO>), grid, block, 0, st, q);
}
template <int BKT, int BNT, int WM, int WN>
static int launch_wgrad_tr(const WgradParams& q, int S, hipStream_t st) {
    dim3 grid(((q.c.K + BKT - 1) / BKT) * ((q.c.Cout + BNT - 1) / BNT) * S), block(WM * WN * 64);
    MMSEG_SET_LAST(6, BKT, BNT);
    if (q.c.C2 > 0) g_last_kernel += 500000;            // the two-input instantiation
    if (q.c.C2 > 0) launch_wgrad_tr_prec<BKT, BNT, WM, WN, true>(q, grid, block, st);
    else launch_wgrad_tr_prec<BKT, BNT, WM, WN, false>(q, grid, block, st);
    return MMSEG_CHECK_LAUNCH();
}

// the any-width instances (one input tensor)
template <int BKT, int BNT, int WM, int WN>
static int launch_wgrad_tr_anyw(const WgradParams& q, int S, hipStream_t st) {
    dim3 grid(((q.c.K + BKT - 1) / BKT) * ((q.c.Cout + BNT - 1) / BNT) * S), block(WM * WN * 64);
    MMSEG_SET_LAST(14, BKT, BNT);
    if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_wgrad_tr_anyw_kernel<BKT, BNT, WM, WN, 1>), grid, block, 0, st, q);
    else if (g_conv_bf16 == 2) hipLaunchKernelGGL((conv_wgrad_tr_anyw_kernel<BKT, BNT, WM, WN, 2>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_wgrad_tr_anyw_kernel<BKT, BNT, WM, WN, 0>), grid, block, 0, st, q);
    return MMSEG_CHECK_LAUNCH();
}

template <int BKT, int BNT, int WM, int WN>
static int launch_wgrad_fast(const WgradParams& q, int S, hipStream_t st) {
    dim3 grid(((q.c.K + BKT - 1) / BKT) * ((q.c.Cout + BNT - 1) / BNT) * S), block(WM * WN * 64);
    MMSEG_SET_LAST(7, BKT, BNT);
    if (g_conv_bf16 == 1) hipLaunchKernelGGL((conv_wgrad_fast_kernel<BKT, BNT, WM, WN, 1>), grid, block, 0, st, q);
    else if (g_conv_bf16 == 2) hipLaunchKernelGGL((conv_wgrad_fast_kernel<BKT, BNT, WM, WN, 2>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_wgrad_fast_kernel<BKT, BNT, WM, WN>), grid, block, 0, st, q);
    return MMSEG_CHECK_LAUNCH();
}

template <int BKT, int BNT, int WM, int WN>
static int launch_wgrad(const WgradParams& q, int S, bool vec, hipStream_t st) {
    dim3 grid((q.c.K + BKT - 1) / BKT, (q.c.Cout + BNT - 1) / BNT, S), block(WM * WN * 64);
    MMSEG_SET_LAST(8, BKT, BNT);
    if (vec) g_last_kernel += 500000;
    if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<BKT, BNT, WM, WN, true>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((conv_wgrad_kernel<BKT, BNT, WM, WN, false>), grid, block, 0, st, q);
    return MMSEG_CHECK_LAUNCH();
}

// flip + transpose: wt[kh][kw][co][ci] = w[KH-1-kh][KW-1-kw][ci][co]
__global__ void wflip_kernel(const float* __restrict__ w, float* __restrict__ wt, int KH, int KW, int Cin, int Cout) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z, kh = tap / KW, kw = tap % KW;
    const float* src = w + (size_t)((KH - 1 - kh) * KW + (KW - 1 - kw)) * Cin * Cout;
    float* dst = wt + (size_t)tap * Cin * Cout;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        t[r][tx] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = t[tx][r];
    }
}

// element i of a fast-path weight image: fp32, or -- in the reduced-precision modes -- the MFMA operand type itself (rounded once
// here instead of on every staging pass of every block; the image then occupies the first half of the fp32-sized buffer)
__device__ __forceinline__ void wimg_store(float* out, size_t i, float v, int prec) {
    if (prec == 1) reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
    else if (prec == 2) reinterpret_cast<_Float16*>(out)[i] = (_Float16)v;
    else out[i] = v;
}
// mode 0 as a tiled transpose: per tap the [Cin][Cout] matrix goes to rows co of out[Cout][ntaps * Cin] at column tap*Cin + ci;
// 32 x 32 tiles through LDS, so both the reads (along co) and the writes (along ci) are coalesced
__global__ void wprep_fwd_tiled_kernel(const float* __restrict__ w, float* __restrict__ out, int ntaps, int Cin, int Cout, int prec) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z;
    const float* src = w + (size_t)tap * Cin * Cout;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        t[r][tx] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
    }
    __syncthreads();
    const size_t K = (size_t)ntaps * Cin;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) wimg_store(out, (size_t)co * K + (size_t)tap * Cin + ci, t[tx][r], prec);
    }
}

// weight re-layouts for the fast path: out[n][tap'][c] with n the GEMM column (output channel of the launch)
//   mode 0 (forward)      : out[co][tap][ci] = w[tap][ci][co]
//   mode 1 (data gradient): out[ci][tap][co] = w[ntaps-1-tap][ci][co]
//   mode 2 (as is)        : out[i] = w[i] -- a kernel whose Keras layout already is the image (the all-taps 1x1 GEMM of the small-channel
//                           data gradient), converted to the image's element type
__global__ void wprep_kernel(const float* __restrict__ w, float* __restrict__ out, int ntaps, int Cin, int Cout, int mode, int prec) {
    const long n = (long)ntaps * Cin * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (mode == 2) {
            wimg_store(out, i, w[i], prec);
        } else if (mode == 0) {
            const int ci = i % Cin; long r = i / Cin;
            const int tap = r % ntaps; const int co = r / ntaps;
            wimg_store(out, i, w[((long)tap * Cin + ci) * Cout + co], prec);
        } else {
            const int co = i % Cout; long r = i / Cout;
            const int tap = r % ntaps; const int ci = r / ntaps;
            wimg_store(out, i, w[((long)(ntaps - 1 - tap) * Cin + ci) * Cout + co], prec);
        }
    }
}

// sub-kernel of parity class (ph, pw): out[ci][(th, tw)][co] = w[ph + s*(TH-1-th)][pw + s*(TW-1-tw)][ci][co]
__global__ void wprep_parity_kernel(const float* __restrict__ w, float* __restrict__ out, int KH, int KW, int Cin, int Cout,
                                    int TH, int TW, int s, int ph, int pw, int prec) {
    const long n = (long)TH * TW * Cin * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = i % Cout; long r = i / Cout;
        const int tw = r % TW; r /= TW;
        const int th = r % TH; const int ci = r / TH;
        const int kh = ph + s * (TH - 1 - th), kw = pw + s * (TW - 1 - tw);
        wimg_store(out, i, w[(((long)kh * KW + kw) * Cin + ci) * Cout + co], prec);
    }
}

// Several weight images in ONE launch (all the forward / data-gradient images of one model after its optimiser step: ~80 launches of
// 4-7 us per iteration otherwise).  table: n + 1 rows of 8 int64 {src, dst, ntaps, Cin, Cout, mode, first block, -}; a block finds its row
// by bisection over the first-block column (row n holds the total), then handles one 32 x 32 (ci, co) tile of one tap:
//   mode 0: out[co][tap][ci] = w[tap][ci][co]  (transpose through LDS, as wprep_fwd_tiled_kernel)
//   mode 1: out[ci][tap][co] = w[ntaps-1-tap][ci][co]  (row copy)
__global__ __launch_bounds__(256) void wprep_batch_kernel(const long long* __restrict__ table, int n, int prec) {
    __shared__ float t[32][33];
    const long long b = blockIdx.x;
    int lo = 0, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (table[(size_t)mid * 8 + 6] <= b) lo = mid; else hi = mid; }
    const long long* r = table + (size_t)lo * 8;
    const float* w = reinterpret_cast<const float*>(r[0]);
    float* out = reinterpret_cast<float*>(r[1]);
    const int ntaps = (int)r[2], Cin = (int)r[3], Cout = (int)r[4], mode = (int)r[5];
    int l = (int)(b - r[6]);
    const int ntx = (Cout + 31) / 32, nty = (Cin + 31) / 32;
    const int co0 = (l % ntx) * 32; l /= ntx;
    const int ci0 = (l % nty) * 32, tap = l / nty;
    if (tap >= ntaps) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (mode == 0) {
        const float* src = w + (size_t)tap * Cin * Cout;
        for (int rr = ty; rr < 32; rr += 8) {
            const int ci = ci0 + rr, co = co0 + tx;
            t[rr][tx] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
        }
        __syncthreads();
        const size_t K = (size_t)ntaps * Cin;
        for (int rr = ty; rr < 32; rr += 8) {
            const int co = co0 + rr, ci = ci0 + tx;
            if (co < Cout && ci < Cin) wimg_store(out, (size_t)co * K + (size_t)tap * Cin + ci, t[tx][rr], prec);
        }
    } else {
        const float* src = w + (size_t)(ntaps - 1 - tap) * Cin * Cout;
        for (int rr = ty; rr < 32; rr += 8) {
            const int ci = ci0 + rr, co = co0 + tx;
            if (ci < Cin && co < Cout) wimg_store(out, ((size_t)ci * ntaps + tap) * Cout + co, src[(size_t)ci * Cout + co], prec);
        }
    }
}

// all s x s parity classes in one launch: grid.y = class (ph * s + pw); the classes' images lie back to back in raster order
__global__ void wprep_parity_all_kernel(const float* __restrict__ w, float* __restrict__ out, int KH, int KW, int Cin, int Cout, int s, int prec) {
    const int ph = blockIdx.y / s, pw = blockIdx.y % s;
    long off = 0;
    for (int c = 0; c < (int)blockIdx.y; ++c) {
        const int qh = c / s, qw = c % s;
        const int th = qh < KH ? (KH - qh + s - 1) / s : 0, tw = qw < KW ? (KW - qw + s - 1) / s : 0;
        off += (long)th * tw * Cin * Cout;
    }
    const int TH = ph < KH ? (KH - ph + s - 1) / s : 0, TW = pw < KW ? (KW - pw + s - 1) / s : 0;
    const long n = (long)TH * TW * Cin * Cout;
    float* o = out + off;                     // class images start at fp32-element offsets in every precision (as mmseg_conv2d_dgrad_parity_all reads them)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int co = i % Cout; long r = i / Cout;
        const int tw = r % TW; r /= TW;
        const int th = r % TH; const int ci = r / TH;
        const int kh = ph + s * (TH - 1 - th), kw = pw + s * (TW - 1 - tw);
        wimg_store(o, i, w[(((long)kh * KW + kw) * Cin + ci) * Cout + co], prec);
    }
}

#include "wgrad32h.hpp"

extern "C" {
// Conv2D(Cout, 3, 'same') of an 8-channel tensor in the 16-bit modes (conv8h_kernel, conv16.hpp): x [B,H,W,8] fp32 (hx 0) or the 16-bit type of
// the active mode (hx = mode), w the Keras kernel [3,3,8,Cout] fp32, y [B,H,W,Cout] fp32 (hy 0) or 16-bit.  W % 32 == 0, Cout % 4 == 0.
int mmseg_conv8h_fwd_t(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, int act, float alpha,
                       int hx, int hy, void* stream) {
    const int prec = g_conv_bf16;
    if (prec == 0 || B <= 0 || H <= 0 || W <= 0 || (W & 31) || (Cout & 3) || act < 0 || act > 2) return (int)hipErrorInvalidValue;
    if ((hx != 0 && hx != prec) || (hy != 0 && hy != prec)) return (int)hipErrorInvalidValue;
    if ((long)B * H * W * 8 * 4 >= (1L << 31) - 64) return (int)hipErrorInvalidValue;
    const uintptr_t al = hy ? 7 : 15;
    if ((reinterpret_cast<uintptr_t>(y) & al) || (reinterpret_cast<uintptr_t>(x) & 15)) return (int)hipErrorInvalidValue;
    if (hy && (Cout & 7)) return (int)hipErrorInvalidValue;        // (16-byte stores of a 16-bit output)
    const int ntn = (Cout + 127) / 128;
    const long items = (long)B * H * (W / 32) * ntn;
    long blocks = (items + 3) / 4;
    if (blocks > 512) blocks = 512;                   // 2 blocks of 4 waves per CU; a wave walks its items with a stride
    if (blocks * 4 < ntn) blocks = (ntn + 3) / 4;     // (every column group gets a wave)
    hipStream_t st = (hipStream_t)stream;
    const float slope = act == 2 ? alpha : 1.f;
#define L8(P, X, Y, R) hipLaunchKernelGGL((conv8h_kernel<P, 4, X, Y, R>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, y, B, H, W, Cout, slope)
#define L8R(P, X, Y) do { if (act == 1) L8(P, X, Y, true); else L8(P, X, Y, false); } while (0)
#define L8P(P) do { if (hx) { if (hy) L8R(P, true, true); else L8R(P, true, false); } else { if (hy) L8R(P, false, true); else L8R(P, false, false); } } while (0)
    if (prec == 1) L8P(1); else L8P(2);
#undef L8R
#undef L8P
#undef L8
    MMSEG_SET_LAST(20, (hx ? 1 : 0) | (hy ? 2 : 0) | (act == 1 ? 4 : 0), 128);      // (the M-tile field carries the instance flags)
    return MMSEG_CHECK_LAUNCH();
}
// large-tile 16-bit kernel (conv16.hpp): 0 off, 1 where it pays (default), 2 wherever it applies; returns the previous mode
int mmseg_conv16_mode(int mode) {
    const int old = g_conv16_mode;
    if (mode >= 0 && mode <= 2) g_conv16_mode = mode;
    return old;
}


// Process-wide precision of the fast-path convolutions (forward, data gradient, weight gradient): 0 = fp32 MFMA, 1 = bf16,
// 2 = fp16 operands with fp32 accumulation (BASELINE configs #3 / #5).  Returns the previous mode.  Not a per-launch argument: the trainers switch it once.
int mmseg_set_conv_precision(int mode) {
    const int old = g_conv_bf16;
    g_conv_bf16 = (mode == 1 || mode == 2) ? mode : 0;
    return old;
}
int mmseg_get_conv_precision(void) { return g_conv_bf16; }
// family * 1000000 + (M or K tile) * 1000 + N tile of the kernel template the last convolution entry point launched (see
// g_last_kernel) -- lets a profiler harness label its per-launch timings with the kernel names rocprofv3 reports
int mmseg_conv2d_last_kernel(void) { return g_last_kernel; }

// Geometry arrays are plain ints so the ABI stays free of C++ types (see include/mmseg_hip.h).
static int conv2d_fwd_impl(const float* x1, const float* x2, const float* w, const float* wt, const float* bias, const float* oscale,
                           float* y, float* y2,
                           int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                           int pad_h, int pad_w, int ups, int transposed, int act, float alpha, int nsplit1,
                           int oH, int oW, int osh, int osw, int ooh, int oow, void* stream, int io = 0) {
    ConvParams p;
    p.oH = oH; p.oW = oW; p.osh = osh; p.osw = osw; p.ooh = ooh; p.oow = oow;
    p.x1 = x1; p.x2 = x2; p.w = w; p.wt = wt; p.bias = bias; p.oscale = oscale; p.y = y; p.y2 = y2;
    p.B = B; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2;
    p.H1 = ups ? H / 2 : H; p.W1 = ups ? W / 2 : W;
    p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride;
    p.pad_h = pad_h; p.pad_w = pad_w; p.ups = ups; p.transposed = transposed; p.act = act; p.alpha = alpha;
    p.M = B * Ho * Wo; p.K = KH * KW * (C1 + C2); p.nsplit1 = nsplit1;
    p.io = io;
    if (io != 0 && g_conv_bf16 == 0) return (int)hipErrorInvalidValue;      // 16-bit tensors exist in the reduced-precision modes only
    if (io != 0 && g_conv_bf16 == 2) p.io |= 8;
    if (C2 > 0 && ((io & 1) != ((io >> 1) & 1))) return (int)hipErrorInvalidValue;       // two inputs: both fp32 or both 16-bit
    if ((io & 1) && ((C1 & 7) || (C2 & 7))) return (int)hipErrorInvalidValue;             // a lane's 16-byte load = 8 channels
    if (ups && ((H & 1) || (W & 1))) return (int)hipErrorInvalidValue;
    if (C2 > 0 && x2 == nullptr) return (int)hipErrorInvalidValue;
    if ((long)B * Ho * Wo >= (1L << 31)) return (int)hipErrorInvalidValue;
    if (osh < 1 || osw < 1 || (long)(Ho - 1) * osh + ooh >= oH || (long)(Wo - 1) * osw + oow >= oW) return (int)hipErrorInvalidValue;
    return conv_dispatch(p, (hipStream_t)stream);
}
int mmseg_conv2d_fwd(const float* x1, const float* x2, const float* w, const float* wt, const float* bias, float* y, float* y2,
                     int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                     int pad_h, int pad_w, int ups, int transposed, int act, float alpha, int nsplit1, void* stream) {
    return conv2d_fwd_impl(x1, x2, w, wt, bias, nullptr, y, y2, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad_h, pad_w, ups,
                           transposed, act, alpha, nsplit1, Ho, Wo, 1, 1, 0, 0, stream);
}
// y = act(conv(x) * oscale[c] + bias[c]): a convolution followed by an inference-mode BatchNormalization (+ReLU) in one
// launch -- `predict` of the UNet / segmentor conv blocks (models/unet.py:94-101 with moving statistics); oscale / bias
// from mmseg_bn_infer_fold.  Same operands as mmseg_conv2d_fwd otherwise (stride-1 or strided, fused up-sampling / concat).
int mmseg_conv2d_fwd_scaled(const float* x1, const float* x2, const float* w, const float* wt, const float* bias, const float* oscale,
                            float* y, int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                            int pad_h, int pad_w, int ups, int act, float alpha, void* stream) {
    return conv2d_fwd_impl(x1, x2, w, wt, bias, oscale, y, nullptr, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad_h, pad_w, ups,
                           0, act, alpha, 0, Ho, Wo, 1, 1, 0, 0, stream);
}
// The same two entry points with 16-bit tensors in HBM (reduced-precision modes: mmseg_set_conv_precision(1 | 2) selects bf16 |
// fp16): io bit 0 = x1, bit 1 = x2 are 16-bit (MFMA fast path only: Cin % 32 == 0), bit 2 = y (and y2) is written 16-bit.
// Arithmetic as before (16-bit MFMA operands, fp32 accumulation, fp32 bias / scale / activation).
int mmseg_conv2d_fwd_t(const void* x1, const void* x2, const float* w, const float* wt, const float* bias, void* y, void* y2,
                       int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad_h, int pad_w, int ups, int transposed, int act, float alpha, int nsplit1, int io, void* stream) {
    return conv2d_fwd_impl((const float*)x1, (const float*)x2, w, wt, bias, nullptr, (float*)y, (float*)y2, B, H, W, C1, C2, Ho, Wo, Cout,
                           KH, KW, stride, pad_h, pad_w, ups, transposed, act, alpha, nsplit1, Ho, Wo, 1, 1, 0, 0, stream, io & 7);
}
int mmseg_conv2d_fwd_scaled_t(const void* x1, const void* x2, const float* w, const float* wt, const float* bias, const float* oscale,
                              void* y, int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                              int pad_h, int pad_w, int ups, int act, float alpha, int io, void* stream) {
    return conv2d_fwd_impl((const float*)x1, (const float*)x2, w, wt, bias, oscale, (float*)y, nullptr, B, H, W, C1, C2, Ho, Wo, Cout,
                           KH, KW, stride, pad_h, pad_w, ups, 0, act, alpha, 0, Ho, Wo, 1, 1, 0, 0, stream, io & 7);
}
// One parity class (ph, pw) of the data gradient of a stride-s convolution (s = 2), exact taps only:
//   dx[b, s*i + ph, s*j + pw, :] = sum_{a,b2} dy[b, i - a, j - b2, :] . W[ph + s*a, pw + s*b2, :, :]^T
// run as a stride-1 convolution over dy with the TH x TW sub-kernel `wt` (mmseg_conv2d_wprep_parity, fast layout) and a
// strided store into dx [B, H, W, Cin].  Needs Cout % 32 == 0 and Cin % 4 == 0 (mmseg_conv2d_fast_path(Cout,0,Cin,0)).
int mmseg_conv2d_dgrad_parity(const float* dy, const float* wt, float* dx, int B, int Ho, int Wo, int Cout, int H, int W, int Cin,
                              int TH, int TW, int stride, int ph, int pw, void* stream) {
    const int Hs = (H - ph + stride - 1) / stride, Ws = (W - pw + stride - 1) / stride;
    if (Hs <= 0 || Ws <= 0) return 0;
    return conv2d_fwd_impl(dy, nullptr, nullptr, wt, nullptr, nullptr, dx, nullptr, B, Ho, Wo, Cout, 0, Hs, Ws, Cin, TH, TW, 1, TH - 1, TW - 1,
                           0, 0, 0, 0.f, 0, H, W, stride, stride, ph, pw, stream);
}

// dx [B,H,W,Cin] = data gradient of a 4x4 stride-2 'valid' convolution with Cin in {1, 4} and Cout = 64 (the first layer of
// D_Image / D_Mask, models/discriminator.py:24); dy [B,Ho,Wo,64], w [4,4,Cin,64] in the Keras layout (no re-layout needed)
int mmseg_conv2d_dgrad_s2k4_smallc(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                                   void* stream) {
    if (Cout != 64 || (Cin != 1 && Cin != 4) || Ho != (H - 4) / 2 + 1 || Wo != (W - 4) / 2 + 1 || H < 4 || W < 4 ||
        !aligned16(dy) || !aligned16(w))
        return (int)hipErrorInvalidValue;
    const long groups = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
    const dim3 grid((unsigned)((groups + 15) / 16)), block(256);
    MMSEG_SET_LAST(5, Cin, 64);
    if (Cin == 1) hipLaunchKernelGGL(conv_dgrad_s2k4_smallc_kernel<1>, grid, block, 0, (hipStream_t)stream, dy, w, dx, B, H, W, Ho, Wo);
    else hipLaunchKernelGGL(conv_dgrad_s2k4_smallc_kernel<4>, grid, block, 0, (hipStream_t)stream, dy, w, dx, B, H, W, Ho, Wo);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_conv2d_parity_taps(int K, int stride, int p);
// All stride x stride parity classes in ONE launch.  `wt_all` holds the classes' sub-kernels back to back in (ph, pw)
// raster order, class (ph, pw) being the TH(ph) x TW(pw) x Cout x Cin block written by mmseg_conv2d_wprep_parity
// (TH(p) = mmseg_conv2d_parity_taps(KH, stride, p)); classes without taps or pixels are skipped (their dx stays as is:
// the caller zero-fills dx when some class has no taps, i.e. when KH < stride or KW < stride).
int mmseg_conv2d_dgrad_parity_all(const float* dy, const float* wt_all, float* dx, int B, int Ho, int Wo, int Cout, int H, int W,
                                  int Cin, int KH, int KW, int stride, void* stream) {
    if (stride != 2 || Cout % 32 != 0 || Cin % 4 != 0) return (int)hipErrorInvalidValue;
    ConvBatch pb;
    int n = 0;
    long woff = 0, maxM = 0;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            const int TH = mmseg_conv2d_parity_taps(KH, stride, ph), TW = mmseg_conv2d_parity_taps(KW, stride, pw);
            const int Hs = (H - ph + stride - 1) / stride, Ws = (W - pw + stride - 1) / stride;
            const long wsz = (long)TH * TW * Cout * Cin;
            if (TH > 0 && TW > 0 && Hs > 0 && Ws > 0) {
                ConvParams& p = pb.p[n++];
                p.oH = H; p.oW = W; p.osh = stride; p.osw = stride; p.ooh = ph; p.oow = pw;
                p.x1 = dy; p.x2 = nullptr; p.w = nullptr; p.wt = wt_all + woff; p.bias = nullptr; p.oscale = nullptr; p.y = dx; p.y2 = nullptr;
                p.B = B; p.H = Ho; p.W = Wo; p.C1 = Cout; p.C2 = 0; p.H1 = Ho; p.W1 = Wo;
                p.Ho = Hs; p.Wo = Ws; p.Cout = Cin; p.KH = TH; p.KW = TW; p.stride = 1;
                p.pad_h = TH - 1; p.pad_w = TW - 1; p.ups = 0; p.transposed = 0; p.act = 0; p.alpha = 0.f;
                p.M = B * Hs * Ws; p.K = TH * TW * Cout; p.nsplit1 = 0; p.io = 0;
                p.qepi = quad_epilogue_ok(p);
                if (p.M > maxM) maxM = p.M;
                if (((uintptr_t)p.wt & 15) != 0) return (int)hipErrorInvalidValue;
            }
            woff += wsz;
        }
    if (n == 0) return 0;
    if (((uintptr_t)dy & 15) != 0 || (long)B * Ho * Wo * Cout * 4 >= (1L << 31) - 64) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const long tiles_big = ((maxM + 127) / 128) * ((Cin + 127) / 128) * n;
    if (Cin > 64 && tiles_big >= 384) return launch_fast_batched<128, 128, 2, 2>(pb, n, st);
    if (Cin > 32) {
        const long tiles_mid = ((maxM + 127) / 128) * ((Cin + 63) / 64) * n;
        if (tiles_mid >= 384) return launch_fast_batched<128, 64, 2, 2>(pb, n, st);
        return launch_fast_batched<64, 64, 2, 2>(pb, n, st);
    }
    return launch_fast_batched<128, 32, 4, 1>(pb, n, st);
}

// number of floats of workspace mmseg_conv2d_wgrad needs for this geometry (0: writes dW directly)
// number of pixel splits (slabs) of the weight gradient: enough blocks to fill the chip, at least 512 pixels each
static int wgrad_target_blocks() {
    static int v = -1;
    if (v < 0) { v = ab_int("MMSEG_WGRAD_BLOCKS", 3072); if (v < 1) v = 3072; }
    return v;
}
static int wgrad_splits(long M, long K, int Cout) {
    const int bnt = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);          // the N tile the dispatcher below picks
    const long tiles = ((K + 127) / 128) * ((Cout + bnt - 1) / bnt);
    const long target = wgrad_target_blocks();
    long S = (target + tiles - 1) / tiles;
    const long maxS = (M + 511) / 512;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return (int)S;
}
// ---- pixel split of the transposed-staging kernel (conv_wgrad_tr_kernel) ----------------------------------------------
// Fewer, longer blocks: pick the number of slabs S that minimises a small cost model -- rounds of resident blocks (2-4 per
// CU by tile, the last round possibly partial and then slower per block) plus the slab write + re-read at HBM speed.
static int wgrad_tr_bnt(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }
static int wgrad_tr_enabled() {
    static int v = -1;
    if (v < 0) v = ab_int("MMSEG_WGRAD_TR", 1);
    return v;
}
// K tile: 192 rows (3 taps of 64 channels) when that divides K exactly and 128 does not -- the 64-channel 3x3 layers (K = 576),
// which would otherwise pad K by 11 %
static int wgrad_tr_bkt(long K, int Cout) { return (K % 128 != 0 && K % 192 == 0 && Cout > 32) ? 192 : 128; }
static int wgrad_tr_splits(long M, long K, int Cout, int* chunk_out) {
    const int bnt = wgrad_tr_bnt(Cout), bkt = wgrad_tr_bkt(K, Cout);
    const int bpc = bkt == 192 ? 2 : (bnt == 128 ? 2 : (bnt == 64 ? 3 : 4));   // resident blocks per CU (LDS 80 or 64 / 64 / 48 / 40 KB)
    const long tiles = ((K + bkt - 1) / bkt) * ((Cout + bnt - 1) / bnt);
    const double cu_rate = 4.0 * 64.0 * 2.4e9;                         // FLOP/s of one CU's fp32 MFMA pipes
    static const double eff[5] = {0.0, 0.55, 0.75, 0.82, 0.85};       // MFMA efficiency by resident blocks per CU (measured shape)
    static int forced = -2;
    if (forced == -2) forced = ab_int("MMSEG_WGRAD_TR_S", -1);
    long maxS = M / 128;
    if (maxS < 1) maxS = 1;
    if (maxS > 4096) maxS = 4096;
    double best = 1e30; long bestS = 1, bestChunk = (M + 31) / 32 * 32;
    for (long S = 1; S <= maxS; ++S) {
        long chunk = ((M + S - 1) / S + 31) / 32 * 32;
        const long Se = (M + chunk - 1) / chunk;
        if (Se != S && forced <= 0) continue;                          // same plan as a smaller S: already priced
        const long blocks = tiles * Se, slots = 256L * bpc;
        const long full = blocks / slots, rem = blocks % slots;
        const double blk_flop = (double)chunk * bkt * bnt * 2.0;
        double t = full * (blk_flop * bpc / (cu_rate * eff[bpc]) + 3e-6);
        if (rem) { const int r = (int)((rem + 255) / 256); t += blk_flop * r / (cu_rate * eff[r]) + 3e-6; }
        t += (double)Se * K * Cout * 8.0 / 4.5e12 + 2e-6;              // slab write + re-read, reduce launch
        if (forced > 0 ? S == forced : t < best) { best = t; bestS = Se; bestChunk = chunk; if (forced > 0) break; }
    }
    if (chunk_out) *chunk_out = (int)bestChunk;
    static int dbg = -1;
    if (dbg < 0) dbg = ab_int("MMSEG_WGRAD_DEBUG", 0);
    if (dbg && chunk_out)
        fprintf(stderr, "[wgrad_tr] M=%ld K=%ld N=%d tiles=%ld bpc=%d -> S=%ld chunk=%ld blocks=%ld model=%.1f us\n", M, K, Cout, tiles, bpc,
                bestS, bestChunk, tiles * bestS, best * 1e6);
    return (int)bestS;
}
static long wgrad_ws_floats(long S, long KN) { return (S + (S > 64 ? (S + 31) / 32 : 0)) * KN; }

long mmseg_conv2d_wgrad_workspace(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
    const long M = (long)B * Ho * Wo, K = (long)KH * KW * Cin;
    const long S = wgrad_splits(M, K, Cout);
    long need = wgrad_ws_floats(S, K * Cout);   // slabs (+ first-level partial sums); one slab even for S = 1
                                                // (accumulating launches stage their single slab)
    if (wgrad_tr_enabled() && Wo >= 4) {        // the transposed-staging kernel may take this geometry: its own split
        const long need_tr = wgrad_ws_floats(wgrad_tr_splits(M, K, Cout, nullptr), K * Cout);
        if (need_tr > need) need = need_tr;
    }
    if (KH == 3 && KW == 3 && Wo % 32 == 0 && Cin % 32 == 0 && Cout % 32 == 0) {
        // the patch-resident fp32 kernel (wgrad32h.hpp) may take this geometry: at most 512 slabs
        const long ntiles = (long)B * (Ho / 2) * (Wo / 32);
        long Sh = 512;
        if (Sh > ntiles / 8) Sh = ntiles / 8;
        if (Sh < 1) Sh = 1;
        const long need_h = wgrad_ws_floats(Sh, K * Cout);
        if (need_h > need) need = need_h;
    }
    if (KH == 5 && KW == 5 && Cin == 16 && Cout == 20) {
        // the localisation network's first layer (s2conv.hpp: locnet5_wgrad_kernel): one slab per block, at most 512 blocks
        const long need_l = wgrad_ws_floats(512, K * Cout);
        if (need_l > need) need = need_l;
    }
    return need;
}

static int conv2d_wgrad_impl(const float* x1, const float* x2, const float* dy, float* dw, float* ws, long ws_floats,
                             int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                             int pad_h, int pad_w, int ups, int accumulate, void* stream, int io);
// 1 when mmseg_conv2d_wgrad_t can read 16-bit operands for this geometry (the transposed-staging kernel takes it); otherwise the
// caller converts the operands to fp32 first
int mmseg_conv2d_wgrad_t_supported(int Ho, int Wo, int stride, int C1, int C2, int Cout) {
    return (wgrad_tr_enabled() && stride == 1 && Wo % 4 == 0 && Ho > 32 / Wo && C1 % 8 == 0 && C2 % 8 == 0 && Cout % 8 == 0 &&
            !(C1 == 8 && C2 == 0 && Cout == 8)) ? 1 : 0;
}
int mmseg_conv2d_wgrad(const float* x1, const float* x2, const float* dy, float* dw, float* ws, long ws_floats,
                       int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad_h, int pad_w, int ups, int accumulate, void* stream) {
    return conv2d_wgrad_impl(x1, x2, dy, dw, ws, ws_floats, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad_h, pad_w, ups, accumulate,
                             stream, 0);
}
// The same with 16-bit operands in HBM (reduced-precision modes; transposed-staging kernel only, i.e. stride 1 and Wo % 4 == 0): io bit 0 =
// x1 (and x2) are 16-bit, bit 2 = dy is 16-bit.  dw and the slabs stay fp32.
int mmseg_conv2d_wgrad_t(const void* x1, const void* x2, const void* dy, float* dw, float* ws, long ws_floats,
                         int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                         int pad_h, int pad_w, int ups, int accumulate, int io, void* stream) {
    return conv2d_wgrad_impl((const float*)x1, (const float*)x2, (const float*)dy, dw, ws, ws_floats, B, H, W, C1, C2, Ho, Wo, Cout, KH, KW,
                             stride, pad_h, pad_w, ups, accumulate, stream, io & 5);
}
static int conv2d_wgrad_impl(const float* x1, const float* x2, const float* dy, float* dw, float* ws, long ws_floats,
                             int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                             int pad_h, int pad_w, int ups, int accumulate, void* stream, int io) {
    WgradParams q;
    ConvParams& p = q.c;
    p.x1 = x1; p.x2 = x2; p.w = nullptr; p.wt = nullptr; p.bias = nullptr; p.oscale = nullptr; p.y = nullptr; p.y2 = nullptr;
    p.B = B; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2;
    p.H1 = ups ? H / 2 : H; p.W1 = ups ? W / 2 : W;
    p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride;
    p.pad_h = pad_h; p.pad_w = pad_w; p.ups = ups; p.transposed = 0; p.act = 0; p.alpha = 0.f;
    p.M = B * Ho * Wo; p.K = KH * KW * (C1 + C2); p.nsplit1 = 0; p.io = io;
    p.oH = Ho; p.oW = Wo; p.osh = 1; p.osw = 1; p.ooh = 0; p.oow = 0;
    if (p.M <= 0 || p.K <= 0 || Cout <= 0) return (int)hipErrorInvalidValue;
    if (io != 0 && g_conv_bf16 == 0) return (int)hipErrorInvalidValue;
    const long need = mmseg_conv2d_wgrad_workspace(B, Ho, Wo, C1 + C2, Cout, KH, KW);
    if (need > ws_floats) return (int)hipErrorInvalidValue;
    const long KN = (long)p.K * Cout;
    const bool vec = (C1 % 4 == 0) && (C2 % 4 == 0) && aligned16(x1) && (C2 == 0 || aligned16(x2));
    const long lim = (1L << 31) - 64;
    const int esx = (io & 1) ? 2 : 4, esd = (io & 4) ? 2 : 4;      // bytes per stored element: 32-bit buffer offsets
    const bool fast = vec && (Cout % 4 == 0) && aligned16(dy) && (long)B * p.H1 * p.W1 * C1 * esx < lim &&
                      (long)B * H * W * C2 * esx < lim && (long)p.M * Cout * esd < lim;
    // transposed-staging kernel: 4 consecutive pixels of a thread's block lie in one output row
    const bool tr = fast && wgrad_tr_enabled() && stride == 1 && Wo % 4 == 0 && Ho > 32 / Wo && !(C1 == 8 && C2 == 0 && Cout == 8);
    // ... or the any-width instance: each pixel with its own coordinates (the discriminators' 4x4 layers, stride 1 or 2, odd widths)
    static const bool anyw_on = ab_int("MMSEG_WGRAD_TR_ANYW", 1) != 0;
    const bool tr_any = anyw_on && !tr && fast && wgrad_tr_enabled() && io == 0 && (stride == 1 || stride == 2) && Wo >= 4 && Ho > 32 / Wo &&
                        C2 == 0 && !ups && Cout > 32 && wgrad_tr_bkt(p.K, Cout) == 128;
    int chunk;
    int S;
    if (tr || tr_any) S = wgrad_tr_splits(p.M, p.K, Cout, &chunk);
    else {
        S = wgrad_splits(p.M, p.K, Cout);
        chunk = (p.M + S - 1) / S;
        chunk = (chunk + 31) / 32 * 32;
    }
    if (io != 0 && !tr) return (int)hipErrorInvalidValue;          // only the transposed-staging kernel reads 16-bit operands
    if ((io & 1) && ((C1 & 7) || (C2 & 7))) return (int)hipErrorInvalidValue;      // a lane's 16-byte load = 8 channels
    if ((io & 4) && (Cout & 7)) return (int)hipErrorInvalidValue;
    if (wgrad_ws_floats(S, KN) > ws_floats) return (int)hipErrorInvalidValue;
    float* tmp = (S > 64) ? ws + (size_t)S * KN : nullptr;
    const bool direct = S == 1 && !accumulate;      // a single slab that overwrites dW needs no staging
    q.dy = dy; q.chunk = chunk; q.ws = direct ? dw : ws;
    hipStream_t st = (hipStream_t)stream;
    if (fast && aligned16(x1) && (C2 == 0 || aligned16(x2))) {
        // fp32, 3x3 'same', channel counts multiples of 32, rows of 32 pixels: the patch-resident kernel (wgrad32h.hpp)
        int Sh = 0, tpb = 0;
        const int nci = wgrad32h_plan(p, &Sh, &tpb);
        if (nci && wgrad_ws_floats(Sh, KN) <= ws_floats) {
            const bool directh = Sh == 1 && !accumulate;
            q.ws = directh ? dw : ws;
            const int Cin = C1 + C2, nco = 4 / nci;
            const unsigned grid = (unsigned)((Cin / (32 * nci)) * ((Cout + 32 * nco - 1) / (32 * nco)) * Sh);
            if (nci == 1) hipLaunchKernelGGL((wgrad32h_kernel<1, 4>), dim3(grid), dim3(768), 0, st, q, tpb);
            else if (nci == 2) hipLaunchKernelGGL((wgrad32h_kernel<2, 2>), dim3(grid), dim3(768), 0, st, q, tpb);
            else hipLaunchKernelGGL((wgrad32h_kernel<4, 1>), dim3(grid), dim3(768), 0, st, q, tpb);
            MMSEG_SET_LAST(18, nci, 4 / nci);
            const int rc = MMSEG_CHECK_LAUNCH();
            if (rc != 0 || directh) return rc;
            launch_slab_reduce(ws, (Sh > 64) ? ws + (size_t)Sh * KN : nullptr, dw, KN, Sh, accumulate, st);
            return MMSEG_CHECK_LAUNCH();
        }
        q.ws = direct ? dw : ws;
        // ... and its 16-bit form (operands through ds_read_b64_tr_b16)
        if (wgrad16h_plan(p, &Sh, &tpb) && wgrad_ws_floats(Sh, KN) <= ws_floats) {
            const bool directh = Sh == 1 && !accumulate;
            q.ws = directh ? dw : ws;
            const unsigned grid = (unsigned)(((C1 + C2) / 64) * ((Cout + 63) / 64) * Sh);
            if (g_conv_bf16 == 1) hipLaunchKernelGGL((wgrad16h_kernel<1>), dim3(grid), dim3(768), 0, st, q, tpb);
            else hipLaunchKernelGGL((wgrad16h_kernel<2>), dim3(grid), dim3(768), 0, st, q, tpb);
            MMSEG_SET_LAST(19, 64, 64);
            const int rc = MMSEG_CHECK_LAUNCH();
            if (rc != 0 || directh) return rc;
            launch_slab_reduce(ws, (Sh > 64) ? ws + (size_t)Sh * KN : nullptr, dw, KN, Sh, accumulate, st);
            return MMSEG_CHECK_LAUNCH();
        }
        q.ws = direct ? dw : ws;
    }
    if (C1 == 8 && C2 == 0 && Cout == 8 && KH == 3 && KW == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && !ups && Ho == H &&
        Wo == W && H % WG8_ROWS == 0 && W % WG8_COLS == 0 && S > 1 && aligned16(x1) && aligned16(dy)) {
        const int ntiles = B * (H / WG8_ROWS) * (W / WG8_COLS);
        const int nblk = ntiles < S ? ntiles : S;
        static const int c8_mfma = ab_int("MMSEG_WGRAD_C8_MFMA", 1);       // 0 (measurement builds): the VALU kernel
        if (c8_mfma && (long)B * H * W * 32 < (1L << 31) - 64) hipLaunchKernelGGL(conv_wgrad_c8m_kernel, dim3(nblk), dim3(256), 0, st, x1, dy, ws, B, H, W, ntiles);
        else hipLaunchKernelGGL(conv_wgrad_c8_kernel, dim3(nblk), dim3(256), 0, st, x1, dy, ws, B, H, W, ntiles);
        MMSEG_SET_LAST(9, 8, 8);
        launch_slab_reduce(ws, tmp, dw, KN, nblk, accumulate, st);
        return MMSEG_CHECK_LAUNCH();
    }
    {
        // HBM-bound small-channel layers (smallconv.hpp): per-block slabs, then the deterministic slab reduction
        static const int sc_mask = ab_int("MMSEG_SMALLCONV", 15);   // 4 reduce wgrad, 8 expand wgrad
        const bool base = io == 0 && C2 == 0 && !ups && aligned16(x1) && aligned16(dy) && KH == KW;
        const bool one = KH == 1 && stride == 1 && pad_h == 0 && pad_w == 0;
        long cap = wgrad_splits(p.M, p.K, Cout);                 // slabs the caller's workspace holds
        if (cap > 1024) cap = 1024;
        // the kernels below write nblk <= cap slabs: check the workspace BEFORE anything is queued (advisor, round 3)
        if (wgrad_ws_floats(cap, KN) > ws_floats) {
            while (cap > 1 && wgrad_ws_floats(cap, KN) > ws_floats) --cap;
            if (wgrad_ws_floats(cap, KN) > ws_floats) return (int)hipErrorInvalidValue;
        }
        int launched = 0, nblk = 0;
#define PWW(LANES, COUT, VPL)                                                                                                         \
        do {                                                                                                                           \
            long nb = (p.M + 256 * (64 / LANES) - 1) / (256 * (64 / LANES));                                                           \
            nblk = (int)(nb < cap ? nb : cap);                                                                                         \
            hipLaunchKernelGGL((pw_reduce_wgrad_kernel<LANES, COUT, VPL>), dim3(nblk), dim3(256),                                      \
                               (size_t)(256 / LANES) * (4 * LANES * VPL) * COUT * sizeof(float), st, x1, dy, ws, (long)p.M,            \
                               (COUT % 4 == 0) ? g_conv_bf16 : 0);                                                                      \
            MMSEG_SET_LAST(12, LANES, COUT);                                                                                           \
            launched = 1;                                                                                                              \
        } while (0)
        if (!(sc_mask & 4)) {}
        else if (base && one && C1 == 64 && Cout == 5) PWW(16, 5, 1);
        else if (base && one && C1 == 64 && Cout == 8) PWW(8, 8, 2);
        else if (base && one && C1 == 16 && Cout == 1) PWW(4, 1, 1);
        else if (base && one && C1 == 8 && Cout == 1) PWW(2, 1, 1);
#undef PWW
        const int L = Cout / 4;
        if (!launched && (sc_mask & 8) && base && C1 == 1 && (KH == 3 || KH == 4) && Cout % 4 == 0 && (L == 2 || L == 4 || L == 16)) {
            long nb = ((long)p.M * L + 4095) / 4096;
            nblk = (int)(nb < cap ? nb : cap);
            const size_t shm = (size_t)4 * p.K * Cout * sizeof(float);
#define SMW(KS, LL) hipLaunchKernelGGL((smallk_wgrad_kernel<KS, 1, LL>), dim3(nblk), dim3(256), shm, st, p, dy, ws)
            if (KH == 3) { if (L == 16) SMW(3, 16); else if (L == 4) SMW(3, 4); else SMW(3, 2); }
            else { if (L == 16) SMW(4, 16); else if (L == 4) SMW(4, 4); else SMW(4, 2); }
#undef SMW
            MMSEG_SET_LAST(13, KH * 20 + L, 1);
            launched = 1;
        }
        static const int s2_on = ab_int("MMSEG_S2CONV", 1);
        if (!launched && s2_on && s2k3c9_geometry(p) && aligned16(dy) && (reinterpret_cast<uintptr_t>(x1) & 3) == 0) {
            // the modality encoder's first layer (s2conv.hpp): one slab per block of four waves, each walking `steps` groups of 4 pixels
            const long nsteps = ((long)p.M + 3) / 4;
            long nb = cap < 512 ? cap : 512;
            if (nb * 4 > nsteps) nb = (nsteps + 3) / 4;
            if (nb < 1) nb = 1;
            const int steps = (int)((nsteps + nb * 4 - 1) / (nb * 4));
            nblk = (int)((nsteps + (long)steps * 4 - 1) / ((long)steps * 4));
            hipLaunchKernelGGL(s2k3c9_wgrad_kernel, dim3(nblk), dim3(256), 0, st, p, dy, ws, steps);
            MMSEG_SET_LAST(22, 16, 16);
            launched = 1;
        }
        if (!launched && s2_on && locnet5_wgrad_geometry(p) && aligned16(x1) && aligned16(x2) && aligned16(dy)) {
            long nb = (long)p.B * ((p.Ho + LW5_TH - 1) / LW5_TH) * ((p.Wo + LW5_TW - 1) / LW5_TW);
            if (nb > LW5_MAX_SLABS) nb = LW5_MAX_SLABS;
            if (wgrad_ws_floats(nb, KN) <= ws_floats) {
                nblk = (int)nb;
                hipLaunchKernelGGL(locnet5_wgrad_kernel, dim3(nblk), dim3(256), 0, st, p, dy, ws);
                MMSEG_SET_LAST(24, 16, 20);
                launched = 1;
            }
        }
        if (launched) {
            float* tmp2 = (nblk > 64) ? ws + (size_t)nblk * KN : nullptr;
            if (wgrad_ws_floats(nblk, KN) > ws_floats) return (int)hipErrorInvalidValue;
            launch_slab_reduce(ws, tmp2, dw, KN, nblk, accumulate, st);
            return MMSEG_CHECK_LAUNCH();
        }
    }
    int rc;
    if (tr && Cout > 64 && wgrad_tr_bkt(p.K, Cout) == 192) rc = launch_wgrad_tr<192, 128, 2, 2>(q, S, st);
    else if (tr && Cout > 32 && wgrad_tr_bkt(p.K, Cout) == 192) rc = launch_wgrad_tr<192, 64, 2, 2>(q, S, st);
    else if (tr && Cout > 64) rc = launch_wgrad_tr<128, 128, 2, 2>(q, S, st);
    else if (tr && Cout > 32) rc = launch_wgrad_tr<128, 64, 2, 2>(q, S, st);
    else if (tr) rc = launch_wgrad_tr<128, 32, 4, 1>(q, S, st);
    else if (tr_any && Cout > 64) rc = launch_wgrad_tr_anyw<128, 128, 2, 2>(q, S, st);
    else if (tr_any) rc = launch_wgrad_tr_anyw<128, 64, 2, 2>(q, S, st);
    else if (fast && Cout > 64) rc = launch_wgrad_fast<128, 128, 2, 2>(q, S, st);
    else if (fast && Cout > 32) rc = launch_wgrad_fast<128, 64, 2, 2>(q, S, st);
    else if (fast) rc = launch_wgrad_fast<128, 32, 4, 1>(q, S, st);
    else if (Cout > 32) rc = launch_wgrad<128, 64, 2, 2>(q, S, vec, st);
    else rc = launch_wgrad<128, 32, 4, 1>(q, S, vec, st);
    if (rc != 0 || direct) return rc;
    launch_slab_reduce(ws, tmp, dw, KN, S, accumulate, st);
    return MMSEG_CHECK_LAUNCH();
}

// 1 when mmseg_conv2d_fwd takes the fast path for this geometry (then it wants `wt` from mmseg_conv2d_wprep)
int mmseg_conv2d_fast_path(int C1, int C2, int Cout, int transposed) {
    return (!transposed && C1 % 32 == 0 && C2 % 32 == 0 && Cout % 4 == 0) ? 1 : 0;
}
// mode 0: forward layout [Cout][KH*KW][Cin]; mode 1: data-gradient layout [Cin][KH*KW flipped][Cout]; mode 2: the kernel as it is
int mmseg_conv2d_wprep(const float* w, float* out, int KH, int KW, int Cin, int Cout, int mode, void* stream) {
    const long n = (long)KH * KW * Cin * Cout;
    if (mode == 0 && KH * KW <= 65535) {
        const dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, KH * KW);
        hipLaunchKernelGGL(wprep_fwd_tiled_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, out, KH * KW, Cin, Cout, g_conv_bf16);
        return MMSEG_CHECK_LAUNCH();
    }
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wprep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, out, KH * KW, Cin, Cout, mode, g_conv_bf16);
    return MMSEG_CHECK_LAUNCH();
}
// taps of parity class p along one axis: kh = p, p + s, ... < K
int mmseg_conv2d_parity_taps(int K, int stride, int p) { return p < K ? (K - p + stride - 1) / stride : 0; }
int mmseg_conv2d_wprep_parity(const float* w, float* out, int KH, int KW, int Cin, int Cout, int stride, int ph, int pw, void* stream) {
    const int TH = mmseg_conv2d_parity_taps(KH, stride, ph), TW = mmseg_conv2d_parity_taps(KW, stride, pw);
    if (TH == 0 || TW == 0) return 0;
    const long n = (long)TH * TW * Cin * Cout;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wprep_parity_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, out, KH, KW, Cin, Cout, TH, TW, stride, ph, pw, g_conv_bf16);
    return MMSEG_CHECK_LAUNCH();
}
// the images of `n` weights in one launch; table: device int64 [n + 1][8] rows {src, dst, KH*KW, Cin, Cout, mode (0 | 1), first block, 0},
// row n = {.., total blocks, ..}; blocks of a row = taps * ceil(Cin / 32) * ceil(Cout / 32).  The caller keeps src / dst alive.
int mmseg_conv2d_wprep_batch(const long long* table, int n, long total_blocks, void* stream) {
    if (n < 1 || total_blocks < 1 || total_blocks >= (1L << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(wprep_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n, g_conv_bf16);
    return MMSEG_CHECK_LAUNCH();
}
// every parity class of a stride-s kernel in one launch: out = the classes' images back to back in (ph, pw) raster order
int mmseg_conv2d_wprep_parity_all(const float* w, float* out, int KH, int KW, int Cin, int Cout, int stride, void* stream) {
    if (stride < 1 || stride > 8) return (int)hipErrorInvalidValue;
    long nmax = 0;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            const long n = (long)mmseg_conv2d_parity_taps(KH, stride, ph) * mmseg_conv2d_parity_taps(KW, stride, pw) * Cin * Cout;
            if (n > nmax) nmax = n;
        }
    if (nmax == 0) return 0;
    long blocks = (nmax + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wprep_parity_all_kernel, dim3((unsigned)blocks, stride * stride), dim3(256), 0, (hipStream_t)stream, w, out, KH, KW, Cin, Cout,
                       stride, g_conv_bf16);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_conv2d_wflip(const float* w, float* wt, int KH, int KW, int Cin, int Cout, void* stream) {
    if (KH * KW > 65535) return (int)hipErrorInvalidValue;
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, KH * KW);
    hipLaunchKernelGGL(wflip_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, KH, KW, Cin, Cout);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
