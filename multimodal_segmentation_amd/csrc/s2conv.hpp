// First layer of the modality encoder: Conv2D(16, 3, strides=2, padding='valid') over Concatenate([anatomy (8 channels), image (1 channel)])
// (model_components/modality_encoder.py:34-38 of the reference: `build_simple_encoder`), its data gradient and its weight gradient.
// Included by conv.hip; fp32 tensors, fp32 arithmetic in every precision mode (the generic kernels these replace never multiplied this
// layer in 16 bits either: 9 input channels are not a multiple of 4).
//
// K = 81, N = 16 and 9 output channels of the data gradient: on the 128 x 32 implicit-GEMM tiles the layer ran at 3 - 9 TFLOP/s -- 110 us
// forward, 130 us weight gradient and 315 us data gradient per launch at 16 x 320 x 320 (30 launches each per MMSDNet iteration: 6.5 % of
// the step) for 85 MB of tensors.  It is a bandwidth layer; here it runs on v_mfma_f32_16x16x4_f32 with operands read as they lie:
//   forward        rows = 16 output channels (A = the kernel, 21 k-steps in registers), columns = 16 output pixels, a lane's B operand is
//                  2 channels of one tap of its pixel (8-byte load) or the image value of one tap; a lane ends with 4 consecutive
//                  channels of one pixel: 16-byte stores, 1 KB contiguous per wave
//   data gradient  one launch row (blockIdx.y) per parity class of the input pixels -- each an exact stride-1 convolution of dy with 1, 2, 2
//                  or 4 of the nine taps; rows = 9 input channels (of 16), columns = 16 input pixels of the class, a lane's B operand is
//                  4 channels of dy of one tap (16-byte load)
//   weight gradient rows = the 81 (tap, channel) pairs in 6 tiles of 16, columns = 16 output channels, K = 4 pixels per MFMA; the four
//                  waves of a block add their tiles in LDS in a fixed order, one slab per block, then the library's slab reduction.
// Sums are fp32 in a fixed order: deterministic like the rest of the library.

typedef float s2_f32x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ f32x4 s2_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
static __device__ __forceinline__ float s2_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
static __device__ __forceinline__ s2_f32x2 s2_load2(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(s2_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}

// y[b, ho, wo, n] = act(bias[n] + sum_{kh, kw} ( sum_{c < 8} x1[b, 2 ho + kh, 2 wo + kw, c] w[kh, kw, c, n] + x2[b, 2 ho + kh, 2 wo + kw] w[kh, kw, 8, n] ))
__global__ __launch_bounds__(256) void s2k3c9_fwd_kernel(ConvParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int nw = (int)gridDim.x * 4, w0 = (int)blockIdx.x * 4 + wid;
    const int ntiles = (p.M + 15) / 16;
    // A fragments: row j = output channel; k-step 2 tap + e holds channel 2 kq + e of the tap, k-step 18 + t the image channel of tap 4 t + kq
    float aw[21];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 2; ++e) aw[2 * tap + e] = p.w[(tap * 9 + 2 * kq + e) * 16 + j];
    int off2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = 4 * t + kq;
        aw[18 + t] = tap < 9 ? p.w[(tap * 9 + 8) * 16 + j] : 0.f;
        off2[t] = tap < 9 ? (tap / 3) * p.W + (tap % 3) : -1;
    }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + 4 * kq);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * 8 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, p.B * p.H * p.W * 4, 0x00020000);
    for (int tile = w0; tile < ntiles; tile += nw) {
        const int m = tile * 16 + j;
        const bool live = m < p.M;
        const int wo = m % p.Wo, t_ = m / p.Wo, ho = t_ % p.Ho, b = t_ / p.Ho;
        const int pix0 = (b * p.H + 2 * ho) * p.W + 2 * wo;
        s2_f32x2 xv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) xv[tap] = s2_load2(r1, live ? ((pix0 + (tap / 3) * p.W + (tap % 3)) * 8 + 2 * kq) * 4 : BUF_OOB);
        float iv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) iv[t] = s2_load1(r2, (live && off2[t] >= 0) ? (pix0 + off2[t]) * 4 : BUF_OOB);
        f32x4 acc = b4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc = s2_mfma(aw[2 * tap], xv[tap][0], acc);
            acc = s2_mfma(aw[2 * tap + 1], xv[tap][1], acc);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) acc = s2_mfma(aw[18 + t], iv[t], acc);
        // lane (pixel j, row group kq): output channels 4 kq .. 4 kq + 3
        if (live) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[r], p.act, p.alpha);
            *reinterpret_cast<f32x4*>(p.y + (size_t)m * 16 + 4 * kq) = v;
        }
    }
}

// The launch is the library's fractionally strided data-gradient launch: p.x1 = dy [B, p.H, p.W, 16] (the forward OUTPUT plane), p.w the
// flipped kernel wf[2 - kh][2 - kw][n][c] (mmseg_conv2d_wflip), p.y = d(anatomy) [B, p.Ho, p.Wo, 8], p.y2 = d(image) [B, p.Ho, p.Wo, 1] or null.
// Input pixel (h, w) = (2 i + ph, 2 jx + pw) of class (ph, pw) receives tap (kh, kw) = (ph + 2 a, pw + 2 b) from dy[i - a, jx - b].
__global__ __launch_bounds__(256) void s2k3c9_dgrad_kernel(ConvParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int ph = (int)blockIdx.y >> 1, pw = (int)blockIdx.y & 1;
    const int Hi = p.Ho, Wi = p.Wo;
    const int Hc = (Hi - ph + 1) / 2, Wc = (Wi - pw + 1) / 2;
    const int Mc = p.B * Hc * Wc, ntiles = (Mc + 15) / 16;
    const int nw = (int)gridDim.x * 4, w0 = (int)blockIdx.x * 4 + wid;
    // A fragments: row j = input channel c (9 of 16), k-step (tap, e) holds output channel n = 4 kq + e
    f32x4 aw[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int kh = ph + 2 * a, kw = pw + 2 * b;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                aw[a][b][e] = (kh < 3 && kw < 3 && j < 9) ? p.w[(((2 - kh) * 3 + (2 - kw)) * 16 + 4 * kq + e) * 9 + j] : 0.f;
        }
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * 16 * 4, 0x00020000);
    for (int tile = w0; tile < ntiles; tile += nw) {
        const int mc = tile * 16 + j;
        const bool live = mc < Mc;
        const int jx = mc % Wc, t_ = mc / Wc, i = t_ % Hc, b = t_ / Hc;
        f32x4 g[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int ho = i - a, wo = jx - bb;
                const bool ok = live && ph + 2 * a < 3 && pw + 2 * bb < 3 && (unsigned)ho < (unsigned)p.H && (unsigned)wo < (unsigned)p.W;
                g[a][bb] = buf_load4(rg, ok ? (((b * p.H + ho) * p.W + wo) * 16 + 4 * kq) * 4 : BUF_OOB);
            }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = s2_mfma(aw[a][bb][e], g[a][bb][e], acc);
        // lane (pixel j, row group kq): input channels 4 kq .. 4 kq + 3 (8 = the image)
        if (live) {
            const size_t pix = ((size_t)b * Hi + 2 * i + ph) * Wi + 2 * jx + pw;
            if (kq < 2) *reinterpret_cast<f32x4*>(p.y + pix * 8 + 4 * kq) = acc;
            else if (kq == 2 && p.y2) p.y2[pix] = acc[0];
        }
    }
}

// slab[blockIdx.x][(tap * 9 + c) * 16 + n] = sum over this block's output pixels of xcat[pixel, tap, c] * dy[pixel, n].
// MFMA rows: k' = 0 .. 71 the anatomy (tap * 8 + c), 72 .. 80 the image (tap); a wave walks `steps` groups of 4 consecutive pixels.
__global__ __launch_bounds__(256) void s2k3c9_wgrad_kernel(ConvParams p, const float* __restrict__ dy, float* __restrict__ ws, int steps) {
    __shared__ float red[4][6 * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    // per-lane element offsets (floats, relative to the pixel's first tap) of the six row tiles
    int offa[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = 16 * t + i, tap = k >> 3, c = k & 7;             // (tile 4: only i < 8 is an anatomy row)
        offa[t] = ((tap / 3) * p.W + (tap % 3)) * 8 + c;
    }
    const int tap4 = i - 8;                                            // tile 4, i >= 8: image taps 0 .. 7; tile 5, i = 0: tap 8
    const int offi4 = (tap4 / 3) * p.W + (tap4 % 3), offi5 = 2 * p.W + 2;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * 8 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, p.B * p.H * p.W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, p.M * 16 * 4, 0x00020000);
    f32x4 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = (p.M + 3) / 4;
    const int s0 = ((int)blockIdx.x * 4 + wid) * steps, s1 = min(nsteps, s0 + steps);
    for (int s = s0; s < s1; ++s) {
        const int m = 4 * s + kq;
        const bool live = m < p.M;
        const int wo = m % p.Wo, t_ = m / p.Wo, ho = t_ % p.Ho, b = t_ / p.Ho;
        const int pix0 = (b * p.H + 2 * ho) * p.W + 2 * wo;
        float a[6];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = s2_load1(r1, live ? (pix0 * 8 + offa[t]) * 4 : BUF_OOB);
        a[4] = s2_load1(r1, (live && i < 8) ? (pix0 * 8 + offa[4]) * 4 : BUF_OOB) + s2_load1(r2, (live && i >= 8) ? (pix0 + offi4) * 4 : BUF_OOB);
        a[5] = s2_load1(r2, (live && i == 0) ? (pix0 + offi5) * 4 : BUF_OOB);
        const float d = s2_load1(rd, live ? (m * 16 + i) * 4 : BUF_OOB);
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = s2_mfma(a[t], d, acc[t]);
    }
    // the block's four partial tiles meet in LDS and are added in wave order
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][(t * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    float* slab = ws + (size_t)blockIdx.x * (81 * 16);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int idx = tid + 256 * q;
        const int l = idx & 63, r = (idx >> 6) & 3, t = idx >> 8;
        const float v = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
        const int kp = 16 * t + 4 * (l >> 4) + r, n = l & 15;
        if (kp < 81) {
            const int k = kp < 72 ? (kp >> 3) * 9 + (kp & 7) : (kp - 72) * 9 + 8;
            slab[k * 16 + n] = v;
        }
    }
}

// =====================================================================================================================================
// First layer of the localisation network of the anatomy fuser: Conv2D(20, 5, padding='valid') + LeakyReLU(0.3) over
// Concatenate([anatomy 1 (8 channels), anatomy 2 (8 channels)]) (layers/stn_spline.py:102-107 of the reference: `build_locnet`), forward.
// 16 input channels and 20 outputs fit no tile of the implicit-GEMM kernels (0.21 of the fp32 peak; 69 TFLOP/s with 16-bit operands); the
// layer reads 16 x 25 values per output pixel that all sit in L1 / L2.  16-bit modes (this kernel; the operands are rounded exactly where
// the generic kernel's 16-bit instance rounded them): v_mfma_f32_16x16x32_*, rows = output channels (2 tiles of 16: 20 used), columns = 16
// output pixels, a k-step = 4 taps x the 8 channels of ONE tensor.  A block owns 8 x 64 output pixels: its 12 x 68-pixel input patch is
// converted once into LDS (a pixel = 16 channels = 32 bytes), a lane's B operand -- 8 channels of one tensor of one tap of its pixel -- is
// ONE ds_read_b128; the kernel (14 k-steps x 2 row tiles of 4 registers) stays in registers; a wave multiplies two pixel tiles per pass.
// A lane ends with 4 consecutive channels of a pixel: 16-byte stores.  (First version: the operands straight from global memory, 28 KB of
// 16-byte loads per 16 pixels -- as slow as the generic kernel: four waves' rows did not fit the 32 KB L1.)
constexpr int LN5_TH = 8, LN5_TW = 64, LN5_PW = LN5_TW + 4, LN5_PH = LN5_TH + 4;
template <int PREC>
__global__ __launch_bounds__(256, 2) void locnet5_fwd_kernel(ConvParams p) {
    typedef typename LowPrec<PREC>::V8 LV8;
    typedef typename LowPrec<PREC>::T LT;
    constexpr int NG = 7;                           // tap groups of 4 (25 taps)
    // the block's input patch, 16-bit, a pixel = [8 channels of x1 | 8 channels of x2] = 32 bytes: a lane's B operand is ONE ds_read_b128
    __shared__ __attribute__((aligned(16))) LT patch[LN5_PH * LN5_PW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int tw = (p.Wo + LN5_TW - 1) / LN5_TW, th = (p.Ho + LN5_TH - 1) / LN5_TH;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, p.B * p.H * p.W * 32, 0x00020000);
    constexpr int NPX = LN5_PH * LN5_PW, NIT = (NPX + 255) / 256;
    // A fragments: row j (+ 16 rt) = output channel, k = 8 kg + c = channel c of tap 4 g + kg of tensor t; fetched ONCE per block (a block
    // walks tiles bid, bid + gridDim.x, ...: 224 scalar loads per lane would otherwise cost more than a tile's arithmetic)
    LV8 aw[NG][2][2];
    int tapoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int tap = 4 * g + kg;
        tapoff[g] = tap < 25 ? ((tap / 5) * LN5_PW + (tap % 5)) * 16 : 0;         // (elements; taps 25 .. 27 multiply zero weights)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int n = 16 * rt + j;
                LV8 v;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = (tap < 25 && n < 20) ? (LT)p.w[(tap * 16 + 8 * t + c) * 20 + n] : (LT)0.f;
                aw[g][t][rt] = v;
            }
    }
    f32x4 b4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (p.bias) {
        b4[0] = *reinterpret_cast<const f32x4*>(p.bias + 4 * kg);
        if (kg == 0) b4[1] = *reinterpret_cast<const f32x4*>(p.bias + 16);
    }
    for (int bid = (int)blockIdx.x; bid < p.B * tw * th; bid += (int)gridDim.x) {
    const int b = bid / (tw * th), tr = bid - b * (tw * th), ty = tr / tw, tx = tr - ty * tw;
    const int r0 = ty * LN5_TH, c0 = tx * LN5_TW;
    // stage: all loads first (branch-free, out-of-image pixels read as zeros), then the converted rows
    f32x4 xv[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int pr = i / LN5_PW, pc = i - pr * LN5_PW;
        const int hi = r0 + pr, wi = c0 + pc;
        const int off = (i < NPX && hi < p.H && wi < p.W) ? ((b * p.H + hi) * p.W + wi) * 32 : BUF_OOB;
        xv[it][0] = buf_load4(r1, off); xv[it][1] = buf_load4(r1, off == BUF_OOB ? BUF_OOB : off + 16);
        xv[it][2] = buf_load4(r2, off); xv[it][3] = buf_load4(r2, off == BUF_OOB ? BUF_OOB : off + 16);
    }
    __syncthreads();                                // the previous tile's patch is consumed (this tile's loads are already in flight)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        if (i < NPX) {
            const f32x4 a0 = xv[it][0], a1 = xv[it][1], c0v = xv[it][2], c1v = xv[it][3];
            *reinterpret_cast<LV8*>(patch + i * 16) = LV8{(LT)a0[0], (LT)a0[1], (LT)a0[2], (LT)a0[3], (LT)a1[0], (LT)a1[1], (LT)a1[2], (LT)a1[3]};
            *reinterpret_cast<LV8*>(patch + i * 16 + 8) = LV8{(LT)c0v[0], (LT)c0v[1], (LT)c0v[2], (LT)c0v[3], (LT)c1v[0], (LT)c1v[1], (LT)c1v[2], (LT)c1v[3]};
        }
    }
    __syncthreads();
    // wave w: output rows 2 w, 2 w + 1 of the tile, four 16-pixel column tiles each, two tiles per pass
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int row = 2 * wid + (pass >> 1), colb = (pass & 1) * 32;
        f32x4 acc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { acc[0][u] = b4[0]; acc[1][u] = b4[1]; }
        const LT* base = patch + (row * LN5_PW + colb + j) * 16;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                LV8 bx[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) bx[u] = *reinterpret_cast<const LV8*>(base + tapoff[g] + (16 * u) * 16 + 8 * t);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if constexpr (PREC == 1) acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[g][t][rt], bx[u], acc[rt][u], 0, 0, 0);
                        else acc[rt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[g][t][rt], bx[u], acc[rt][u], 0, 0, 0);
                    }
            }
        // lane (pixel j, row group kg): channels 4 kg .. 4 kg + 3 of row tile 0, 16 .. 19 of row tile 1 (kg = 0)
        const int ho = r0 + row;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int wo = c0 + colb + 16 * u + j;
            if (ho < p.Ho && wo < p.Wo) {
                const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[0][u][r], p.act, p.alpha);
                *reinterpret_cast<f32x4*>(p.y + m * 20 + 4 * kg) = v;
                if (kg == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[1][u][r], p.act, p.alpha);
                    *reinterpret_cast<f32x4*>(p.y + m * 20 + 16) = v;
                }
            }
        }
    }
    }
}

// The 5 x 5 layers of the localisation network in fp32 (v_mfma_f32_16x16x4_f32, exact fp32 products like every fp32 kernel of the library):
// <8, 8, 20, .> the first layer, <20, 0, 20, .> the second and third (stn_spline.py:108-113) and their data gradients (padding 4, flipped kernel),
// <20, 0, 16> the first layer's data gradient (its 16 outputs go to the two anatomies' gradients, 8 + 8).  25 taps x CT / 4 k-steps, lane
// (pixel, kq) supplies channel 4 q + kq.  Two row tiles (20 outputs): wave w owns row tile w & 1 and half of the tile's 32 pixel groups; one
// row tile (16 outputs): a quarter each.  The wave's weight values (100 / 125) stay in registers; the patch sits in LDS in fp32 with an 80-byte
// pixel pitch (the 16 pixels of a read then cover the 64 banks once; 64 bytes would put pixels j and j + 4 on one bank).  MFMA-bound: 32 cycles
// per k-step, 16 pixels and row tile (the second row tile multiplies 12 rows of zeros).
// PREC != 0 (16-bit modes): the operands are rounded to the mode's 16-bit type exactly where the generic kernel's 16-bit instance -- which these
// launches used to run on -- rounded them (activations and weights of forward and stride-1 data-gradient launches); products and sums stay fp32.
template <int PREC> __device__ __forceinline__ float ln5_round(float v) {
    if constexpr (PREC == 0) return v;
    else return (float)(typename LowPrec<PREC>::T)v;
}
template <int CA, int CB, int NOUT, int PREC>
__global__ __launch_bounds__(256, 2) void locnet5_f32_kernel(ConvParams p) {
    constexpr int PP = 20, CT = CA + CB, NQ = CT / 4, NRT = (NOUT + 15) / 16;
    static_assert(CT <= PP && CT % 4 == 0 && CA % 4 == 0 && NOUT % 4 == 0, "channel quads");
    __shared__ __attribute__((aligned(16))) float patch[LN5_PH * LN5_PW * PP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int rt = NRT == 2 ? (wid & 1) : 0, part = NRT == 2 ? (wid >> 1) : wid;
    constexpr int NGRP = NRT == 2 ? 8 : 4;          // passes (of two 16-pixel groups) per wave and tile
    const int tw = (p.Wo + LN5_TW - 1) / LN5_TW, th = (p.Ho + LN5_TH - 1) / LN5_TH;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * CA * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(CB ? p.x2 : p.x1), 0, CB ? p.B * p.H * p.W * CB * 4 : 0, 0x00020000);
    constexpr int NPX = LN5_PH * LN5_PW, NIT = (NPX + 255) / 256;
    // A operand of this wave's row tile: k-step (tap, q) -> w[tap][4 q + kq][16 rt + j]
    float aw[25][NQ];
    const int n = 16 * rt + j;
#pragma unroll
    for (int tap = 0; tap < 25; ++tap)
#pragma unroll
        for (int q = 0; q < NQ; ++q) aw[tap][q] = n < NOUT ? ln5_round<PREC>(p.w[(tap * CT + 4 * q + kq) * NOUT + n]) : 0.f;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && 16 * rt + 4 * kq < NOUT) b4 = *reinterpret_cast<const f32x4*>(p.bias + 16 * rt + 4 * kq);
    for (int bid = (int)blockIdx.x; bid < p.B * tw * th; bid += (int)gridDim.x) {
        const int b = bid / (tw * th), tr = bid - b * (tw * th), ty = tr / tw, tx = tr - ty * tw;
        const int r0 = ty * LN5_TH, c0 = tx * LN5_TW;
        f32x4 xv[NIT][NQ];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int pr = i / LN5_PW, pc = i - pr * LN5_PW;
            const int hi = r0 + pr - p.pad_h, wi = c0 + pc - p.pad_w;
            const bool ok = i < NPX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int pix = (b * p.H + hi) * p.W + wi;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                xv[it][q] = 4 * q < CA ? buf_load4(r1, ok ? (pix * CA + 4 * q) * 4 : BUF_OOB) : buf_load4(r2, ok ? (pix * CB + 4 * q - CA) * 4 : BUF_OOB);
        }
        __syncthreads();                            // the previous tile's patch is consumed (this tile's loads are already in flight)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            if (i < NPX) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 v = xv[it][q];
                    *reinterpret_cast<f32x4*>(patch + i * PP + 4 * q) = f32x4{ln5_round<PREC>(v[0]), ln5_round<PREC>(v[1]), ln5_round<PREC>(v[2]), ln5_round<PREC>(v[3])};
                }
            }
        }
        __syncthreads();
        // two 16-pixel groups per pass (two independent accumulator chains)
#pragma unroll 1
        for (int gq = 0; gq < NGRP; ++gq) {
            const int row = (LN5_TH / (NRT == 2 ? 2 : 4)) * part + (gq >> 1), col = (gq & 1) * 32;
            const float* base = patch + (row * LN5_PW + col + j) * PP + kq;
            f32x4 acc[2] = {b4, b4};
#pragma unroll
            for (int tap = 0; tap < 25; ++tap)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[u] = s2_mfma(aw[tap][q], base[((tap / 5) * LN5_PW + (tap % 5) + 16 * u) * PP + 4 * q], acc[u]);
            // lane (pixel j, row group kq): channels 16 rt + 4 kq .. + 3
            const int ho = r0 + row, ch = 16 * rt + 4 * kq;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int wo = c0 + col + 16 * u + j;
                if (ho < p.Ho && wo < p.Wo && ch < NOUT) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_apply(acc[u][r], p.act, p.alpha);
                    const size_t m = ((size_t)b * p.Ho + ho) * p.Wo + wo;
                    if (p.y2 == nullptr) *reinterpret_cast<f32x4*>(p.y + m * NOUT + ch) = v;
                    else if (ch < p.nsplit1) *reinterpret_cast<f32x4*>(p.y + m * p.nsplit1 + ch) = v;
                    else *reinterpret_cast<f32x4*>(p.y2 + m * (NOUT - p.nsplit1) + (ch - p.nsplit1)) = v;
                }
            }
        }
    }
}

// Weight gradient of the localisation network's first layer: dW[tap][c][n] = sum over output pixels of xcat[pixel + tap][c] * dy[pixel][n].
// v_mfma_f32_16x16x4_f32 with K = 4 consecutive pixels: a tap's 16 input channels are exactly one 16-row tile, the 20 output channels two
// column tiles; the 25 taps are dealt to the four waves (7 / 6 / 6 / 6: 56 accumulator registers), every wave walks all pixel groups of the
// block's tile (4 x 64 output pixels: 8 x 68-pixel patch + the dy tile in LDS, both fp32 with an 80-byte pixel pitch).  One slab per block,
// every element written by its one owner; then the library's fixed-order slab reduction.  fp32 in every mode, like the kernel it replaces.
constexpr int LW5_TH = 4, LW5_TW = 64, LW5_PW = LW5_TW + 4, LW5_PH = LW5_TH + 4;
__global__ __launch_bounds__(256, 2) void locnet5_wgrad_kernel(ConvParams p, const float* __restrict__ dy, float* __restrict__ ws) {
    constexpr int PP = 20, NPX = LW5_PH * LW5_PW, NITX = (NPX + 255) / 256, NT = 7;
    __shared__ __attribute__((aligned(16))) float patch[NPX * PP];
    __shared__ __attribute__((aligned(16))) float dyt[LW5_TH * LW5_TW * PP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int tw = (p.Wo + LW5_TW - 1) / LW5_TW, th = (p.Ho + LW5_TH - 1) / LW5_TH;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H * p.W * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, p.B * p.H * p.W * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, p.M * 80, 0x00020000);
    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // this wave's taps: wid, wid + 4, ...; operand offsets (floats) of lane (row / column j, pixel kq of the group)
    int offa[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = wid + 4 * t;
        offa[t] = tap < 25 ? ((tap / 5) * LW5_PW + (tap % 5) + kq) * PP + j : -1;
    }
    const int offb = kq * PP + j;
    for (int bid = (int)blockIdx.x; bid < p.B * tw * th; bid += (int)gridDim.x) {
        const int b = bid / (tw * th), tr = bid - b * (tw * th), ty = tr / tw, tx = tr - ty * tw;
        const int r0 = ty * LW5_TH, c0 = tx * LW5_TW;
        f32x4 xv[NITX][4], dv[5];
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int i = tid + 256 * it;
            const int pr = i / LW5_PW, pc = i - pr * LW5_PW;
            const int hi = r0 + pr, wi = c0 + pc;
            const int off = (i < NPX && hi < p.H && wi < p.W) ? ((b * p.H + hi) * p.W + wi) * 32 : BUF_OOB;
            xv[it][0] = buf_load4(r1, off); xv[it][1] = buf_load4(r1, off == BUF_OOB ? BUF_OOB : off + 16);
            xv[it][2] = buf_load4(r2, off); xv[it][3] = buf_load4(r2, off == BUF_OOB ? BUF_OOB : off + 16);
        }
        {
            const int pr = tid / LW5_TW, pc = tid - pr * LW5_TW;
            const int ho = r0 + pr, wo = c0 + pc;
            const int off = (ho < p.Ho && wo < p.Wo) ? ((b * p.Ho + ho) * p.Wo + wo) * 80 : BUF_OOB;
#pragma unroll
            for (int q = 0; q < 5; ++q) dv[q] = buf_load4(rd, off == BUF_OOB ? BUF_OOB : off + 16 * q);
        }
        __syncthreads();                            // the previous tile is consumed (this tile's loads are already in flight)
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int i = tid + 256 * it;
            if (i < NPX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(patch + i * PP + 4 * q) = xv[it][q];
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) *reinterpret_cast<f32x4*>(dyt + tid * PP + 4 * q) = dv[q];
        __syncthreads();
#pragma unroll 2
        for (int g = 0; g < LW5_TH * LW5_TW / 4; ++g) {
            const int row = g / (LW5_TW / 4), col = (g % (LW5_TW / 4)) * 4;
            const float* pa = patch + (row * LW5_PW + col) * PP;
            const float* pb = dyt + (row * LW5_TW + col) * PP + offb;
            const float d0 = pb[0];
            const float d1v = pb[16];
            const float d1 = j < 4 ? d1v : 0.f;                               // (columns 20 .. 31 of the second tile do not exist)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (wid + 4 * t < 25) {
                    const float a = pa[offa[t]];
                    acc[t][0] = s2_mfma(a, d0, acc[t][0]);
                    acc[t][1] = s2_mfma(a, d1, acc[t][1]);
                }
        }
    }
    // lane (column j, row group kq), register r: dW[tap][c = 4 kq + r][n = 16 ct + j]
    float* slab = ws + (size_t)blockIdx.x * (400 * 20);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = wid + 4 * t;
        if (tap < 25) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                slab[(tap * 16 + 4 * kq + r) * 20 + j] = acc[t][0][r];
                if (j < 4) slab[(tap * 16 + 4 * kq + r) * 20 + 16 + j] = acc[t][1][r];
            }
        }
    }
}
static bool locnet5_wgrad_geometry(const ConvParams& p) {
    return p.KH == 5 && p.KW == 5 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && !p.ups && p.io == 0 && p.C1 == 8 && p.C2 == 8 && p.Cout == 20 &&
           p.Ho == p.H - 4 && p.Wo == p.W - 4 && p.Ho >= 1 && p.Wo >= 1 && (long)p.B * p.H * p.W * 32 < (1L << 31) - 64 && (long)p.M * 80 < (1L << 31) - 64;
}
constexpr int LW5_MAX_SLABS = 512;

static bool s2k3c9_geometry(const ConvParams& p) {
    return p.KH == 3 && p.KW == 3 && p.stride == 2 && p.pad_h == 0 && p.pad_w == 0 && !p.ups && p.io == 0 && p.C1 == 8 && p.C2 == 1 && p.Cout == 16 &&
           p.Ho == (p.H - 3) / 2 + 1 && p.Wo == (p.W - 3) / 2 + 1 && p.H >= 3 && p.W >= 3 && (long)p.B * p.H * p.W * 8 * 4 < (1L << 31) - 64 &&
           (long)p.M * 16 * 4 < (1L << 31) - 64;
}
// forward / data-gradient launches of conv_dispatch: >= 0 = launched (hipError_t), -1 = not this layer
static int s2conv_dispatch(const ConvParams& p, hipStream_t st) {
    static const int on = ab_int("MMSEG_S2CONV", 1);            // measurement builds: 0 = the generic kernels
    if (!on) return -1;
    const bool plain = p.oscale == nullptr && p.osh == 1 && p.osw == 1 && p.ooh == 0 && p.oow == 0 && p.oH == p.Ho && p.oW == p.Wo;
    if (!p.transposed && plain && p.y2 == nullptr && p.w != nullptr && s2k3c9_geometry(p) && aligned16(p.y) && aligned16(p.w) &&
        (reinterpret_cast<uintptr_t>(p.x1) & 7) == 0 && (p.bias == nullptr || aligned16(p.bias))) {
        long blocks = ((p.M + 15) / 16 + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(s2k3c9_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
        MMSEG_SET_LAST(21, 16, 16);
        return MMSEG_CHECK_LAUNCH();
    }
    // the localisation network's 5 x 5 layers and their data gradients (the first layer's forward pass on 16-bit MFMA operands in the 16-bit
    // modes, everything else on the fp32 MFMA)
    if (!p.transposed && plain && p.w != nullptr && p.KH == 5 && p.KW == 5 && p.stride == 1 && p.pad_h == p.pad_w && (p.pad_h == 0 || p.pad_h == 4) &&
        !p.ups && p.io == 0 && p.Ho == p.H + 2 * p.pad_h - 4 && p.Wo == p.W + 2 * p.pad_w - 4 && p.Ho >= 1 && p.Wo >= 1 &&
        aligned16(p.x1) && (p.C2 == 0 || aligned16(p.x2)) && aligned16(p.y) && (p.y2 == nullptr || aligned16(p.y2)) && aligned16(p.w) &&
        (p.bias == nullptr || aligned16(p.bias)) && (long)p.B * p.H * p.W * 80 < (1L << 31) - 64) {
        const int shape = (p.C1 == 8 && p.C2 == 8 && p.Cout == 20 && p.y2 == nullptr) ? 1
                        : (p.C1 == 20 && p.C2 == 0 && p.Cout == 20 && p.y2 == nullptr) ? 2
                        : (p.C1 == 20 && p.C2 == 0 && p.Cout == 16 && (p.y2 == nullptr || p.nsplit1 == 8)) ? 3 : 0;
        if (shape) {
            long blocks = (long)p.B * ((p.Ho + LN5_TH - 1) / LN5_TH) * ((p.Wo + LN5_TW - 1) / LN5_TW);
            if (blocks > 512) blocks = 512;             // two blocks per CU, each walking tiles with a stride of the grid
            const dim3 grid((unsigned)blocks), blk(256);
            if (shape == 1 && p.pad_h == 0 && g_conv_bf16 == 1) hipLaunchKernelGGL((locnet5_fwd_kernel<1>), grid, blk, 0, st, p);
            else if (shape == 1 && p.pad_h == 0 && g_conv_bf16 == 2) hipLaunchKernelGGL((locnet5_fwd_kernel<2>), grid, blk, 0, st, p);
#define LN5(CA, CB, NO) do { if (g_conv_bf16 == 1) hipLaunchKernelGGL((locnet5_f32_kernel<CA, CB, NO, 1>), grid, blk, 0, st, p);         \
                             else if (g_conv_bf16 == 2) hipLaunchKernelGGL((locnet5_f32_kernel<CA, CB, NO, 2>), grid, blk, 0, st, p);    \
                             else hipLaunchKernelGGL((locnet5_f32_kernel<CA, CB, NO, 0>), grid, blk, 0, st, p); } while (0)
            else if (shape == 1) LN5(8, 8, 20);
            else if (shape == 2) LN5(20, 0, 20);
            else LN5(20, 0, 16);
#undef LN5
            MMSEG_SET_LAST(23, p.C1 + p.C2, p.Cout);
            return MMSEG_CHECK_LAUNCH();
        }
    }
    // the data gradient arrives as a fractionally strided launch over dy: 16 input channels, 9 outputs split 8 + 1, padding K - 1
    if (p.transposed && plain && p.w != nullptr && p.KH == 3 && p.KW == 3 && p.stride == 2 && p.pad_h == 2 && p.pad_w == 2 && !p.ups && p.io == 0 &&
        p.C1 == 16 && p.C2 == 0 && p.Cout == 9 && p.y2 != nullptr && p.nsplit1 == 8 && p.bias == nullptr && p.act == 0 &&
        p.H == (p.Ho - 3) / 2 + 1 && p.W == (p.Wo - 3) / 2 + 1 && aligned16(p.x1) && aligned16(p.y) &&
        (long)p.B * p.Ho * p.Wo * 8 * 4 < (1L << 31) - 64 && (long)p.B * p.H * p.W * 16 * 4 < (1L << 31) - 64) {
        const long mc = (long)p.B * ((p.Ho + 1) / 2) * ((p.Wo + 1) / 2);          // the largest parity class
        long blocks = ((mc + 15) / 16 + 3) / 4;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(s2k3c9_dgrad_kernel, dim3((unsigned)blocks, 4), dim3(256), 0, st, p);
        MMSEG_SET_LAST(21, 16, 9);
        return MMSEG_CHECK_LAUNCH();
    }
    return -1;
}
