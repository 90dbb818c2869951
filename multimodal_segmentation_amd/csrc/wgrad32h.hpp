// Patch-resident fp32 weight gradient of the 3x3 'same' stride-1 convolutions (round 4):
//     dW[kh][kw][ci][co] = sum over pixels (b, y, x) of  x[b, y + kh - 1, x + kw - 1, ci] * dy[b, y, x, co]
// (reference: the gradient Keras/TF computes for Conv2D(f, 3, padding='same') of models/unet.py:94-101, model_components/segmentor.py).
//
// conv_wgrad_tr_kernel stages both operands through registers to transpose them (the reduction index of this GEMM is the pixel) and
// splits K = 9 taps x channels into 128-row tiles that each re-gather the activations: 0.63 of the fp32 MFMA peak, 2.05 x the
// algorithmic HBM bytes (round 3).  On the fp32 MFMA no transposition is needed at all: v_mfma_f32_32x32x2_f32 takes ONE float per lane
// for A[i][k] (lane = i + 32 k) and B[k][j]; with i = input channel, j = output channel and k = pixel, a lane's A operand is
// x[pixel + k][ci0 + i] -- 32 lanes read the 32 channels of one pixel, a contiguous 128-byte row of the NHWC tensor as it lies in LDS
// (ds_read_b32, conflict-free) -- and its B operand is dy[pixel + k][co0 + j] likewise.  So:
//   * a block owns NCI x 32 input channels, NCO x 32 output channels (NCI NCO = 4) and a share of the image's pixel tiles (2 rows x 32
//     columns); per tile it brings the 4 x 34-pixel activation patch (128 bytes = 32 fp32 channels per pixel and plane) and the 64-pixel
//     dy tile into LDS by buffer_load ... lds (1 KB pieces, source-side chunk swizzle as in conv16h), double buffered: the next tile lands
//     while this one is multiplied (~ 9 us), one barrier per tile;
//   * 12 waves = (input-channel plane, output-channel plane, kernel row kh): a wave keeps the three 32 x 32 accumulators of its kernel
//     row's taps kw = 0, 1, 2 in registers for the whole launch; per pixel pair it reads one dy operand and three shifted activation
//     operands (4 ds_read_b32) for 3 MFMAs.  Every tap reads the SAME patch: activations enter the CU once per tile, not once per tap;
//   * the block writes its part of slab `s` (the dW layout) once at the end; the existing fixed-order slab reduction adds the slabs.
#pragma once
#include <type_traits>

template <int NCI, int NCO>
__device__ __forceinline__ void wgrad32h_body(const WgradParams& q, const int tpb) {
    static_assert(NCI * NCO == 4, "12 waves = 4 (channel plane pairs) x 3 (kernel rows)");
    constexpr int NW = 12;
    constexpr int XPIX = 4 * 34, XP = XPIX * 128, DT = 64 * 128;         // patch: 4 rows x 34 pixels (17 pieces of 8); dy tile: 64 pixels
    constexpr int XPC = XPIX / 8, DPC = 8;                                // pieces per plane
    constexpr int NPC = NCI * XPC + NCO * DPC;                            // pieces per stage
    constexpr int PPW = (NPC + NW - 1) / NW;                              // pieces per wave
    constexpr int SS = NCI * XP + NCO * DT;                               // bytes per stage
    static_assert(2 * SS <= 160 * 1024, "LDS");
    const ConvParams& p = q.c;

    __shared__ __attribute__((aligned(1024))) char smem[2 * SS];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int kh = wid % 3, pl = wid / 3;                                 // kernel row; plane pair
    const int cis = pl / NCO, cos = pl % NCO;                             // this wave's input / output channel plane
    const int Cin = p.C1 + p.C2;
    const int ncb = Cin / (32 * NCI), nnb = (p.Cout + 32 * NCO - 1) / (32 * NCO);
    // the blocks of one pixel share (all channel-plane pairs) are consecutive logical ids = one XCD: they walk the same dy / activation
    // tiles at the same pace and meet in that XCD's L2 (the same tiles from eight different L2s cost a fabric read each)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % ncb, nb = (bid / ncb) % nnb, s = bid / (ncb * nnb);
    const int tw = p.W / 32, th = p.H / 2, tpi = tw * th, ntiles = p.B * tpi;
    const int t0 = s * tpb, t1 = min(ntiles, t0 + tpb);

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)q.dy, 0, p.M * p.Cout * 4, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int FAR = 0x40000000;

    // pieces of this wave: id = jj * NW + wid.  id < NCI * 17: activation plane id / 17, piece id % 17; else dy plane (id - NCI * 17) / 8.
    // Per lane: the pixel of its piece and the byte offset of its source chunk inside a pixel row; everything that depends on the tile
    // (image position, border validity) is recomputed per tile -- a few VALU against ~ 6 000 MFMA cycles per wave and tile
    auto issue_tile = [&](const int t, const int stage) {
        char* base = smem + stage * SS;
        // tiles walk DOWN a 32-pixel column strip (ty fastest): consecutive tiles of a block share two of their four patch rows, re-read 9 us
        // later from L2; in raster order the rows were re-read 8 tiles later, after the XCD had streamed 12 MB through its 4 MB L2
        // (measured 1.9 x the algorithmic bytes at the fabric)
        const int b = t / tpi, tr = t - b * tpi;
        const int tx = tr / th, ty = tr - tx * th;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int id = jj * NW + wid;
            if (id >= NPC) break;
            if (id < NCI * XPC) {
                const int plane = id / XPC, pp = id - plane * XPC;
                const int pix = 8 * pp + (lane >> 3);
                const int sc16 = 16 * ((lane & 7) ^ ((pix >> 1) & 7));
                const int hy = pix / 34, hx = pix - hy * 34;
                const int y = 2 * ty - 1 + hy, x = 32 * tx - 1 + hx;
                const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const int c0 = (cb * NCI + plane) * 32;                   // first channel of the plane: in x1 or in x2 (wave-uniform)
                if (c0 < p.C1) {
                    const int pixoff = p.ups ? ((b * p.H1 + (y >> 1)) * p.W1 + (x >> 1)) : ((b * p.H + y) * p.W + x);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(base + plane * XP + pp * 1024), 16,
                                                             ok ? (pixoff * p.C1 + c0) * 4 + sc16 : FAR, 0, 0, 0);
                } else {
                    const int pixoff = (b * p.H + y) * p.W + x;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)(base + plane * XP + pp * 1024), 16,
                                                             ok ? (pixoff * p.C2 + (c0 - p.C1)) * 4 + sc16 : FAR, 0, 0, 0);
                }
            } else {
                const int d = id - NCI * XPC;
                const int plane = d / DPC, pp = d - plane * DPC;
                const int pix = 8 * pp + (lane >> 3);                     // 0 .. 63: row pix >> 5, column pix & 31 of the tile
                const int sc16 = 16 * ((lane & 7) ^ ((pix >> 1) & 7));
                const int m = (b * p.H + 2 * ty + (pix >> 5)) * p.W + 32 * tx + (pix & 31);
                const int n0 = (nb * NCO + plane) * 32;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr)(base + NCI * XP + plane * DT + pp * 1024), 16,
                                                         n0 < p.Cout ? (m * p.Cout + n0) * 4 + sc16 : FAR, 0, 0, 0);
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kw][r] = 0.f;

    // operand addresses inside a stage: element (pixel hp, channel li) of a plane sits at hp * 128 + 16 * ((li >> 2) ^ ((hp >> 1) & 7)) + 4 * (li & 3).
    // The 32 pixel pairs of a tile are walked as (row r, 16-column block blk, pair j): hp = (r + kh) * 34 + 16 blk + 2 j + kw + lh, so
    // (hp >> 1) & 7 = ((r + kh) * 17 + j + ((kw + lh) >> 1)) & 7 -- independent of blk, a compile-time rotation of j once kh is a constant
    // (kw = 0: + 0, kw = 2: + 1, kw = 1: + lh).  With the 8 swizzled chunk offsets of a lane in registers (T0; T1 = rotated by lh) an
    // operand address is ONE add (row base + T[...]) plus an immediate (blk * 2048 + j * 256): the first version recomputed every
    // address from its pixel (8 VALU per MFMA, 0.57 of the fp32 peak).
    const int cq = li >> 2, cr4 = 4 * (li & 3);
    int T0[8], T1[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { T0[v] = 16 * (cq ^ v); T1[v] = 16 * (cq ^ ((v + lh) & 7)); }
    auto mma_tile_kh = [&](const int stage, auto khc) {
        constexpr int KHc = decltype(khc)::value;
        const int xb = stage * SS + cis * XP + cr4, db = stage * SS + NCI * XP + cos * DT + cr4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int sbr = db + (r * 32 + lh) * 128;
            int sar[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) sar[kw] = xb + ((r + KHc) * 34 + kw + lh) * 128;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float bv = *reinterpret_cast<const float*>(smem + (sbr + T0[j]) + blk * 2048 + j * 256);
                    float av[3];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int rot = ((r + KHc) * 17 + j + (kw == 2 ? 1 : 0)) & 7;
                        av[kw] = *reinterpret_cast<const float*>(smem + (sar[kw] + (kw == 1 ? T1[rot] : T0[rot])) + blk * 2048 + j * 256);
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kw], bv, acc[kw], 0, 0, 0);
                    // (three waves per SIMD cover the LDS latency; without a fence hipcc hoists dozens of the 128 reads of this straight-line
                    // tile ahead and spills their destinations)
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    auto mma_tile = [&](const int stage) {
        if (kh == 0) mma_tile_kh(stage, std::integral_constant<int, 0>());
        else if (kh == 1) mma_tile_kh(stage, std::integral_constant<int, 1>());
        else mma_tile_kh(stage, std::integral_constant<int, 2>());
    };

    if (t0 < t1) {
        issue_tile(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int t = t0; t < t1; ++t) {
            const int stage = (t - t0) & 1;
            if (t + 1 < t1) issue_tile(t + 1, stage ^ 1);                 // (that stage was read one iteration ago: barrier below)
            mma_tile(stage);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    // ---- this wave's three 32 x 32 blocks of slab s: dW[kh][kw][ci][co], accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lh (ci), column li (co)
    float* slab = q.ws + (size_t)s * ((size_t)p.K * p.Cout);
    const int co = (nb * NCO + cos) * 32 + li;
    if (co < p.Cout) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (cb * NCI + cis) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[((size_t)(kh * 3 + kw) * Cin + ci) * p.Cout + co] = acc[kw][r];
            }
    }
}
template <int NCI, int NCO>
__global__ __launch_bounds__(768, 3) void wgrad32h_kernel(WgradParams q, int tpb) {
    wgrad32h_body<NCI, NCO>(q, tpb);
}

// plan: 0 = not applicable; else NCI (1, 2 or 4), with the number of pixel splits S and the tiles per block
static int wgrad32h_plan(const ConvParams& p, int* S_out, int* tpb_out) {
    static const int on = ab_int("MMSEG_WGRAD32H", 1);
    if (!on || g_conv_bf16 != 0 || p.io != 0 || g_conv16_mode == 0) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.W % 32 || p.H % 2 || p.C1 % 32 || p.C2 % 32 || p.Cout % 32) return 0;
    const int Cin = p.C1 + p.C2;
    int nci;
    if (p.Cout % 128 == 0) nci = 1;
    else if (p.Cout % 64 == 0 && p.C1 % 64 == 0 && p.C2 % 64 == 0) nci = 2;
    else if (p.C1 % 128 == 0 && p.C2 % 128 == 0) nci = 4;
    else return 0;
    const int nco = 4 / nci;
    const long nbk = (long)(Cin / (32 * nci)) * ((p.Cout + 32 * nco - 1) / (32 * nco));
    const long ntiles = (long)p.B * (p.H / 2) * (p.W / 32);
    // one block per CU (12 waves, 98 - 152 KB of LDS): ONE round of 256 blocks where the pixel tiles allow, each block at least 8 tiles long
    // (fewer, longer blocks = fewer slabs to write and reduce)
    static const int target = ab_int("MMSEG_WGRAD32H_BLOCKS", 256);      // one round of blocks (256 / 512 / 768 / 1024 measured: tools/ab_wgrad_blocks.sh)
    long S = (target + nbk - 1) / nbk;
    if (S > ntiles / 8) S = ntiles / 8;
    if (S < 1) S = 1;
    if (g_conv16_mode != 2 && nbk * S < 192) return 0;
    long tpb = (ntiles + S - 1) / S;
    S = (ntiles + tpb - 1) / tpb;
    *S_out = (int)S; *tpb_out = (int)tpb;
    return nci;
}

// =====================================================================================================================================
// wgrad16h: the same weight gradient for 16-bit tensors (bf16 / fp16 operands on v_mfma_f32_32x32x16_*, fp32 accumulation and slabs).
// Here the MFMA's K index holds 16 PIXELS and a lane's operand is 8 consecutive pixels of ONE channel -- a column of the [pixel][channel]
// image in LDS.  gfx950's ds_read_b64_tr_b16 delivers exactly that: per group of 16 lanes it reads 4 rows (pixels) x 16 columns (channels)
// and hands lane i column i; lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 (checked element by element on
// the hardware before this kernel was written).  Two such reads give a lane its 8 pixels; the image stays as buffer_load ... lds wrote it.
//   * block = 64 input channels x 64 output channels (one 128-byte-row plane each) and a share of the image's pixel tiles (8 rows x 32
//     columns); 12 waves = (input half, output half, kernel row), three 32 x 32 tap accumulators per wave for the whole launch;
//   * per tile the 10 x 34-pixel activation patch (43 KB) and the 256-pixel dy tile (32 KB) arrive by LDS-DMA, double buffered (152 KB);
//   * swizzle: 16-byte chunk position = chunk ^ 4 ((pixel >> 1) & 1): the four pixels of a transposed read then cover both halves of
//     the 256-byte bank row without meeting (pixels p and p + 2 use chunk sets 4 a .. 4 a + 3 and its complement);
//   * every operand address is a per-lane constant per (kernel column, row parity) plus an IMMEDIATE: the tile loop is straight-line
//     code with no address arithmetic (three copies, one per kernel row).
template <int PREC>
__device__ __forceinline__ void wgrad16h_body(const WgradParams& q, const int tpb) {
    constexpr int NW = 12, TH = 8;
    constexpr int XPIX = (TH + 2) * 34, XPC = (XPIX + 7) / 8, XP = XPC * 1024;      // 340 pixels -> 43 pieces
    constexpr int DPIX = TH * 32, DPC = DPIX / 8, DT = DPIX * 128;                  // 256 pixels -> 32 pieces
    constexpr int NPC = XPC + DPC, PPW = (NPC + NW - 1) / NW;                       // 75 pieces, 7 per wave
    constexpr int SS = XP + DT;
    static_assert(2 * SS <= 160 * 1024, "LDS");
    typedef typename LowPrec<PREC>::V8 LV8;
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const ConvParams& p = q.c;

    __shared__ __attribute__((aligned(1024))) char smem[2 * SS];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int kh = wid % 3, pl = wid / 3;
    const int ha = pl >> 1, hb = pl & 1;                                  // this wave's 32-channel half of the input / output plane
    const int Cin = p.C1 + p.C2;
    const int ncb = Cin / 64, nnb = (p.Cout + 63) / 64;                   // (Cout % 64 == 32: the last output plane is half filled)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % ncb, nb = (bid / ncb) % nnb, s = bid / (ncb * nnb);
    const int tw = p.W / 32, th = p.H / TH, tpi = tw * th, ntiles = p.B * tpi;
    const int t0 = s * tpb, t1 = min(ntiles, t0 + tpb);
    const bool idle = nb * 64 + hb * 32 >= p.Cout;                        // this wave's output half lies beyond Cout: it loads and meets the barriers only

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)q.dy, 0, p.M * p.Cout * 2, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int FAR = 0x40000000;

    auto issue_tile = [&](const int t, const int stage) {
        char* base = smem + stage * SS;
        const int b = t / tpi, tr = t - b * tpi;
        const int tx = tr / th, ty = tr - tx * th;                        // down a 32-pixel column strip (wgrad32h)
        const int c0 = cb * 64;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int id = jj * NW + wid;
            if (id >= NPC) break;
            if (id < XPC) {
                const int pix = 8 * id + (lane >> 3);
                const int sc16 = 16 * ((lane & 7) ^ (4 * ((pix >> 1) & 1)));
                const int hy = pix / 34, hx = pix - hy * 34;
                const int y = TH * ty - 1 + hy, x = 32 * tx - 1 + hx;
                const bool ok = pix < XPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                if (c0 < p.C1) {
                    const int pixoff = p.ups ? ((b * p.H1 + (y >> 1)) * p.W1 + (x >> 1)) : ((b * p.H + y) * p.W + x);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(base + id * 1024), 16, ok ? (pixoff * p.C1 + c0) * 2 + sc16 : FAR, 0, 0, 0);
                } else {
                    const int pixoff = (b * p.H + y) * p.W + x;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)(base + id * 1024), 16, ok ? (pixoff * p.C2 + (c0 - p.C1)) * 2 + sc16 : FAR, 0, 0, 0);
                }
            } else {
                const int pp = id - XPC;
                const int pix = 8 * pp + (lane >> 3);                     // row pix >> 5, column pix & 31 of the tile
                const int sc = (lane & 7) ^ (4 * ((pix >> 1) & 1)), sc16 = 16 * sc;
                const int m = (b * p.H + TH * ty + (pix >> 5)) * p.W + 32 * tx + (pix & 31);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr)(base + XP + pp * 1024), 16,
                                                         nb * 64 + 8 * sc < p.Cout ? (m * p.Cout + nb * 64) * 2 + sc16 : FAR, 0, 0, 0);
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kw][r] = 0.f;

    // transposed-read addressing of this lane: group g = lane >> 4 (channels 16 (g & 1) + ., pixels 8 (g >> 1) + .), i = lane & 15 -> row q = i >> 2,
    // columns 4 (i & 3) ..; chunk of the plane = 4 half + 2 (g & 1) + (p >> 1), 8 bytes into it for odd p
    const int g = lane >> 4, qi = (lane >> 2) & 3, pi = lane & 3;
    const int rowoff = (8 * (g >> 1) + qi) * 128 + 8 * (pi & 1);
    const int chA = 4 * ha + 2 * (g & 1) + (pi >> 1), chB = 4 * hb + 2 * (g & 1) + (pi >> 1);
    // key of pixel P = (patch row) * 34 + 16 blk + kw + 8 h + 4 rd + q is 4 ((row * 17 + ((kw + q) >> 1)) & 1): per (kw, row parity) a lane constant
    int offA[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int par = 0; par < 2; ++par) offA[kw][par] = rowoff + 16 * (chA ^ (4 * ((par + ((kw + qi) >> 1)) & 1)));
    const int offB = rowoff + 16 * (chB ^ (4 * ((qi >> 1) & 1))) + XP;

    auto tr8 = [&](const int addr) -> LV8 {          // 8 pixels of this lane's channel: two transposed reads, 4 pixels (512 bytes of image) apart
        const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(smem + addr));
        const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(smem + addr + 4 * 128));
        return __builtin_bit_cast(LV8, s8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };
    auto mma_tile_kh = [&](const int stage, auto khc) {
        constexpr int KHc = decltype(khc)::value;
        const int sb = stage * SS;
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const LV8 bv = tr8(sb + offB + (r * 32 + 16 * blk) * 128);
                LV8 av[3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) av[kw] = tr8(sb + offA[kw][(r + KHc) & 1] + ((r + KHc) * 34 + 16 * blk + kw) * 128);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[kw] = LowPrec<PREC>::mfma(av[kw], bv, acc[kw]);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto mma_tile = [&](const int stage) {
        if (kh == 0) mma_tile_kh(stage, std::integral_constant<int, 0>());
        else if (kh == 1) mma_tile_kh(stage, std::integral_constant<int, 1>());
        else mma_tile_kh(stage, std::integral_constant<int, 2>());
    };

    if (t0 < t1) {
        issue_tile(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int t = t0; t < t1; ++t) {
            const int stage = (t - t0) & 1;
            if (t + 1 < t1) issue_tile(t + 1, stage ^ 1);
            if (!idle) mma_tile(stage);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }

    if (idle) return;
    float* slab = q.ws + (size_t)s * ((size_t)p.K * p.Cout);
    const int co = nb * 64 + hb * 32 + li;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = cb * 64 + ha * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[((size_t)(kh * 3 + kw) * Cin + ci) * p.Cout + co] = acc[kw][r];
        }
}
template <int PREC>
__global__ __launch_bounds__(768, 3) void wgrad16h_kernel(WgradParams q, int tpb) {
    wgrad16h_body<PREC>(q, tpb);
}
// plan of the 16-bit form: 1 if applicable (64-channel planes -- the last output plane may be half filled --, 8-row tiles)
static int wgrad16h_plan(const ConvParams& p, int* S_out, int* tpb_out) {
    static const int on = ab_int("MMSEG_WGRAD16H", 1);
    if (!on || g_conv_bf16 == 0 || (p.io & 5) != 5 || g_conv16_mode == 0) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.W % 32 || p.H % 8 || p.C1 % 64 || p.C2 % 64 || p.Cout % 32) return 0;
    const long nbk = (long)((p.C1 + p.C2) / 64) * ((p.Cout + 63) / 64);
    const long ntiles = (long)p.B * (p.H / 8) * (p.W / 32);
    static const int target = ab_int("MMSEG_WGRAD16H_BLOCKS", 256);
    long S = (target + nbk - 1) / nbk;
    if (S > ntiles / 2) S = ntiles / 2;
    if (S < 1) S = 1;
    if (g_conv16_mode != 2 && nbk * S < 192) return 0;
    long tpb = (ntiles + S - 1) / S;
    S = (ntiles + tpb - 1) / tpb;
    *S_out = (int)S; *tpb_out = (int)tpb;
    return 1;
}
