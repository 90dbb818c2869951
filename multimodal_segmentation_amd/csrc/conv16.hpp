// Large-tile implicit-GEMM convolution for 16-bit tensors (bf16 / fp16 operands on v_mfma_f32_32x32x16_*, fp32 accumulation):
// the reduced-precision configurations of BASELINE.json (#3 DAFNet-SPADE bf16, #5 MMSDNet fp16) -- reference layers
// models/unet.py:94-101, layers/spade.py:26-33 (Conv2D 3x3 'same' on 64..1024 channels) and their data gradients.
//
// Why a second kernel: the 4-wave 128 x 128 kernel of conv.hip stages every tile through VGPRs (global load -> ds_write_b128) and
// synchronises once per 64-deep K tile with 16 MFMAs per wave in between; with 16-bit MFMAs (32 cycles each instead of 64 x 8) the
// LDS store path (~79 B/clk/CU for ds_write_b128) plus the LDS reads cost more cycles per K tile than the matrix pipe has work, and
// the single load stage in flight does not cover an L2 round trip: 0.15-0.23 of the 2.5 PFLOP/s peak (round 3).  Here:
//   * block tile 256 pixels x 256 / 128 / 64 output channels, K tile 64, 8 waves (4 for the 64-channel tile); a wave owns 128 x 64 or
//     64 x 64 outputs = 8 or 4 accumulator tiles -> 32 / 16 MFMAs per wave between two barriers, 6 or 4 LDS reads per 8 or 4 MFMAs;
//   * tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write), one 1 KB piece = 8 rows x 128
//     bytes per wave-instruction; the LDS image is lane-linear, so the bank swizzle of the fragment reads ((row >> 1) & 7 on the
//     16-byte chunk index, conflict-free for ds_read_b128) is applied to each lane's SOURCE address; image borders and rows beyond
//     the tensors use out-of-range buffer offsets, which the hardware turns into zeros in LDS;
//   * a ring of NS LDS stages filled NS - 1 tiles ahead; one counted `s_waitcnt vmcnt` + one raw s_barrier per K tile: the wait
//     retires the tile needed NEXT (issued a whole iteration earlier), the youngest tile stays in flight across the barrier;
//   * XCD-aware block order, tap-fastest K order and 2-D pixel tiles as in conv_fast_body.
// Hazards.  RAW: a stage is read only after every wave waited for its own pieces of it (vmcnt) and passed the barrier behind that
// wait.  WAR: a stage is refilled one barrier after the iteration that read it, and that barrier is preceded by lgkmcnt(0).
#pragma once

template <int BM, int BN, int WM, int WN, int NS, int PREC>
__device__ __forceinline__ void conv16_body(const ConvParams& p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KT = 64, ROWB = 128;                        // K tile (elements), bytes per LDS row
    constexpr int AG = BM / 8 / NW, BG = BN / 8 / NW;         // 8-row pieces per wave and tile: activations, weights
    constexpr int LPT = AG + BG;                              // LDS-DMA instructions per wave (= per lane) and tile
    constexpr int STAGE = (BM + BN) * ROWB;                   // bytes per ring stage
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces divide evenly over the waves");
    static_assert(NS == 2 || NS == 3, "ring depth");
    static_assert(NS * STAGE <= 160 * 1024, "LDS");
    typedef typename LowPrec<PREC>::V8 LV8;
    typedef typename LowPrec<PREC>::T LT;

    __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];          // ONE shared object (guide: a second one de-pipelines the loop)

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = gridDim.x;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const int ntm = nblk / ntn;
    const bool w_major = (long)p.K * p.Cout > (long)p.M * (p.C1 + p.C2);      // consecutive ids share the bigger operand (conv_fast_body)
    const int mt = w_major ? lb % ntm : lb / ntn;
    const int n0 = (w_major ? lb / ntm : lb % ntn) * BN;
    constexpr int TH = BM / 16;
    const bool tile2d = (p.Wo % 16 == 0) && (p.Ho % TH == 0);
    const int tpr = tile2d ? p.Wo / 16 : 1, tpi = tile2d ? (p.Ho / TH) * tpr : 1;
    const int t_b = tile2d ? mt / tpi : 0, t_r = tile2d ? mt - t_b * tpi : 0;
    const int t_y0 = tile2d ? (t_r / tpr) * TH : 0, t_x0 = tile2d ? (t_r % tpr) * 16 : 0;
    const int m0 = mt * BM;
    auto row_to_m = [&](int row) -> int {
        return tile2d ? (t_b * p.Ho + t_y0 + (row >> 4)) * p.Wo + t_x0 + (row & 15) : m0 + row;
    };

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.K * p.Cout * 2, 0x00020000);

    // ---- per-lane staging state.  Piece g of a tile = rows 8 g .. 8 g + 7; lane l fills LDS chunk (l & 7) of row 8 g + (l >> 3) with
    // SOURCE chunk sc = (l & 7) ^ swz(row): the read side looks chunk j up at position j ^ swz(row).  swz(row) = (row >> 1) & 7 =
    // (4 g + (l >> 4)) & 7, so sc = lx ^ (4 (g & 1)) with lx = (l & 7) ^ (l >> 4).
    // Per row: a_o1 / a_o2 = BYTE offset of (b, hb, wb, source chunk) in x1 / x2 (x1 with up-sampling: of the low-resolution pixel
    // (hb >> 1, wb >> 1)), a_msk = validity bit per tap (bit kh * KW + kw; the host keeps KH * KW <= 30) and, with up-sampling, the
    // parities of hb / wb in bits 30 / 31: (hb + kh) >> 1 = (hb >> 1) + ((hb & 1) + kh) >> 1, so a tap's offset is a_o1 + a small
    // per-lane delta.  The issue of a tile is then one add and one select per 1 KB piece (the loop is bound by instruction issue, not
    // by the matrix pipe, if it carries more: measured 8 VALU + 7 SALU per MFMA in the first version of this kernel).
    const int HoWo = p.Ho * p.Wo;
    const int lx = (lane & 7) ^ (lane >> 4);
    int a_o1[AG], a_o2[AG];
    unsigned a_msk[AG];
#pragma unroll
    for (int j = 0; j < AG; ++j) {
        const int g = wid * AG + j;
        const int row = 8 * g + (lane >> 3);
        const int sc16 = 16 * (lx ^ (4 * (g & 1)));
        const int m = row_to_m(row);
        a_msk[j] = 0u; a_o1[j] = 0; a_o2[j] = 0;
        if (m < p.M) {
            int b, ho, wo;
            if (tile2d) { b = t_b; ho = t_y0 + (row >> 4); wo = t_x0 + (row & 15); }
            else { b = m / HoWo; const int r = m - b * HoWo; ho = r / p.Wo; wo = r - ho * p.Wo; }
            const int hb = ho * p.stride - p.pad_h, wb = wo * p.stride - p.pad_w;
            unsigned wm_ = 0u, msk = 0u;
            for (int kw = 0; kw < p.KW; ++kw)
                if ((unsigned)(wb + kw) < (unsigned)p.W) wm_ |= 1u << kw;
            for (int kh = 0; kh < p.KH; ++kh)
                if ((unsigned)(hb + kh) < (unsigned)p.H) msk |= wm_ << (kh * p.KW);
            if (p.ups) {
                a_o1[j] = ((b * p.H1 + (hb >> 1)) * p.W1 + (wb >> 1)) * p.C1 * 2 + sc16;
                msk |= ((unsigned)(hb & 1) << 30) | ((unsigned)(wb & 1) << 31);
            } else {
                a_o1[j] = ((b * p.H + hb) * p.W + wb) * p.C1 * 2 + sc16;
            }
            a_o2[j] = ((b * p.H + hb) * p.W + wb) * p.C2 * 2 + sc16;
            a_msk[j] = msk;
        }
    }
    constexpr int FAR = 0x40000000;
    int b_o[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        const int g = wid * BG + j;
        const int n = n0 + 8 * g + (lane >> 3);
        b_o[j] = n < p.Cout ? (n * p.K + 8 * (lx ^ (4 * (g & 1)))) * 2 : FAR;
    }
    // wave-uniform position of the next K tile to issue: tap (s_kh, s_kw), channel base s_c0, and what follows from them
    int s_c0 = 0, s_kh = 0, s_kw = 0, s_tap = 0;
    const int Cin2 = (p.C1 + p.C2) * 2;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_tile = [&](const int stage, const bool live) {
        char* base = smem + stage * STAGE + wid * (AG * 1024);
        char* bbase = smem + stage * STAGE + BM * ROWB + wid * (BG * 1024);
        const unsigned tbit = live ? 1u << s_tap : 0u;
        const int b_koff = live ? s_tap * Cin2 + s_c0 * 2 : FAR;
        if (s_c0 < p.C1) {                                    // the tile lies in x1 (wave-uniform)
            if (p.ups) {
#pragma unroll
                for (int j = 0; j < AG; ++j) {
                    const int dh = ((int)((a_msk[j] >> 30) & 1u) + s_kh) >> 1, dw = ((int)(a_msk[j] >> 31) + s_kw) >> 1;
                    const int off = a_o1[j] + ((dh * p.W1 + dw) * p.C1 + s_c0) * 2;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(base + j * 1024), 16, (a_msk[j] & tbit) ? off : BUF_OOB, 0, 0, 0);
                }
            } else {
                const int toff = ((s_kh * p.W + s_kw) * p.C1 + s_c0) * 2;
#pragma unroll
                for (int j = 0; j < AG; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(base + j * 1024), 16, (a_msk[j] & tbit) ? a_o1[j] + toff : BUF_OOB, 0, 0, 0);
            }
        } else {
            const int toff = ((s_kh * p.W + s_kw) * p.C2 + (s_c0 - p.C1)) * 2;
#pragma unroll
            for (int j = 0; j < AG; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)(base + j * 1024), 16, (a_msk[j] & tbit) ? a_o2[j] + toff : BUF_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BG; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(bbase + j * 1024), 16, b_o[j] + b_koff, 0, 0, 0);
        // taps fastest inside a 64-channel chunk (consecutive tiles re-read the same cache lines, shifted)
        ++s_tap;
        if (++s_kw == p.KW) { s_kw = 0; if (++s_kh == p.KH) { s_kh = 0; s_tap = 0; s_c0 += KT; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wm * (BM / WM) + li;
    const int b_col = wn * (BN / WN) + li;
    auto mma_tile = [&](const int stage) {
        const char* A = smem + stage * STAGE;
        const char* Bt = A + BM * ROWB;
        // k-steps of 16: lane half lh supplies k = 16 q + 8 lh + [0, 8) = chunk 2 q + lh.  The fragments of step q + 1 are requested
        // before the MFMAs of step q (two register sets with static names): the LDS latency sits under 8 / 4 MFMAs, and the
        // register budget (acc 128 + 2 x 24) stays inside 256 at two waves per SIMD
        LV8 a[2][TM], b[2][TN];
        auto frags = [&](const int set, const int q) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = b_col + j * 32;
                b[set][j] = *reinterpret_cast<const LV8*>(Bt + row * ROWB + 16 * ((2 * q + lh) ^ ((row >> 1) & 7)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = a_row + i * 32;
                a[set][i] = *reinterpret_cast<const LV8*>(A + row * ROWB + 16 * ((2 * q + lh) ^ ((row >> 1) & 7)));
            }
        };
        frags(0, 0);
#pragma unroll
        for (int q = 0; q < KT / 16; ++q) {
            if (q + 1 < KT / 16) frags((q + 1) & 1, q + 1);
            // pin the order: without this hipcc sinks the reads of step q + 1 below the MFMAs of step q (one register set, every
            // k-step then starts with an exposed LDS round trip -- measured: the matrix pipe idle 65 % of the time with NO loads at all)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = LowPrec<PREC>::mfma(a[q & 1][i], b[q & 1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nkt = p.K / KT;
    // ---- prologue: NS - 1 tiles in flight, the first one landed
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_tile(s, s < nkt);
    if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int cur = 0, nxt = NS - 1;                       // stage of tile kt; stage the next issue goes to
    // The two waves that share a SIMD (w and w + NW / 2) run the two halves of an iteration -- queueing the next tile's loads (address
    // arithmetic, vector-memory issue) and the MFMAs of the current tile -- in OPPOSITE order, so that one of them feeds the matrix
    // pipe while the other computes addresses; both orders are legal between the same two barriers (the refilled stage was read one
    // iteration ago, the stage being read is complete) and leave the same loads outstanding at the counted wait.
    const bool mma_first = NW >= 8 && wid >= NW / 2;
    for (int kt = 0; kt < nkt; ++kt) {
        // (tiles beyond K: out-of-range offsets = zeros, no traffic; the outstanding-load count stays exact)
#ifdef MMSEG_AB
        const int abl = p.transposed;        // measurement builds: 1 = no loads in the loop, 2 = no MFMAs / fragment reads (results are wrong)
        if (!mma_first && abl != 1) issue_tile(nxt, kt + NS - 1 < nkt);
        if (abl != 2) mma_tile(cur);
        if (mma_first && abl != 1) issue_tile(nxt, kt + NS - 1 < nkt);
#else
        if (!mma_first) issue_tile(nxt, kt + NS - 1 < nkt);
        mma_tile(cur);
        if (mma_first) issue_tile(nxt, kt + NS - 1 < nkt);
#endif
        // tile kt + 1 has landed for this wave once all but the youngest LPT * (NS - 2) loads are done; the barrier makes that true of
        // every wave's pieces and tells everybody that stage `cur` has been read (it is refilled by the next issue)
        if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (zero-fill tiles issued past the end)

    // ---- epilogue: conv_fast_body's quad-transposed stores (a lane ends up with 4 consecutive channels of one row), with the row
    // arithmetic written out as the affine map it is -- m(i, g) = mb + i * si + (g >> 1) * sg + 8 (g & 1) for accumulator tile row
    // i and register group g, both for the 2-D pixel tiles and for raster rows -- so that nothing but `acc` is live across it (the
    // generic form made the compiler precompute every row of the 8 tiles and park the accumulators in scratch).  The launcher only
    // sends launches without an output mapping and with aligned outputs here.
    {
        const int tq = lane & 3;
        const int rb = wm * (BM / WM) + 4 * lh + tq;
        const int mb = tile2d ? (t_b * p.Ho + t_y0 + (rb >> 4)) * p.Wo + t_x0 + (rb & 15) : m0 + rb;
        const int si = tile2d ? 2 * p.Wo : 32, sg = tile2d ? p.Wo : 16;
        const int c1 = p.y2 == nullptr ? p.Cout : p.nsplit1, c2 = p.Cout - c1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + li;
            const bool nok = n < p.Cout;                 // uniform inside a quad (Cout % 4 == 0)
            const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
            const float sv = (nok && p.oscale) ? p.oscale[n] : 1.f;
            const int nq = n - tq;
            const bool first = nq < c1;
            char* const obase = reinterpret_cast<char*>(first ? (void*)p.y : (void*)p.y2);
            const int ocol = first ? nq : nq - c1, ocn = first ? c1 : c2;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = act_apply(acc[i][j][4 * g + e] * sv + bv, p.act, p.alpha);
                    quad_transpose4(a, tq);
                    const int m = mb + i * si + (g >> 1) * sg + 8 * (g & 1);
                    if (!nok || m >= p.M) continue;
#ifdef MMSEG_AB
                    if (p.transposed == 3 && a[0] != 12345.678f) continue;       // measurement builds: no output stores
#endif
                    store4_out(obase, (size_t)m * ocn + ocol, a, p.io);
                }
            }
        }
    }
}

// (the body lives in a __device__ function: the host pass must be able to instantiate the __global__ stub without the gfx950 builtins)
template <int BM, int BN, int WM, int WN, int NS, int PREC>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 4) void conv16_kernel(ConvParams p) {
    conv16_body<BM, BN, WM, WN, NS, PREC>(p);
}

// 0: never use this kernel; 1 (default): where it pays; 2: wherever it applies (tests put small problems on it) -- mmseg_conv16_mode
static int g_conv16_mode = 1;
// tile choice.  0 = leave the layer to conv_fast_kernel (too few 256-pixel tiles to fill the chip, or K tiles that straddle tensors)
// what both kernels need: 16-bit inputs, channel counts multiples of 64 (a K tile inside one tap of one tensor), no output mapping,
// aligned outputs (the lean epilogue)
static bool conv16_applicable(const ConvParams& p) {
    if (g_conv16_mode == 0) return false;
    if (!(p.io & 1) || (p.C2 && !(p.io & 2)) || p.C1 % 64 || p.C2 % 64 || p.Cout % 8 || p.KH * p.KW > 30) return false;
    const bool omap = p.osh != 1 || p.osw != 1 || p.ooh != 0 || p.oow != 0 || p.oH != p.Ho || p.oW != p.Wo;
    return !(omap || !p.qepi || (p.y2 != nullptr && p.nsplit1 % 4));
}
static int conv16_tile(const ConvParams& p) {
    if (!conv16_applicable(p)) return 0;
    static const int force = ab_int("MMSEG_CONV16", -1);      // measurement builds: 0 off, 64 / 128 / 256 force a tile
    if (force == 0) return 0;
    const long mt = (p.M + 255) / 256;
    int bn = p.Cout > 128 ? 256 : (p.Cout > 64 ? 128 : 64);
    if (force > 0) bn = force;
    // mode 1: only where it measured faster than conv_fast_kernel (tools/conv16_bench.py, profiles/r04_conv16_ab.txt): at least 3/4 of a
    // wave of 256-pixel blocks on the 256 CUs, and more than 64 output channels (with a 64-wide tile every wave re-reads the whole weight
    // tile: LDS-read bound, 0.6 - 0.9 x the 128 x 64 tile of the register-staged kernel, which runs 3 blocks per CU)
    if (force < 0 && g_conv16_mode != 2 && (mt * ((p.Cout + bn - 1) / bn) < 192 || bn == 64)) return 0;
    return bn;
}
template <int PREC>
static int launch_conv16_prec(const ConvParams& p_, int bn, hipStream_t st) {
    ConvParams p = p_;
    p.transposed = ab_int("MMSEG_CONV16_ABL", 0);       // (always 0 on this path; measurement builds: ablation code)
    const int ntm = (p.M + 255) / 256;
    if (bn == 256) hipLaunchKernelGGL((conv16_kernel<256, 256, 2, 4, 2, PREC>), dim3(ntm * ((p.Cout + 255) / 256)), dim3(512), 0, st, p);
    else if (bn == 128) hipLaunchKernelGGL((conv16_kernel<256, 128, 4, 2, 3, PREC>), dim3(ntm * ((p.Cout + 127) / 128)), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv16_kernel<256, 64, 4, 1, 3, PREC>), dim3(ntm * ((p.Cout + 63) / 64)), dim3(256), 0, st, p);
    return MMSEG_CHECK_LAUNCH();
}

// =====================================================================================================================================
// conv16h: the 3x3 'same' stride-1 case with the activation patch RESIDENT in LDS.
//
// Measured on conv16_kernel (tools/abl_conv16.sh, profiles/r04_conv16_ablation.txt): per 64-deep K tile a 256 x 128 block moves 48 KB
// through the CU's texture addresser (64 B/clk -> 768 cycles against 1 024 cycles of MFMA work), 32 KB of it activations -- and the nine
// taps of a 64-channel chunk are nine shifted copies of the SAME pixels.  Here a block owns 8 image rows x 32 columns; per 64-channel
// chunk the 10 x 34-pixel halo patch (43.5 KB) is brought into LDS once (1 KB pieces spread over the chunk's nine K tiles, into the
// other of two patch buffers) and the A fragments of tap (kh, kw) are read from it at pixel (y + kh, x + kw): activation traffic / 6.6,
// loads per wave and K tile 6 -> 3 (N = 128), no per-tap address or validity arithmetic on the load side at all (a patch pixel outside
// the image has an out-of-range offset for the whole launch).  An MFMA tile's 32 rows are 32 consecutive pixels of one image row, so the
// 16-lane groups of ds_read_b128 see 16 distinct (pixel & 15) residues for every tap shift: with the chunk swizzle (pixel >> 1) & 7 the
// reads are conflict-free (same argument as for the weight rows).  Weights stream through a ring of NS stages as in conv16_kernel.
// Geometry: W % 32 == 0, H % 8 == 0, C1 % 64 == C2 % 64 == 0, optional nearest x2 up-sampling of x1 and concatenation with x2.
// SUP = K tiles per barrier.  SUP = 2 (N tiles of 128 / 64 channels: their weight stages are small enough for four ring slots): two taps
// are queued, multiplied (one chain of 8 k-steps: the fragment pipeline does not restart between them) and awaited together -- 32 / 16
// MFMAs per wave between two barriers instead of 16 / 8.  The patch piece loaded during tap t is piece t - 2 (taps 2 .. 7): a chunk's
// first tap may share a barrier interval with the previous chunk's last tap, which still reads the buffer being refilled.
// TH = image rows per block (8 or 16), PB = patch buffers.  The 64-channel N tile uses 16 rows x 32 columns with the 8 waves stacked along M
// (a wave: 2 image rows x 64 channels = the 2 x 2 accumulator tiles of the 128-channel form, 4 fragment reads per 4 MFMAs; with 8 rows a wave
// would own 64 x 32 outputs and read 3 fragments per 2 MFMAs -- LDS-read bound, 0.9 x conv_fast_kernel).  Its 18 x 34 patch (77 KB) is single
// buffered (PB = 1): at a chunk boundary the block reloads it behind a barrier; the N = 64 layers have one or two 64-channel chunks.
template <int BN, int WM, int WN, int NS, int PREC, int SUP = 1, int TH = 8, int PB = 2>
__device__ __forceinline__ void conv16h_body(const ConvParams& p) {
    constexpr int TW = 32, NW = WM * WN;
    constexpr int TM = TH / WM, TN = BN / WN / 32;
    constexpr int HWD = TW + 2, HP = (TH + 2) * HWD, NP = (HP + 7) / 8;      // patch: 34 pixels wide, 340 pixels, 43 pieces of 8
    constexpr int PPW = (NP + NW - 1) / NW;                                   // patch pieces per wave (6)
    constexpr int HALO = NP * 1024, BST = BN * 128, BG = BN / 8 / NW;
    constexpr int ES = PREC ? 2 : 4;                                           // bytes per element: 16-bit modes / fp32 (PREC 0)
    constexpr int KT = 128 / ES;                                               // channels per chunk: one 128-byte LDS row per pixel
    constexpr int NST = NS * SUP;                                              // ring slots (one K tile each)
    constexpr int PT0 = SUP > 1 ? 2 : 0;                                       // first tap during which a patch piece is loaded
    static_assert(NW == 8 && TH % WM == 0 && BN % (8 * NW) == 0 && (PB == 1 || PT0 + PPW <= 9), "8 waves; one patch piece per wave and tap");
    static_assert(PB == 2 || SUP == 1, "single patch buffer: one K tile per barrier");
    static_assert(SUP == 1 || NS == 2, "several tiles per barrier: everything queued is awaited at the barrier");
    static_assert(PB * HALO + NST * BST <= 160 * 1024, "LDS");
    typedef typename LowPrec<PREC>::V8 LV8;

    __shared__ __attribute__((aligned(1024))) char smem[PB * HALO + NST * BST];
    char* const bring = smem + PB * HALO;

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = gridDim.x;
    const int ntn = (p.Cout + BN - 1) / BN, ntm = nblk / ntn;
    const int lb = xcd_remap(blockIdx.x, nblk);
    const bool w_major = (long)p.K * p.Cout > (long)p.M * (p.C1 + p.C2);
    const int mt = w_major ? lb % ntm : lb / ntn;
    const int n0 = (w_major ? lb / ntm : lb % ntn) * BN;
    const int tpr = p.W / TW, tpi = (p.H / TH) * tpr;
    const int t_b = mt / tpi, t_r = mt - t_b * tpi;
    const int y0 = (t_r / tpr) * TH, x0 = (t_r % tpr) * TW;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.B * p.H1 * p.W1 * p.C1 * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 ? p.x2 : p.x1), 0,
                                                                        p.C2 ? p.B * p.H * p.W * p.C2 * ES : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, p.K * p.Cout * ES, 0x00020000);
    constexpr int FAR = 0x40000000;

    // ---- patch pieces of this lane: piece pc = jj * NW + wid (jj = the tap during which it is loaded), pixel hp = 8 pc + (lane >> 3),
    // LDS chunk (lane & 7) <- source chunk (lane & 7) ^ ((hp >> 1) & 7); byte offsets in x1 / x2, FAR for pixels outside the image
    int h_o1[PPW], h_o2[PPW];
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
        const int pc = jj * NW + wid;
        const int hp = 8 * pc + (lane >> 3);
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const int sc16 = 16 * ((lane & 7) ^ ((hp >> 1) & 7));
        const bool ok = hp < HP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        h_o1[jj] = ok ? (p.ups ? ((t_b * p.H1 + (y >> 1)) * p.W1 + (x >> 1)) : ((t_b * p.H + y) * p.W + x)) * p.C1 * ES + sc16 : FAR;
        h_o2[jj] = (ok && p.C2) ? ((t_b * p.H + y) * p.W + x) * p.C2 * ES + sc16 : FAR;
    }
    int b_o[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        const int g = wid * BG + j;
        const int n = n0 + 8 * g + (lane >> 3);
        b_o[j] = n < p.Cout ? n * p.K * ES + 16 * ((lane & 7) ^ ((4 * g + (lane >> 4)) & 7)) : FAR;
    }
    const int Cin2 = (p.C1 + p.C2) * ES;
    const int nch = (p.C1 + p.C2) / KT, nkt = nch * 9;

    typedef __attribute__((address_space(3))) void* lds_ptr;
    // one patch piece of chunk c (wave-uniform guards: the piece exists, the chunk exists) -> 1 if a load was issued
    auto issue_patch = [&](const int c, const int jj) -> int {
        if (c >= nch || jj >= PPW || jj * NW + wid >= NP) return 0;
        char* dst = smem + (PB == 2 ? (c & 1) * HALO : 0) + (jj * NW + wid) * 1024;
        const int c0 = c * KT;
        int off = 0;
#pragma unroll
        for (int q = 0; q < PPW; ++q) if (q == jj) off = c0 < p.C1 ? h_o1[q] : h_o2[q];      // (static register names)
        if (c0 < p.C1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)dst, 16, off + c0 * ES, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr)dst, 16, off + (c0 - p.C1) * ES, 0, 0, 0);
        return 1;
    };
    auto issue_b = [&](const int stage, const int c, const int tap, const bool live) {
        char* dst = bring + stage * BST + wid * (BG * 1024);
        const int koff = live ? tap * Cin2 + c * 128 : FAR;
#pragma unroll
        for (int j = 0; j < BG; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + j * 1024), 16, b_o[j] + koff, 0, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int b_col = wn * (BN / WN) + li;
    // A fragment of row-block i (image row y0 + wm * TM + i), tap (kh, kw), k-step q: patch pixel hp = (wm TM + i + kh) * 34 + li + kw,
    // chunk (2 q + lh) ^ key with key = (hp >> 1) & 7.  SUP tiles form one chain of 4 SUP k-steps.
    struct Tile { int hbuf, slot, kh, kw; bool live; };
    auto mma_tiles = [&](const Tile (&tl)[SUP]) {
        int abase[SUP][TM], akey[SUP][TM];
#pragma unroll
        for (int u = 0; u < SUP; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int hp = (wm * TM + i + tl[u].kh) * HWD + li + tl[u].kw;
                abase[u][i] = tl[u].hbuf * HALO + hp * 128;
                akey[u][i] = (hp >> 1) & 7;
            }
        LV8 a[2][TM], b[2][TN];
        auto frags = [&](const int set, const int qq) {
            const int u = qq / 4, q = qq % 4;
            const char* Bt = bring + tl[u].slot * BST;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = b_col + j * 32;
                b[set][j] = *reinterpret_cast<const LV8*>(Bt + row * 128 + 16 * ((2 * q + lh) ^ ((row >> 1) & 7)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) a[set][i] = *reinterpret_cast<const LV8*>(smem + abase[u][i] + 16 * ((2 * q + lh) ^ akey[u][i]));
        };
        frags(0, 0);
#pragma unroll
        for (int qq = 0; qq < 4 * SUP; ++qq) {
            if (qq + 1 < 4 * SUP) frags((qq + 1) & 1, qq + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (SUP == 1 || tl[qq / 4].live) {           // (a tile beyond K: its slot and patch hold finite data, the products are skipped)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (PREC != 0) {
                            acc[i][j] = LowPrec<PREC>::mfma(a[qq & 1][i], b[qq & 1][j], acc[i][j]);
                        } else {
                            // fp32 (v_mfma_f32_32x32x2_f32): lane half lh read chunk 2 q + lh; element t of both operands is
                            // k = 4 (2 q + lh) + t, so the four MFMAs of a chunk pair cover its 8 k-values
#pragma unroll
                            for (int t4 = 0; t4 < 4; ++t4)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[qq & 1][i][t4], b[qq & 1][j][t4], acc[i][j], 0, 0, 0);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: the whole patch of chunk 0, the first (NS - 1) SUP weight tiles
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) issue_patch(0, jj);
#pragma unroll
    for (int u = 0; u < (NS - 1) * SUP; ++u) issue_b(u, u / 9, u % 9, u < nkt);
    if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BG) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // position of K tile t: chunk c, tap (kh, kw); of the first weight tile queued in this interval (tile t + (NS - 1) SUP): (ci, ti)
    int c = 0, kh = 0, kw = 0, tap = 0;
    int ci = ((NS - 1) * SUP) / 9, ti = ((NS - 1) * SUP) % 9;
    int cur = 0, nxt = ((NS - 1) * SUP) % NST;
    const bool mma_first = wid >= NW / 2;      // SIMD partners queue loads and MFMAs in opposite order (conv16_kernel)
    for (int t = 0; t < nkt; t += SUP) {
        Tile tl[SUP];
        int pc_[SUP], pt_[SUP];                // (chunk whose patch is being filled, piece) per tile of the interval
        {
            int c_ = c, kh_ = kh, kw_ = kw, tap_ = tap, slot = cur;
#pragma unroll
            for (int u = 0; u < SUP; ++u) {
                tl[u].hbuf = PB == 2 ? (c_ & 1) : 0; tl[u].slot = slot; tl[u].kh = kh_; tl[u].kw = kw_; tl[u].live = t + u < nkt;
                pc_[u] = c_ + 1; pt_[u] = tap_ - PT0;
                slot = slot + 1 == NST ? 0 : slot + 1;
                ++tap_;
                if (++kw_ == 3) { kw_ = 0; if (++kh_ == 3) { kh_ = 0; tap_ = 0; ++c_; } }
            }
            c = c_; kh = kh_; kw = kw_; tap = tap_; cur = slot;
        }
        if constexpr (PB == 1) {
            if (tl[0].kh == 0 && tl[0].kw == 0 && t > 0) {
                // a new 64-channel chunk: every wave has finished the previous chunk's taps (barrier of the last iteration); refill the
                // one patch buffer and wait for it -- the only point where this form exposes a load latency inside the K loop
                const int cn = t / 9;
#pragma unroll
                for (int jj = 0; jj < PPW; ++jj) issue_patch(cn, jj);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        int hp_issued = 0;
        auto issue_all = [&]() {
            int ci_ = ci, ti_ = ti, slot = nxt;
#pragma unroll
            for (int u = 0; u < SUP; ++u) {
                if (PB == 2 && pt_[u] >= 0 && tl[u].live) hp_issued += issue_patch(pc_[u], pt_[u]);
                issue_b(slot, ci_, ti_, t + (NS - 1) * SUP + u < nkt);
                slot = slot + 1 == NST ? 0 : slot + 1;
                if (++ti_ == 9) { ti_ = 0; ++ci_; }
            }
            ci = ci_; ti = ti_; nxt = slot;
        };
        if (!mma_first) issue_all();
        mma_tiles(tl);
        if (mma_first) issue_all();
        // NS == 3 (SUP == 1): weight tile t + 1 was queued one interval ago; younger than it are this interval's patch piece and weight
        // tile; the patch pieces of chunk c + 1 are all older than the weight tile awaited at the end of tap 8 (in-order retirement).
        // NS == 2: everything queued is awaited.
        if constexpr (NS == 3) {
            if (hp_issued) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(BG + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(BG) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: accumulator tile (i, j) = 32 pixels of image row y0 + wm TM + i x 32 channels; quad-transposed 4-channel stores
    {
        const int tq = lane & 3;
        const int c1 = p.y2 == nullptr ? p.Cout : p.nsplit1, c2 = p.Cout - c1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + li;
            const bool nok = n < p.Cout;
            const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
            const float sv = (nok && p.oscale) ? p.oscale[n] : 1.f;
            const int nq = n - tq;
            const bool first = nq < c1;
            char* const obase = reinterpret_cast<char*>(first ? (void*)p.y : (void*)p.y2);
            const int ocol = first ? nq : nq - c1, ocn = first ? c1 : c2;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mrow = (t_b * p.H + y0 + wm * TM + i) * p.W + x0 + 4 * lh + tq;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = act_apply(acc[i][j][4 * g + e] * sv + bv, p.act, p.alpha);
                    quad_transpose4(a, tq);
                    if (!nok) continue;
                    store4_out(obase, (size_t)(mrow + 8 * g) * ocn + ocol, a, p.io);
                }
            }
        }
    }
}
template <int BN, int WM, int WN, int NS, int PREC, int SUP = 1, int TH = 8, int PB = 2>
__global__ __launch_bounds__(512, 2) void conv16h_kernel(ConvParams p) {
    conv16h_body<BN, WM, WN, NS, PREC, SUP, TH, PB>(p);
}
// fp32 tensors (PREC 0: rows of 32 channels): the same kernel on v_mfma_f32_32x32x2_f32
static bool conv32h_applicable(const ConvParams& p) {
    if (g_conv16_mode == 0 || p.io != 0) return false;
    if (p.C1 % 32 || p.C2 % 32 || p.Cout % 8) return false;
    const bool omap = p.osh != 1 || p.osw != 1 || p.ooh != 0 || p.oow != 0 || p.oH != p.Ho || p.oW != p.Wo;
    return !(omap || !p.qepi || (p.y2 != nullptr && p.nsplit1 % 4));
}
// applicability of the patch-resident kernel; 0 or the N tile
// share of the last round of blocks that is filled on 256 CUs (one block per CU): 384 blocks = 1.5 rounds -> 0.75
static double conv16h_tail(long blocks) {
    if (blocks <= 0) return 0.0;
    const long rounds = (blocks + 255) / 256;
    return (double)blocks / (double)(rounds * 256);
}
static bool conv16h_grid_ok(long blocks) { return g_conv16_mode == 2 || (blocks >= 192 && conv16h_tail(blocks) >= 0.8); }
static int conv16h_tile(const ConvParams& p, bool fp32 = false) {
    if (!(fp32 ? conv32h_applicable(p) : conv16_applicable(p))) return 0;
    static const int hmode = ab_int("MMSEG_CONV16H", 1);     // measurement builds: 0 = never
    if (!hmode) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.W % 32 || p.H % 8) return 0;
    const int bn = p.Cout > 128 ? 256 : (p.Cout > 64 ? 128 : 64);
    const long ntn = (p.Cout + bn - 1) / bn;
    // mode 1: where it measured faster than conv_fast_kernel (tools/conv16_bench.py, DTYPE=f32 for the fp32 instances): grids that fill
    // their last round of blocks; in fp32 more than 32 output channels (a 64-wide tile on 32 channels multiplies half of it by zero:
    // 0.63 x in fp32, still 1.16 x in bf16 where the alternative is further from its roofline)
    if (g_conv16_mode != 2 && fp32 && p.Cout <= 32) return 0;
    if (bn == 64) {                                          // 16 image rows per block (the 64-channel form)
        static const int h64 = ab_int("MMSEG_CONV16H_N64", 1);
        if (!h64 || p.H % 16) return 0;
        return conv16h_grid_ok((long)(p.M / 512)) ? 64 : 0;
    }
    if (bn == 128 && p.H % 16 == 0 && conv16h_grid_ok((long)(p.M / 512) * ntn)) return bn;
    return conv16h_grid_ok((long)(p.M / 256) * ntn) ? bn : 0;
}
// image rows per block of the patch-resident kernel for this launch: 16 for the 64-channel form; for 128 channels 16 where the image
// and the grid allow (a wave then owns 4 rows x 64 channels = 128 x 64 outputs, 6 fragment reads per 8 MFMAs and 32 MFMAs between two
// barriers: 1.17 - 1.21 x the 8-row form, tools/ab_th16.sh), else 8
static int conv16h_rows(const ConvParams& p, int bn) {
    if (bn == 64) return 16;
    if (bn == 128 && p.H % 16 == 0 && ab_int("MMSEG_CONV16H_TH16", 1) && conv16h_grid_ok((long)(p.M / 512) * ((p.Cout + 127) / 128))) return 16;
    return 8;
}
template <int PREC>
static int launch_conv16h_prec(const ConvParams& p, int bn, hipStream_t st) {
    const int ntm = p.M / 256;
    if (bn == 256) hipLaunchKernelGGL((conv16h_kernel<256, 2, 4, 2, PREC>), dim3(ntm * ((p.Cout + 255) / 256)), dim3(512), 0, st, p);
    // (two K tiles per barrier, SUP = 2, measured within +-2 % of one per barrier on every shape -- tools/ab_sup.sh, profiles/r04_conv16_notes.txt
    // -- so the product keeps the smaller-LDS three-stage form; the SUP = 2 instances exist in measurement builds only)
#ifdef MMSEG_AB
    else if (ab_int("MMSEG_CONV16H_SUP", 1) == 2 && bn == 128)
        hipLaunchKernelGGL((conv16h_kernel<128, 4, 2, 2, PREC, 2>), dim3(ntm * ((p.Cout + 127) / 128)), dim3(512), 0, st, p);
#endif
    else if (bn == 128 && conv16h_rows(p, bn) == 16)
        hipLaunchKernelGGL((conv16h_kernel<128, 4, 2, 3, PREC, 1, 16, 1>), dim3((p.M / 512) * ((p.Cout + 127) / 128)), dim3(512), 0, st, p);
    else if (bn == 128) hipLaunchKernelGGL((conv16h_kernel<128, 4, 2, 3, PREC>), dim3(ntm * ((p.Cout + 127) / 128)), dim3(512), 0, st, p);
#ifdef MMSEG_AB
    else if (ab_int("MMSEG_CONV16H_N64_TH8", 0))        // measurement builds: 8 rows, two patch buffers, wave tile 64 x 32
        hipLaunchKernelGGL((conv16h_kernel<64, 4, 2, 3, PREC>), dim3(ntm), dim3(512), 0, st, p);
#endif
    else hipLaunchKernelGGL((conv16h_kernel<64, 8, 1, 3, PREC, 1, 16, 1>), dim3(p.M / 512), dim3(512), 0, st, p);
    return MMSEG_CHECK_LAUNCH();
}

// =====================================================================================================================================
// conv8h: Conv2D(Cout, 3, padding='same') of an 8-CHANNEL tensor in the 16-bit modes -- the shared convolution of every SPADE unit
// (layers/spade.py:27 of the reference: the 8-channel anatomy -> 128 hidden channels + ReLU) and the segmentor's first layer.
// Round 2 / 3 ran it as im2col (72 + 24 zero columns per pixel written as 16-bit rows) + a 1x1 product on conv_fast_kernel with three
// 32-deep K tiles: 0.06 of the MFMA peak, and bound by neither HBM nor the matrix pipe but by the per-block prologue / epilogue of a
// K = 96 product.  The layer is OUTPUT-WRITE bound (16 - 32 bytes in, 256 bytes out per pixel: ~ 70 FLOP/byte), so:
//   * no LDS, no barrier: on v_mfma_f32_32x32x16_* a lane's A operand is 8 consecutive k = the 8 channels of ONE tap of ONE pixel -- exactly
//     one 16-byte (16-bit input) or 32-byte (fp32 input, converted on the fly) row of the NHWC tensor; lane half h takes tap 2 s + h of
//     k-step s, so the nine taps are five k-steps (the tenth half-step multiplies zero weights).  Image borders = out-of-range buffer
//     offsets = zeros;
//   * the weights (9 x 8 x Cout, as 5 x TN fragments) live in REGISTERS for the whole launch; a wave computes 32 pixels x 32 TN output
//     channels per work item (20 MFMAs at TN = 4) and walks items wave by wave (grid-stride): nothing is shared between waves;
//   * epilogue: bias, activation; a 16-bit output tile (32 pixels x 128 channels = 8 KB, CONTIGUOUS in HBM when Cout = 128) goes through a
//     per-wave LDS slab (rows padded to 272 bytes) and leaves as 16-byte stores, 1 KB contiguous per instruction -- measured: the 8-byte quad
//     stores of the other kernels (64-byte runs) hold this store-bound layer at 1.05 TB/s.  The WEIGHTS are the MFMA's A operand (rows of
//     the product = output channels, in an order that makes a lane's 16 accumulator registers 16 consecutive channels of one pixel): no
//     lane transposition, two 16-byte LDS writes per 32 channels; fp32 output: 16-byte stores straight from the registers;
//   * the next item's operand rows are requested before the epilogue of the current one.
template <int PREC, bool X16>
__device__ __forceinline__ void conv8h_load(typename LowPrec<PREC>::V8 (&af)[5], const __amdgpu_buffer_rsrc_t rx, int pix0, int yy, bool live, int H, int W,
                                            int li, int lh) {
    // pix0: index of the segment's first pixel in the [B*H*W] pixel list, yy its image row
    typedef typename LowPrec<PREC>::V8 LV8;
    typedef typename LowPrec<PREC>::T LT;
    const int px = (pix0 % W) + li;                 // (W is a multiple of 32: cheap when a power of two, one 32-bit division otherwise)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + lh;                 // lh = 0: taps 0 2 4 6 8; lh = 1: taps 1 3 5 7 (9: none)
        const int kh = (s * 2) / 3 + (lh & ((s * 2) % 3 == 2)), kw = tap - kh * 3;
        const int ys = yy + kh - 1, xs = px + kw - 1;
        const bool ok = live && tap < 9 && (unsigned)ys < (unsigned)H && (unsigned)xs < (unsigned)W;
        const int pix = pix0 + li + (kh - 1) * W + (kw - 1);
        if constexpr (X16) {
            af[s] = __builtin_bit_cast(LV8, buf_load4(rx, ok ? pix * 16 : BUF_OOB));
        } else {
            const f32x4 lo = buf_load4(rx, ok ? pix * 32 : BUF_OOB), hi = buf_load4(rx, ok ? pix * 32 + 16 : BUF_OOB);
            af[s] = LV8{(LT)lo[0], (LT)lo[1], (LT)lo[2], (LT)lo[3], (LT)hi[0], (LT)hi[1], (LT)hi[2], (LT)hi[3]};
        }
    }
}

// ReLU, or the branch-free form of none (slope 1) / LeakyReLU (slope alpha)
template <bool RELU>
__device__ __forceinline__ float conv8h_act(float v, float slope) {
    return RELU ? fmaxf(v, 0.f) : fmaxf(v, 0.f) + slope * fminf(v, 0.f);
}

constexpr int C8H_ROW = 272;                      // bytes per pixel row of the epilogue slab (256 + 16: see above)

template <int PREC, int TN, bool X16, bool Y16, bool RELU>
__global__ __launch_bounds__(256, 2) void conv8h_kernel(const void* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          void* __restrict__ y, int B, int H, int W, int Cout, float slope) {
    typedef typename LowPrec<PREC>::V8 LV8;
    typedef typename LowPrec<PREC>::T LT;
    static_assert(TN == 4, "the epilogue slab holds 128 channels per pixel");
    __shared__ __attribute__((aligned(16))) unsigned char slab_all[Y16 ? 4 * 32 * C8H_ROW : 16];
    __shared__ __attribute__((aligned(16))) float bias_all[4][32 * TN];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ntn = (Cout + 32 * TN - 1) / (32 * TN);
    const int segs = W / 32;
    const int nt = (int)(((long)blockIdx.x * 4 + wid) % ntn);             // (a wave keeps ONE column group for the whole launch)
    // ---- weight fragments of this lane (the MFMA's A operand: ROWS of the product = output channels): row i of column tile j is channel
    //      16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3), so that the accumulator registers of a lane -- rows 8 g + 4 lh + e in register 4 g + e -- are
    //      16 CONSECUTIVE channels of one pixel; k = 8 lh + c of k-step s = channel c of tap 2 s + lh
    const int chan = 16 * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3);
    LV8 bf[5][TN];
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 72 * Cout * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tap = 2 * s + lh, n = (nt * TN + j) * 32 + chan;
                const int o = (tap < 9 && n < Cout) ? (tap * 8 * Cout + n) * 4 : BUF_OOB;      // (out of range -> 0: no branch per element)
                LV8 v;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = (LT)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, o, c * Cout * 4, 0));
                bf[s][j] = v;
            }
    }
    float* bias_s = bias_all[wid];
    for (int c = lane; c < 32 * TN; c += 64) bias_s[c] = (bias && nt * 32 * TN + c < Cout) ? bias[nt * 32 * TN + c] : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, B * H * W * 8 * (X16 ? 2 : 4), 0x00020000);
    // the (pixel row segment) items of this wave: its slot among the waves of the same column group, then a stride of their number
    const int G = (int)gridDim.x * 4;
    const int wslot = ((int)blockIdx.x * 4 + wid) / ntn, nslots = (G + ntn - 1 - nt) / ntn;
    const int nseg = B * H * segs;
    // image row of an item, advanced without a division per item: row index r = item / segs over all images, yy = r % H
    const int dr = nslots / segs, ds = nslots - dr * segs, dyy = dr % H;
    int seg = wslot % segs, yy = (wslot / segs) % H;
    LV8 af[5];
    conv8h_load<PREC, X16>(af, rx, wslot * 32, yy, wslot < nseg, H, W, li, lh);
    for (int it = wslot; it < nseg; it += nslots) {
        f32x16 acc[TN];                                                   // starts from the bias (the MFMA's C operand)
        int boff = 16 * lh;
        asm volatile("" : "+v"(boff));                                    // (re-read per item: hoisted out of the loop the table would occupy 64 registers)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + j * 32 + boff + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * q + e] = b4[e];
            }
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = LowPrec<PREC>::mfma(bf[s][j], af[s], acc[j]);
        seg += ds; yy += dyy;
        if (seg >= segs) { seg -= segs; ++yy; }
        if (yy >= H) yy -= H;
        conv8h_load<PREC, X16>(af, rx, (it + nslots) * 32, yy, it + nslots < nseg, H, W, li, lh);      // next item's rows: in flight during the epilogue
        // accumulator: column = pixel li of the segment, registers = channels 32 j + 16 lh + 0..15 of the column group
        const size_t mrow0 = (size_t)it * 32;                               // first pixel of the segment (items walk pixels in memory order)
        if constexpr (Y16) {
            unsigned char* slab = slab_all + wid * 32 * C8H_ROW;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                LV8 lo, hi;
#pragma unroll
                for (int r = 0; r < 8; ++r) { lo[r] = (LT)conv8h_act<RELU>(acc[j][r], slope); hi[r] = (LT)conv8h_act<RELU>(acc[j][8 + r], slope); }
                if constexpr (!RELU) __builtin_amdgcn_sched_barrier(0);   // (one column tile at a time: 64 pending min / max results would spill)
                *reinterpret_cast<LV8*>(slab + li * C8H_ROW + (j * 32 + 16 * lh) * 2) = lo;
                *reinterpret_cast<LV8*>(slab + li * C8H_ROW + (j * 32 + 16 * lh) * 2 + 16) = hi;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int ch = (lane & 15) * 8 + nt * 128;                      // this lane's 8 channels (16 bytes) of pixel (lane >> 4) + 4 i
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pr = (lane >> 4) + 4 * i;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + pr * C8H_ROW + (lane & 15) * 16);
                if (ch < Cout) *reinterpret_cast<f32x4*>(reinterpret_cast<LT*>(y) + (mrow0 + pr) * Cout + ch) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n0 = (nt * TN + j) * 32 + 16 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n0 + 4 * q < Cout)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (mrow0 + li) * Cout + n0 + 4 * q) =
                            f32x4{conv8h_act<RELU>(acc[j][4 * q], slope), conv8h_act<RELU>(acc[j][4 * q + 1], slope),
                                  conv8h_act<RELU>(acc[j][4 * q + 2], slope), conv8h_act<RELU>(acc[j][4 * q + 3], slope)};
            }
        }
    }
}
