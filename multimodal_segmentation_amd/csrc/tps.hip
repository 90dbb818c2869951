// Thin-plate-spline warp of the anatomy factor (layers/stn_spline.py:36-67 + layers/interpolate_spline.py
// + tf.contrib.resampler of the reference) for gfx950.
//
// Because the STN is built with inverse=False (model_components/anatomy_fuser.py:30) the spline is fitted on
// the FIXED 5x5 control grid, so the sample locations are an affine function of the predicted offsets:
//     loc_normalised[b, p, :] = grid[p, :] + sum_j Mb[p, j] * theta[b, j, :]
// with a constant Mb [H*W, 25] computed once on the host in fp64 (see layers/stn_spline.py of this package).
// The 28x28 solve per sample of the reference therefore disappears; the kernels below are a 25-term dot
// product + a 4-tap gather per pixel (HBM/latency bound).  Coordinates follow the reference: normalised
// (row, col) -> reversed -> scaled by (W-1, H-1) -> (x, y) in pixels; taps outside the image contribute 0.
#include "common.hpp"

#define TPS_NCP 25

__device__ __forceinline__ void tps_loc(const float* __restrict__ Mb, const float* th /* LDS [25][2] */, int p, int H, int W,
                                        float& x, float& y) {
    const int row = p / W, col = p - row * W;
    float lr = (float)row / (float)(H - 1), lc = (float)col / (float)(W - 1);
    const float* m = Mb + (size_t)p * TPS_NCP;
    float ar = 0.f, ac = 0.f;
#pragma unroll
    for (int j = 0; j < TPS_NCP; ++j) { const float mv = m[j]; ar += mv * th[2 * j]; ac += mv * th[2 * j + 1]; }
    lr += ar; lc += ac;
    x = lc * (float)(W - 1);
    y = lr * (float)(H - 1);
}

// grid (ceil(HW/256), B)
template <int C>
__global__ void tps_warp_fwd_kernel(const float* __restrict__ vol, const float* __restrict__ theta, const float* __restrict__ Mb,
                                    float* __restrict__ out, float* __restrict__ loc, int H, int W) {
    __shared__ float th[2 * TPS_NCP];
    const int b = blockIdx.y;
    if (threadIdx.x < 2 * TPS_NCP) th[threadIdx.x] = theta[(size_t)b * 2 * TPS_NCP + threadIdx.x];
    __syncthreads();
    const int HW = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float x, y;
    tps_loc(Mb, th, p, H, W, x, y);
    if (loc) { loc[((size_t)b * HW + p) * 2] = x; loc[((size_t)b * HW + p) * 2 + 1] = y; }
    const float fxf = floorf(x), fyf = floorf(y);
    const float ax = x - fxf, ay = y - fyf;
    // clamp before the int conversion so that wild offsets cannot overflow
    const int fx = (int)fminf(fmaxf(fxf, -2.f), (float)W), fy = (int)fminf(fmaxf(fyf, -2.f), (float)H);
    const float* vb = vol + (size_t)b * HW * C;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xi = fx + (t & 1), yi = fy + (t >> 1);
        const float wgt = ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
            const float* src = vb + ((size_t)yi * W + xi) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += wgt * src[c];
        }
    }
    float* o = out + ((size_t)b * HW + p) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = acc[c];
}

// backward: d_vol is a scatter (4 taps per output pixel).  The contributions are accumulated with 64-bit INTEGER atomics in
// fixed point (2^-36 resolution, range +-1.3e8): integer addition is associative, so the result does not depend on the order
// in which the atomics land and the whole training step stays bitwise reproducible.  d_loc[b,p,2] = (dL/dx, dL/dy) in pixels.
#define TPS_FX_SCALE 68719476736.0      /* 2^36 */
template <int C>
__global__ void tps_warp_bwd_kernel(const float* __restrict__ vol, const float* __restrict__ loc, const float* __restrict__ dout,
                                    unsigned long long* __restrict__ dvol, float* __restrict__ dloc, int H, int W) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float x = loc[((size_t)b * HW + p) * 2], y = loc[((size_t)b * HW + p) * 2 + 1];
    const float fxf = floorf(x), fyf = floorf(y);
    const float ax = x - fxf, ay = y - fyf;
    const int fx = (int)fminf(fmaxf(fxf, -2.f), (float)W), fy = (int)fminf(fmaxf(fyf, -2.f), (float)H);
    const float* vb = vol + (size_t)b * HW * C;
    unsigned long long* dvb = dvol ? dvol + (size_t)b * HW * C : nullptr;
    float g[C];
    const float* go = dout + ((size_t)b * HW + p) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = go[c];
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xi = fx + (t & 1), yi = fy + (t >> 1);
        if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
            const float wx = (t & 1) ? ax : 1.f - ax, wy = (t >> 1) ? ay : 1.f - ay;
            const float sx = (t & 1) ? 1.f : -1.f, sy = (t >> 1) ? 1.f : -1.f;
            const float* src = vb + ((size_t)yi * W + xi) * C;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dot += g[c] * src[c];
                if (dvb) atomicAdd(dvb + ((size_t)yi * W + xi) * C + c,
                                   (unsigned long long)__double2ll_rn((double)(g[c] * wx * wy) * TPS_FX_SCALE));
            }
            gx += dot * sx * wy;
            gy += dot * sy * wx;
        }
    }
    if (dloc) { dloc[((size_t)b * HW + p) * 2] = gx; dloc[((size_t)b * HW + p) * 2 + 1] = gy; }
}

// Tiled variant of the data-gradient scatter: a block owns a 16 x 16 tile of OUTPUT pixels and pre-accumulates their taps in an
// LDS window of 32 x 32 source pixels around the tile (the warp is smooth: displacements beyond 8 pixels are rare and fall
// back to global atomics), then flushes only the touched entries -- ~3x fewer 64-bit global atomics than one per tap.
// Same fixed-point integer accumulation, hence the same bits as the untiled kernel.
#define TPS_TILE 16
#define TPS_WIN 32
template <int C>
__global__ __launch_bounds__(256) void tps_warp_bwd_tiled_kernel(const float* __restrict__ vol, const float* __restrict__ loc,
                                                                 const float* __restrict__ dout, unsigned long long* __restrict__ dvol,
                                                                 float* __restrict__ dloc, int H, int W) {
    __shared__ unsigned long long win[TPS_WIN * TPS_WIN * C];        // 64 KB
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * TPS_TILE, tx0 = blockIdx.x * TPS_TILE;
    const int wy0 = ty0 - (TPS_WIN - TPS_TILE) / 2, wx0 = tx0 - (TPS_WIN - TPS_TILE) / 2;
    for (int i = threadIdx.x; i < TPS_WIN * TPS_WIN * C; i += 256) win[i] = 0ull;
    __syncthreads();
    const int HW = H * W;
    const int y = ty0 + (threadIdx.x >> 4), x0 = tx0 + (threadIdx.x & 15);
    unsigned long long* dvb = dvol + (size_t)b * HW * C;
    if (y < H && x0 < W) {
        const int p = y * W + x0;
        const float x = loc[((size_t)b * HW + p) * 2], yy = loc[((size_t)b * HW + p) * 2 + 1];
        const float fxf = floorf(x), fyf = floorf(yy);
        const float ax = x - fxf, ay = yy - fyf;
        const int fx = (int)fminf(fmaxf(fxf, -2.f), (float)W), fy = (int)fminf(fmaxf(fyf, -2.f), (float)H);
        const float* vb = vol + (size_t)b * HW * C;
        float g[C];
        const float* go = dout + ((size_t)b * HW + p) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = go[c];
        float gx = 0.f, gy = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xi = fx + (t & 1), yi = fy + (t >> 1);
            if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
                const float wx = (t & 1) ? ax : 1.f - ax, wy = (t >> 1) ? ay : 1.f - ay;
                const float sx = (t & 1) ? 1.f : -1.f, sy = (t >> 1) ? 1.f : -1.f;
                const float* src = vb + ((size_t)yi * W + xi) * C;
                const int ly = yi - wy0, lx = xi - wx0;
                const bool inwin = (unsigned)ly < (unsigned)TPS_WIN && (unsigned)lx < (unsigned)TPS_WIN;
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dot += g[c] * src[c];
                    const unsigned long long v = (unsigned long long)__double2ll_rn((double)(g[c] * wx * wy) * TPS_FX_SCALE);
                    if (inwin) atomicAdd(&win[(ly * TPS_WIN + lx) * C + c], v);
                    else atomicAdd(dvb + ((size_t)yi * W + xi) * C + c, v);
                }
                gx += dot * sx * wy;
                gy += dot * sy * wx;
            }
        }
        if (dloc) { dloc[((size_t)b * HW + p) * 2] = gx; dloc[((size_t)b * HW + p) * 2 + 1] = gy; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TPS_WIN * TPS_WIN * C; i += 256) {
        const unsigned long long v = win[i];
        if (v != 0ull) {
            const int c = i % C, q = i / C;
            const int yi = wy0 + q / TPS_WIN, xi = wx0 + q % TPS_WIN;       // in range by construction (only valid taps were added)
            atomicAdd(dvb + ((size_t)yi * W + xi) * C + c, v);
        }
    }
}

__global__ void tps_fx_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (float)((double)acc[i] * (1.0 / TPS_FX_SCALE));
}
__global__ void tps_zero_kernel(unsigned long long* __restrict__ acc, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc[i] = 0ull;
}

// part[chunk][b][j][2]: dtheta[b][j][0] (row offset) = sum_p Mb[p][j] * dloc_y * (H-1); [1] (col) uses dloc_x * (W-1).
// grid (chunks, batch groups of TPS_DT_GB): the block stages TPS_DT_PIX pixels of Mb and of its batches' dloc in LDS with coalesced
// loads; a thread owns up to two (b, j, k) combinations and walks the pixels in order (the strided global reads of the one-thread-
// per-combination form took 149 us for 13 MB).
#define TPS_DT_PIX 128
#define TPS_DT_GB 8
__global__ __launch_bounds__(256) void tps_dtheta_partial_kernel(const float* __restrict__ dloc, const float* __restrict__ Mb,
                                                                 float* __restrict__ part, int B, int H, int W, int per) {
    constexpr int U = (TPS_DT_GB * TPS_NCP * 2 + 255) / 256;
    __shared__ float mb_s[TPS_DT_PIX * TPS_NCP];
    __shared__ float dl_s[TPS_DT_GB][TPS_DT_PIX * 2];
    const int HW = H * W, tid = threadIdx.x;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    const int b0 = blockIdx.y * TPS_DT_GB, nb = min(TPS_DT_GB, B - b0);
    const int ncombo = B * TPS_NCP * 2, nloc = nb * TPS_NCP * 2;
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = 0.f;
    for (int s = p0; s < p1; s += TPS_DT_PIX) {
        const int n = min(TPS_DT_PIX, p1 - s);
        for (int i = tid; i < n * TPS_NCP; i += 256) mb_s[i] = Mb[(size_t)s * TPS_NCP + i];
        for (int bb = 0; bb < nb; ++bb)
            for (int i = tid; i < n * 2; i += 256) dl_s[bb][i] = dloc[((size_t)(b0 + bb) * HW + s) * 2 + i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = tid + u * 256;
            if (q < nloc) {
                const int k = q & 1, j = (q >> 1) % TPS_NCP, bb = (q >> 1) / TPS_NCP;
                const float* dl = dl_s[bb] + (k == 0 ? 1 : 0);
                float acc = a[u];
                for (int p = 0; p < n; ++p) acc += mb_s[p * TPS_NCP + j] * dl[p * 2];
                a[u] = acc;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = tid + u * 256;
        if (q < nloc) {
            const float sc = (q & 1) == 0 ? (float)(H - 1) : (float)(W - 1);
            part[(size_t)blockIdx.x * ncombo + b0 * TPS_NCP * 2 + q] = a[u] * sc;
        }
    }
}
// 16 lanes per combination walk the chunk partials, then a fixed-order sum over the lanes; block 256 = 16 combinations
__global__ __launch_bounds__(256) void tps_dtheta_final_kernel(const float* __restrict__ part, float* __restrict__ dtheta, int ncombo, int nchunk) {
    __shared__ float sm[16][17];
    const int ql = threadIdx.x & 15, ln = threadIdx.x >> 4;
    const int q = blockIdx.x * 16 + ql;
    float a = 0.f;
    if (q < ncombo) {
#pragma unroll 4
        for (int s = ln; s < nchunk; s += 16) a += part[(size_t)s * ncombo + q];
    }
    sm[ln][ql] = a;
    __syncthreads();
    if (ln == 0 && q < ncombo) {
        float v = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) v += sm[l][ql];
        dtheta[q] = v;
    }
}

#define TPS_CHUNKS 256

extern "C" {

int mmseg_tps_workspace_floats(int B) { return TPS_CHUNKS * B * TPS_NCP * 2; }

int mmseg_tps_warp_fwd(const float* vol, const float* theta, const float* Mb, float* out, float* loc, int B, int H, int W, int C,
                       void* stream) {
    if (C != 8 || H < 2 || W < 2) return (int)hipErrorInvalidValue;
    dim3 grid((H * W + 255) / 256, B);
    hipLaunchKernelGGL(tps_warp_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, vol, theta, Mb, out, loc, H, W);
    return MMSEG_CHECK_LAUNCH();
}
// floats of `acc` scratch for the deterministic scatter of d_vol: one 64-bit accumulator per element
long mmseg_tps_scatter_workspace_floats(int B, int H, int W, int C) { return 2L * B * H * W * C; }
// dvol or dtheta may be nullptr to skip that gradient; acc: mmseg_tps_scatter_workspace_floats floats, 8-byte aligned
// (needed only when dvol != nullptr; zeroed here)
int mmseg_tps_warp_bwd(const float* vol, const float* loc, const float* Mb, const float* dout, float* dvol, float* dtheta, float* dloc,
                       float* ws, float* acc, int B, int H, int W, int C, void* stream) {
    if (C != 8) return (int)hipErrorInvalidValue;
    if (dvol && (acc == nullptr || ((uintptr_t)acc & 7))) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((H * W + 255) / 256, B);
    const long n = (long)B * H * W * C;
    unsigned long long* acc64 = dvol ? reinterpret_cast<unsigned long long*>(acc) : nullptr;
    if (dvol) hipLaunchKernelGGL(tps_zero_kernel, dim3(2048), dim3(256), 0, st, acc64, n);
    if (dvol) {
        const dim3 tgrid((W + TPS_TILE - 1) / TPS_TILE, (H + TPS_TILE - 1) / TPS_TILE, B);
        hipLaunchKernelGGL(tps_warp_bwd_tiled_kernel<8>, tgrid, dim3(256), 0, st, vol, loc, dout, acc64, dtheta ? dloc : (float*)nullptr, H, W);
    } else {
        hipLaunchKernelGGL(tps_warp_bwd_kernel<8>, grid, dim3(256), 0, st, vol, loc, dout, acc64, dtheta ? dloc : (float*)nullptr, H, W);
    }
    if (dvol) hipLaunchKernelGGL(tps_fx_to_float_kernel, dim3(2048), dim3(256), 0, st, (const long long*)acc64, dvol, n);
    if (dtheta) {
        const int per = (H * W + TPS_CHUNKS - 1) / TPS_CHUNKS;
        const int ncombo = B * TPS_NCP * 2;
        hipLaunchKernelGGL(tps_dtheta_partial_kernel, dim3(TPS_CHUNKS, (B + TPS_DT_GB - 1) / TPS_DT_GB), dim3(256), 0, st, (const float*)dloc, Mb, ws, B, H, W, per);
        hipLaunchKernelGGL(tps_dtheta_final_kernel, dim3((ncombo + 15) / 16), dim3(256), 0, st, (const float*)ws, dtheta, ncombo, TPS_CHUNKS);
    }
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
