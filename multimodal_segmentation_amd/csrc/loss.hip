// Loss reductions and their gradients for the DAFNet/MMSDNet trainers (gfx950, HBM-bound, deterministic
// two-stage reductions).  Restates costs.py of the reference:
//   * make_combined_dice_bce (costs.py:129-136) INCLUDING the swapped call bce(y_true, y_pred) into
//     weighted_cross_entropy_loss(y_pred, y_true) (costs.py:70-85): class weights are computed from the
//     PREDICTIONS over the whole batch and the log is taken of the LABELS;
//   * make_dice_loss_fnc / dice_coef_loss (costs.py:43-67);
//   * keras 'mae' / 'mse' and costs.ypred (mean of the output).
#include "common.hpp"

#define LOSS_CHUNKS 128
#define SEG_MAXC 8

// ---- segmentation loss statistics ---------------------------------------------------------------------------
// part[b][chunk][3 + 2*C]: I, T, P over the first nm channels; n_c = sum p_c ; S_c = sum p_c*log(t_c + 1e-12)
__global__ void segloss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ part,
                                       long HW, int C, int nm) {
    __shared__ float red[17];
    const int b = blockIdx.y;
    const long per = (HW + LOSS_CHUNKS - 1) / LOSS_CHUNKS;
    const long p0 = (long)blockIdx.x * per, p1 = min(HW, p0 + per);
    float sI = 0.f, sT = 0.f, sP = 0.f, n[SEG_MAXC], S[SEG_MAXC];
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c) { n[c] = 0.f; S[c] = 0.f; }
    for (long px = p0 + threadIdx.x; px < p1; px += blockDim.x) {
        const size_t o = ((size_t)b * HW + px) * C;
#pragma unroll
        for (int c = 0; c < SEG_MAXC; ++c) {
            if (c < C) {
                const float p = pred[o + c], t = target[o + c];
                if (c < nm) { sI += t * p; sT += t; sP += p; }
                n[c] += p;
                S[c] += p * logf(t + 1e-12f);
            }
        }
    }
    float* o = part + ((size_t)b * LOSS_CHUNKS + blockIdx.x) * (3 + 2 * SEG_MAXC);
    float v;
    v = block_sum(sI, red); if (threadIdx.x == 0) o[0] = v;
    v = block_sum(sT, red); if (threadIdx.x == 0) o[1] = v;
    v = block_sum(sP, red); if (threadIdx.x == 0) o[2] = v;
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c) {
        v = block_sum(n[c], red); if (threadIdx.x == 0) o[3 + c] = v;
        v = block_sum(S[c], red); if (threadIdx.x == 0) o[3 + SEG_MAXC + c] = v;
    }
}
// stats: [B][3] then [2][SEG_MAXC] class sums (summed over the local batch)
// block 1024 = 64 outputs x 16 lanes: a lane walks every 16th chunk partial, then a fixed-order sum over the lanes
// (one thread per output walked up to B * LOSS_CHUNKS partials with dependent loads: 31 us)
__global__ __launch_bounds__(1024) void segloss_stats_final_kernel(const float* __restrict__ part, float* __restrict__ stats, int B) {
    __shared__ float sm[16][65];
    const int W = 3 + 2 * SEG_MAXC;
    const int ln = threadIdx.x & 15;
    const int nout = B * 3 + 2 * SEG_MAXC;
    for (int base = 0; base < nout; base += 64) {
        const int t = base + (threadIdx.x >> 4);
        float a = 0.f;
        if (t < B * 3) {
            const int b = t / 3, k = t % 3;
#pragma unroll 8
            for (int ch = ln; ch < LOSS_CHUNKS; ch += 16) a += part[((size_t)b * LOSS_CHUNKS + ch) * W + k];
        } else if (t < nout) {
            const int k = t - B * 3;
#pragma unroll 8
            for (int q = ln; q < B * LOSS_CHUNKS; q += 16) a += part[(size_t)q * W + 3 + k];
        }
        sm[ln][threadIdx.x >> 4] = a;
        __syncthreads();
        if (ln == 0 && t < nout) {
            float v = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) v += sm[l][threadIdx.x >> 4];
            stats[t] = v;
        }
        __syncthreads();
    }
}
// loss value + gradient coefficients.  n_pix_global: pixel count of the WHOLE (all-rank) batch, normalises the loss value;
// n_pix_grad: pixel count the gradient coefficients are normalised with -- the LOCAL count under data parallelism, because the
// gradient all-reduce averages over ranks (mean of local-normalised seeds == gradient of the global-batch loss).
//  coef: [B][2] dice (a_b, b_b: d/dp = a_b * t + b_b), then [C] w_c, [C] k_c
__global__ void segloss_finalize_kernel(const float* __restrict__ stats, float* __restrict__ loss, float* __restrict__ coef,
                                        int B, int C, float n_pix_global, float n_pix_grad, float lambda_bce) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dice = 0.f;
    for (int b = 0; b < B; ++b) {
        const float I = stats[3 * b], U = stats[3 * b + 1] + stats[3 * b + 2];
        const float den = U + 1e-12f, num = 2.f * I + 1e-12f;
        dice += 1.f - num / den;
        coef[2 * b] = -2.f / den / (float)B;
        coef[2 * b + 1] = num / (den * den) / (float)B;
    }
    dice /= (float)B;
    float bce = 0.f;
    if (lambda_bce != 0.f) {
        const float* n = stats + 3 * B;
        const float* S = n + SEG_MAXC;
        float T = 0.f;
        for (int c = 0; c < C; ++c) T += n[c];
        float sumq = 0.f;
        for (int c = 0; c < C; ++c) sumq += S[c] / (n[c] + 1e-12f);
        for (int c = 0; c < C; ++c) {
            const float w = T / (n[c] + 1e-12f);
            bce -= w * S[c];
            coef[2 * B + c] = -lambda_bce / n_pix_grad * w;
            coef[2 * B + C + c] = -lambda_bce / n_pix_grad * (sumq - S[c] * T / ((n[c] + 1e-12f) * (n[c] + 1e-12f)));
        }
        bce /= n_pix_global;
    }
    loss[0] = dice + lambda_bce * bce;
}
__global__ void segloss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ coef,
                                    float* __restrict__ dpred, int B, long HW, int C, int nm, float scale, int use_bce) {
    const long n = (long)B * HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = i % C; const int b = i / (HW * C);
        const float t = target[i];
        float g = 0.f;
        if (c < nm) g += coef[2 * b] * t + coef[2 * b + 1];
        if (use_bce) g += coef[2 * B + c] * logf(t + 1e-12f) + coef[2 * B + C + c];
        dpred[i] = g * scale;
        (void)pred;
    }
}

// ---- mean |p - t|, mean (p - t)^2, mean p ----------------------------------------------------------------
// mode 0: mae, 1: mse, 2: mean(p).  target == nullptr -> constant tconst
__global__ void diffloss_partial_kernel(const float* __restrict__ p, const float* __restrict__ t, float tconst, long n, int mode,
                                        float* __restrict__ part) {
    __shared__ float red[17];
    float a = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = p[i] - (t ? t[i] : tconst);
        a += mode == 0 ? fabsf(d) : (mode == 1 ? d * d : p[i]);
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ void diffloss_final_kernel(const float* __restrict__ part, int nblk, float inv_n, float* __restrict__ loss) {
    __shared__ float red[17];
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += part[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) loss[0] = a * inv_n;
}
// keras/TF: d|d|/dd = sign(d) (0 at 0)
__global__ void diffloss_grad_kernel(const float* __restrict__ p, const float* __restrict__ t, float tconst, long n, int mode,
                                     float scale, float* __restrict__ dp) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = p[i] - (t ? t[i] : tconst);
        float g;
        if (mode == 0) g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        else if (mode == 1) g = 2.f * d;
        else g = 1.f;
        dp[i] = g * scale;
    }
}

static inline int lgrid(long n) {
    long b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" {

int mmseg_segloss_workspace_floats(int B) { return B * LOSS_CHUNKS * (3 + 2 * SEG_MAXC); }
int mmseg_segloss_stats_floats(int B) { return B * 3 + 2 * SEG_MAXC; }
int mmseg_segloss_coef_floats(int B, int C) { return 2 * B + 2 * C; }
int mmseg_segloss_class_offset(int B) { return 3 * B; }   // where the 2*SEG_MAXC batch-global class sums start

int mmseg_segloss_stats(const float* pred, const float* target, float* stats, float* ws, int B, long HW, int C, int nm, void* stream) {
    if (C > SEG_MAXC || nm > C || B * 3 > 1024) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(segloss_partial_kernel, dim3(LOSS_CHUNKS, B), dim3(256), 0, st, pred, target, ws, HW, C, nm);
    hipLaunchKernelGGL(segloss_stats_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)ws, stats, B);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_segloss_finalize(const float* stats, float* loss, float* coef, int B, int C, float n_pix_global, float n_pix_grad, float lambda_bce, void* stream) {
    hipLaunchKernelGGL(segloss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, loss, coef, B, C, n_pix_global, n_pix_grad, lambda_bce);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_segloss_grad(const float* pred, const float* target, const float* coef, float* dpred, int B, long HW, int C, int nm,
                       float scale, int use_bce, void* stream) {
    const long n = (long)B * HW * C;
    hipLaunchKernelGGL(segloss_grad_kernel, dim3(lgrid(n) * 4), dim3(256), 0, (hipStream_t)stream, pred, target, coef, dpred, B, HW, C, nm, scale, use_bce);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_diffloss_workspace_floats(void) { return 1024; }
int mmseg_diffloss(const float* p, const float* t, float tconst, long n, int mode, float* loss, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int nblk = lgrid(n);
    hipLaunchKernelGGL(diffloss_partial_kernel, dim3(nblk), dim3(256), 0, st, p, t, tconst, n, mode, ws);
    hipLaunchKernelGGL(diffloss_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nblk, 1.f / (float)n, loss);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_diffloss_grad(const float* p, const float* t, float tconst, long n, int mode, float scale, float* dp, void* stream) {
    hipLaunchKernelGGL(diffloss_grad_kernel, dim3(lgrid(n) * 4), dim3(256), 0, (hipStream_t)stream, p, t, tconst, n, mode, scale, dp);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
