// In-graph, per-sample loss terms of the automated-pairing trainers (reference models/dafnet.py:248-334) for gfx950:
//   pair dice      model_components/balancer.py:33-38   d_b = (2 sum a*b + 1e-12) / (sum a + sum b + 1e-12)
//   row mae        costs.py:24-26                       l_b = mean_{h,w} |x - y|
//   seg per-batch  costs.py:43-49,88-108,138-143        l_b = 1 - dice_b(first nm channels)
//                                                             + 0.01 * mean_pix sum_c w_c * p_c * (-log(softmax(t)_c + 1e-12))
//        (the reference calls weighted_cross_entropy_perbatch(y_true, y_pred) into a function declared (y_pred, y_true):
//         the class weights w_c = N / (n_c + 1e-12), n_c = sum_{batch,pix} p_c come from the PREDICTION and the softmax
//         + log is applied to the LABELS)
//   row dot        keras Multiply + Add                 o_b = sum_j w[b,j] * l[b,j]
// All are reductions over one sample followed by tiny per-sample arithmetic: HBM-bound, one read of the operands in the
// forward pass and one read + one write in the backward pass.  Reductions are two-level (fixed chunking, no atomics) so
// results are bit-reproducible.
#include "common.hpp"

#define PL_CHUNKS 128
#define PL_MAXC 8

// ---- pair dice ---------------------------------------------------------------------------------------------------------
// part[b][chunk][3] = sum a*b, sum a, sum b            grid (PL_CHUNKS, B)
__global__ void pair_dice_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                         long per_sample) {
    __shared__ float red[17];
    const int s = blockIdx.y;
    const long n4 = per_sample >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + (size_t)s * per_sample);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b + (size_t)s * per_sample);
    float sI = 0.f, sA = 0.f, sB = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 u = a4[i], v = b4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { sI += u[e] * v[e]; sA += u[e]; sB += v[e]; }
    }
    float* o = part + ((size_t)s * PL_CHUNKS + blockIdx.x) * 3;
    float r;
    r = block_sum(sI, red); if (threadIdx.x == 0) o[0] = r;
    r = block_sum(sA, red); if (threadIdx.x == 0) o[1] = r;
    r = block_sum(sB, red); if (threadIdx.x == 0) o[2] = r;
}
// stats[b][3]; out[b*ldo] = dice
__global__ void pair_dice_final_kernel(const float* __restrict__ part, float* __restrict__ stats, float* __restrict__ out, int ldo,
                                       int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    float s[3] = {0.f, 0.f, 0.f};
    for (int ch = 0; ch < PL_CHUNKS; ++ch)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += part[((size_t)t * PL_CHUNKS + ch) * 3 + k];
    stats[3 * t] = s[0]; stats[3 * t + 1] = s[1]; stats[3 * t + 2] = s[2];
    out[(size_t)t * ldo] = (2.f * s[0] + 1e-12f) / (s[1] + s[2] + 1e-12f);
}
// d dice / d a = 2 b / den - num / den^2 (and symmetrically for b).  da accumulates when acc_a != 0.
__global__ void pair_dice_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ stats,
                                     const float* __restrict__ g, int ldg, float* __restrict__ da, float* __restrict__ db,
                                     long per_sample, int acc_a) {
    const int s = blockIdx.y;
    const float den = stats[3 * s + 1] + stats[3 * s + 2] + 1e-12f, num = 2.f * stats[3 * s] + 1e-12f;
    const float gs = g[(size_t)s * ldg];
    const float k1 = gs * 2.f / den, k0 = -gs * num / (den * den);
    const long n4 = per_sample >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + (size_t)s * per_sample);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b + (size_t)s * per_sample);
    f32x4* da4 = reinterpret_cast<f32x4*>(da + (size_t)s * per_sample);
    f32x4* db4 = db ? reinterpret_cast<f32x4*>(db + (size_t)s * per_sample) : nullptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 u = a4[i], v = b4[i];
        f32x4 ra = acc_a ? da4[i] : (f32x4){0.f, 0.f, 0.f, 0.f}, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[e] += k1 * v[e] + k0; rb[e] = k1 * u[e] + k0; }
        da4[i] = ra;
        if (db4) db4[i] = rb;
    }
}

// ---- row mae -----------------------------------------------------------------------------------------------------------
__global__ void row_mae_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part,
                                       long per_sample) {
    __shared__ float red[17];
    const int s = blockIdx.y;
    const float* xs = x + (size_t)s * per_sample;
    const float* ys = y + (size_t)s * per_sample;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long)gridDim.x * blockDim.x)
        acc += fabsf(xs[i] - ys[i]);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[(size_t)s * PL_CHUNKS + blockIdx.x] = acc;
}
__global__ void row_mae_final_kernel(const float* __restrict__ part, float* __restrict__ out, float inv_n, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    float a = 0.f;
    for (int ch = 0; ch < PL_CHUNKS; ++ch) a += part[(size_t)t * PL_CHUNKS + ch];
    out[t] = a * inv_n;
}
// dy = g_b * sign(y - x) / n   (TF: sign(0) = 0)
__global__ void row_mae_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
                                   float* __restrict__ dy, long per_sample, float inv_n) {
    const int s = blockIdx.y;
    const float gs = g[s] * inv_n;
    const size_t base = (size_t)s * per_sample;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long)gridDim.x * blockDim.x) {
        const float d = y[base + i] - x[base + i];
        dy[base + i] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    }
}

// ---- per-sample segmentation loss ----------------------------------------------------------------------------------------
__device__ __forceinline__ void label_nll(const float* __restrict__ t, int C, float* q /* [PL_MAXC] */) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < PL_MAXC; ++c) if (c < C) mx = fmaxf(mx, t[c]);
    float e[PL_MAXC], sum = 0.f;
#pragma unroll
    for (int c = 0; c < PL_MAXC; ++c) if (c < C) { e[c] = expf(t[c] - mx); sum += e[c]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < PL_MAXC; ++c) if (c < C) q[c] = -logf(e[c] * inv + 1e-12f);
}
// part[b][chunk][3 + 2*PL_MAXC]: I, T, P over the first nm channels; n_bc = sum_pix p_c; R_bc = sum_pix p_c * q_c
__global__ void segpb_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ part,
                                     long HW, int C, int nm) {
    __shared__ float red[17];
    const int b = blockIdx.y;
    const long per = (HW + PL_CHUNKS - 1) / PL_CHUNKS;
    const long p0 = (long)blockIdx.x * per, p1 = min(HW, p0 + per);
    float sI = 0.f, sT = 0.f, sP = 0.f, n[PL_MAXC], R[PL_MAXC];
#pragma unroll
    for (int c = 0; c < PL_MAXC; ++c) { n[c] = 0.f; R[c] = 0.f; }
    for (long px = p0 + threadIdx.x; px < p1; px += blockDim.x) {
        const size_t o = ((size_t)b * HW + px) * C;
        float t[PL_MAXC], q[PL_MAXC];
#pragma unroll
        for (int c = 0; c < PL_MAXC; ++c) t[c] = c < C ? target[o + c] : 0.f;
        label_nll(t, C, q);
#pragma unroll
        for (int c = 0; c < PL_MAXC; ++c) {
            if (c < C) {
                const float p = pred[o + c];
                if (c < nm) { sI += t[c] * p; sT += t[c]; sP += p; }
                n[c] += p;
                R[c] += p * q[c];
            }
        }
    }
    float* o = part + ((size_t)b * PL_CHUNKS + blockIdx.x) * (3 + 2 * PL_MAXC);
    float v;
    v = block_sum(sI, red); if (threadIdx.x == 0) o[0] = v;
    v = block_sum(sT, red); if (threadIdx.x == 0) o[1] = v;
    v = block_sum(sP, red); if (threadIdx.x == 0) o[2] = v;
#pragma unroll
    for (int c = 0; c < PL_MAXC; ++c) {
        v = block_sum(n[c], red); if (threadIdx.x == 0) o[3 + c] = v;
        v = block_sum(R[c], red); if (threadIdx.x == 0) o[3 + PL_MAXC + c] = v;
    }
}
// stats: [B][3 + PL_MAXC] (I, T, P, R_b0..R_b7) then [PL_MAXC] batch class sums n_c (local batch; the caller may all-reduce)
__global__ void segpb_stats_final_kernel(const float* __restrict__ part, float* __restrict__ stats, int B) {
    const int W = 3 + 2 * PL_MAXC, SW = 3 + PL_MAXC;
    const int t = threadIdx.x;
    if (t < B * SW) {
        const int b = t / SW, k = t % SW;
        const int src = k < 3 ? k : k + PL_MAXC;
        float a = 0.f;
        for (int ch = 0; ch < PL_CHUNKS; ++ch) a += part[((size_t)b * PL_CHUNKS + ch) * W + src];
        stats[t] = a;
    }
    if (t < PL_MAXC) {
        float a = 0.f;
        for (int b = 0; b < B; ++b)
            for (int ch = 0; ch < PL_CHUNKS; ++ch) a += part[((size_t)b * PL_CHUNKS + ch) * W + 3 + t];
        stats[B * SW + t] = a;
    }
}
__global__ void segpb_loss_kernel(const float* __restrict__ stats, float* __restrict__ loss, int B, int C, float inv_hw,
                                  float lambda_bce) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int SW = 3 + PL_MAXC;
    const float* s = stats + (size_t)b * SW;
    const float* n = stats + (size_t)B * SW;
    float N = 0.f;
    for (int c = 0; c < C; ++c) N += n[c];
    float ce = 0.f;
    for (int c = 0; c < C; ++c) ce += N / (n[c] + 1e-12f) * s[3 + c];
    loss[b] = 1.f - (2.f * s[0] + 1e-12f) / (s[1] + s[2] + 1e-12f) + lambda_bce * ce * inv_hw;
}
// A_c = sum_b g_b * R_bc   (local batch; the caller may all-reduce before the gradient kernel)
__global__ void segpb_classgrad_kernel(const float* __restrict__ stats, const float* __restrict__ g, float* __restrict__ A, int B) {
    const int c = threadIdx.x;
    if (c >= PL_MAXC) return;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += g[b] * stats[(size_t)b * (3 + PL_MAXC) + 3 + c];
    A[c] = a;
}
// dpred = [c < nm] g_b (-2 t / den_b + num_b / den_b^2) + lambda / HW * (g_b q_c w_c + K_c),
//   K_c = sum_c' A_c' / (n_c' + e) - A_c N / (n_c + e)^2
__global__ void segpb_grad_kernel(const float* __restrict__ target, const float* __restrict__ stats, const float* __restrict__ g,
                                  const float* __restrict__ A, float* __restrict__ dpred, int B, long HW, int C, int nm,
                                  float inv_hw, float lambda_bce) {
    __shared__ float w[PL_MAXC], K[PL_MAXC];
    const int SW = 3 + PL_MAXC;
    if (threadIdx.x == 0) {
        const float* n = stats + (size_t)B * SW;
        float N = 0.f, sumq = 0.f;
        for (int c = 0; c < C; ++c) { N += n[c]; sumq += A[c] / (n[c] + 1e-12f); }
        for (int c = 0; c < C; ++c) {
            const float d = n[c] + 1e-12f;
            w[c] = N / d;
            K[c] = sumq - A[c] * N / (d * d);
        }
    }
    __syncthreads();
    const int b = blockIdx.y;
    const float* s = stats + (size_t)b * SW;
    const float den = s[1] + s[2] + 1e-12f, num = 2.f * s[0] + 1e-12f;
    const float gb = g[b];
    const float k1 = -2.f * gb / den, k0 = gb * num / (den * den);
    const float lam = lambda_bce * inv_hw;
    for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += (long)gridDim.x * blockDim.x) {
        const size_t o = ((size_t)b * HW + px) * C;
        float t[PL_MAXC], q[PL_MAXC];
#pragma unroll
        for (int c = 0; c < PL_MAXC; ++c) t[c] = c < C ? target[o + c] : 0.f;
        label_nll(t, C, q);
#pragma unroll
        for (int c = 0; c < PL_MAXC; ++c) {
            if (c < C) {
                float v = lam * (gb * q[c] * w[c] + K[c]);
                if (c < nm) v += k1 * t[c] + k0;
                dpred[o + c] = v;
            }
        }
    }
}

// ---- row dot -----------------------------------------------------------------------------------------------------------
__global__ void rowdot_fwd_kernel(const float* __restrict__ w, const float* __restrict__ l, float* __restrict__ out, int B, int J) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float a = 0.f;
    for (int j = 0; j < J; ++j) a += w[b * J + j] * l[b * J + j];
    out[b] = a;
}
__global__ void rowdot_bwd_kernel(const float* __restrict__ w, const float* __restrict__ l, const float* __restrict__ g,
                                  float* __restrict__ dw, float* __restrict__ dl, int B, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * J) return;
    const float gb = g[i / J];
    dw[i] = gb * l[i];
    dl[i] = gb * w[i];
}

extern "C" {

int mmseg_pairloss_workspace_floats(int B) { return B * PL_CHUNKS * (3 + 2 * PL_MAXC); }

// a, b [B, per_sample] (per_sample % 4 == 0); stats [B,3]; out[b * ldo] (ldo = row stride of the [B,J] dice matrix)
int mmseg_pair_dice_fwd(const float* a, const float* b, float* stats, float* out, int ldo, float* ws, int B, long per_sample,
                        void* stream) {
    if (B <= 0 || B > 65535 || (per_sample & 3) || ldo < 1) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_dice_partial_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, st, a, b, ws, per_sample);
    hipLaunchKernelGGL(pair_dice_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, stats, out, ldo, B);
    return MMSEG_CHECK_LAUNCH();
}
// da (+)= g_b * d dice/d a ; db = g_b * d dice/d b (db may be nullptr)
int mmseg_pair_dice_bwd(const float* a, const float* b, const float* stats, const float* g, int ldg, float* da, float* db,
                        int accumulate_a, int B, long per_sample, void* stream) {
    if (B <= 0 || B > 65535 || (per_sample & 3) || ldg < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_dice_bwd_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, a, b, stats, g, ldg, da, db,
                       per_sample, accumulate_a);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_row_mae_fwd(const float* x, const float* y, float* out, float* ws, int B, long per_sample, void* stream) {
    if (B <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(row_mae_partial_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, st, x, y, ws, per_sample);
    hipLaunchKernelGGL(row_mae_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, (const float*)ws, out, 1.f / (float)per_sample, B);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_row_mae_bwd(const float* x, const float* y, const float* g, float* dy, int B, long per_sample, void* stream) {
    if (B <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(row_mae_bwd_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, x, y, g, dy, per_sample,
                       1.f / (float)per_sample);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_segpb_stats_floats(int B) { return B * (3 + PL_MAXC) + PL_MAXC; }
int mmseg_segpb_class_offset(int B) { return B * (3 + PL_MAXC); }   // where the PL_MAXC batch class sums n_c start
int mmseg_segpb_stats(const float* pred, const float* target, float* stats, float* ws, int B, long HW, int C, int nm, void* stream) {
    if (C > PL_MAXC || nm > C || B <= 0 || B * (3 + PL_MAXC) > 1024) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(segpb_partial_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, st, pred, target, ws, HW, C, nm);
    hipLaunchKernelGGL(segpb_stats_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)ws, stats, B);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_segpb_loss(const float* stats, float* loss, int B, long HW, int C, float lambda_bce, void* stream) {
    hipLaunchKernelGGL(segpb_loss_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, loss, B, C,
                       1.f / (float)HW, lambda_bce);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_segpb_classgrad(const float* stats, const float* g, float* A, int B, void* stream) {
    hipLaunchKernelGGL(segpb_classgrad_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, g, A, B);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_segpb_grad(const float* target, const float* stats, const float* g, const float* A, float* dpred, int B, long HW, int C,
                     int nm, float lambda_bce, void* stream) {
    if (C > PL_MAXC || nm > C || B <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(segpb_grad_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, target, stats, g, A, dpred, B, HW,
                       C, nm, 1.f / (float)HW, lambda_bce);
    return MMSEG_CHECK_LAUNCH();
}

int mmseg_rowdot_fwd(const float* w, const float* l, float* out, int B, int J, void* stream) {
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, w, l, out, B, J);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_rowdot_bwd(const float* w, const float* l, const float* g, float* dw, float* dl, int B, int J, void* stream) {
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((B * J + 63) / 64), dim3(64), 0, (hipStream_t)stream, w, l, g, dw, dl, B, J);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
