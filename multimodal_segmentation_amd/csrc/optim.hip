// Optimiser and regulariser kernels (gfx950, HBM-bound).
//   * Keras 2.1.6 Adam:  m,v EMA;  p -= lr_t * m / (sqrt(v) + eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps = 1e-7
//     (call sites models/dafnet.py:93,114,155,161,349 of the reference) over one flat parameter arena:
//     28 B/param (read p,g,m,v; write p,m,v), one launch per trainer.
//   * Spectral regulariser (layers/spectralnorm.py:199-239): 3 power iterations from the constant u0, then
//     alpha*mean|stop_grad(W/sigma) - W| whose gradient is (alpha/N)*sign(W)*sign(1 - 1/sigma).
#include "common.hpp"

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n4, long n, float lr_host, const float* __restrict__ lr_dev, float b1, float b2, float eps) {
    // lr_dev != NULL: the bias-corrected step size is read from device memory, so that a captured launch (hipGraph replay of a
    // training step) sees the value of the step being replayed, not the one baked into the launch at capture time
    const float lr_t = lr_dev ? lr_dev[0] : lr_host;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pp = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
            pp[e] = pp[e] - lr_t * mm[e] / (sqrtf(vv[e]) + eps);
        }
        reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv; reinterpret_cast<f32x4*>(p)[i] = pp;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (long i = n4 * 4; i < n; ++i) {
            m[i] = b1 * m[i] + (1.f - b1) * g[i];
            v[i] = b2 * v[i] + (1.f - b2) * g[i] * g[i];
            p[i] = p[i] - lr_t * m[i] / (sqrtf(v[i]) + eps);
        }
    }
}

// ---- power iteration pieces ---------------------------------------------------------------------------
#define SPEC_KS 32
// part[ks][n] = sum_{k in slice} W[k][n] * u[k]
__global__ void spec_gemv_t_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ part, int K, int N, int kper) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int k0 = blockIdx.y * kper, k1 = min(K, k0 + kper);
    float a = 0.f;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) a += w[(size_t)k * N + n] * u[k];          // (unrolled: 8 independent row loads in flight, same summation order)
    part[(size_t)blockIdx.y * N + n] = a;
}
// out[k] = sum_n W[k][n] * v[n]  (one wave per row)
__global__ void spec_gemv_n_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ out, int K, int N) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    float a = 0.f;
    for (int n = lane; n < N; n += 64) a += w[(size_t)k * N + n] * v[n];
    a = wave_sum(a);
    if (lane == 0) out[k] = a;
}
// vec[i] = (sum_s part[s][i]) / || . ||   ; norm_out[0] = || . ||     (single block)
__global__ void spec_normalize_kernel(const float* __restrict__ part, int S, int n, float* __restrict__ vec, float* __restrict__ norm_out) {
    __shared__ float red[17];
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[(size_t)s * n + i];
        vec[i] = a;
        ss += a * a;
    }
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    for (int i = threadIdx.x; i < n; i += blockDim.x) vec[i] = vec[i] / nrm;
    if (threadIdx.x == 0 && norm_out) norm_out[0] = nrm;
}
// sum |W| partials
__global__ void spec_abs_partial_kernel(const float* __restrict__ w, long n, float* __restrict__ part) {
    __shared__ float red[17];
    float a = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += fabsf(w[i]);
    a = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
// loss = alpha * |1/sigma - 1| * mean|W| ; sgn[0] = alpha/N * sign(1 - 1/sigma)
__global__ void spec_final_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ sigma, long n, float alpha,
                                  float* __restrict__ loss, float* __restrict__ sgn) {
    __shared__ float red[17];
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += part[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) {
        const float d = 1.f - 1.f / sigma[0];
        loss[0] = alpha * fabsf(d) * a / (float)n;
        sgn[0] = (alpha / (float)n) * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
}
// dW = scale * sgn[0] * sign(W)
__global__ void spec_grad_kernel(const float* __restrict__ w, const float* __restrict__ sgn, float scale, long n, float* __restrict__ dw) {
    const float s = sgn[0] * scale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i];
        dw[i] = x > 0.f ? s : (x < 0.f ? -s : 0.f);
    }
}

// ---- the same, for up to 4 independent matrices per launch (the 4 down-sample blocks of one discriminator) --------------
// Each single-matrix step above is a tiny launch (the largest W is 4096 x 512); a discriminator's regulariser is 4 x 14 of
// them and an iteration evaluates 7 discriminators.  Batched: blockIdx.z (or .y) selects the matrix, grids cover the largest.
#define SPEC_MAXP 4
struct SpecBatch {
    const float* w[SPEC_MAXP];
    const float* u0[SPEC_MAXP];
    float* ws[SPEC_MAXP];          // per-matrix workspace: part | u | v | apart | sigma  (mmseg_spectral_workspace_floats)
    int K[SPEC_MAXP], N[SPEC_MAXP];
};
__device__ __forceinline__ long spec_part_floats_dev(int K, int N) { const long a = (long)SPEC_KS * N; return a > K ? a : K; }
__global__ void spec_gemv_t_multi_kernel(SpecBatch b, int first) {
    const int z = blockIdx.z, K = b.K[z], N = b.N[z];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* w = b.w[z];
    float* part = b.ws[z];
    const float* u = first ? b.u0[z] : part + spec_part_floats_dev(K, N);
    const int kper = (K + SPEC_KS - 1) / SPEC_KS;
    const int k0 = blockIdx.y * kper, k1 = min(K, k0 + kper);
    float a = 0.f;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) a += w[(size_t)k * N + n] * u[k];          // (unrolled: 8 independent row loads in flight, same summation order)
    part[(size_t)blockIdx.y * N + n] = a;
}
__global__ void spec_gemv_n_multi_kernel(SpecBatch b) {
    const int z = blockIdx.y, K = b.K[z], N = b.N[z];
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    const float* w = b.w[z];
    float* part = b.ws[z];
    const float* v = part + spec_part_floats_dev(K, N) + K;
    float a = 0.f;
    for (int n = lane; n < N; n += 64) a += w[(size_t)k * N + n] * v[n];
    a = wave_sum(a);
    if (lane == 0) part[k] = a;
}
// which = 0: v = normalise(sum_s part[s][:N]);  which = 1: u = normalise(part[:K]), sigma = the norm
__global__ void spec_normalize_multi_kernel(SpecBatch b, int which) {
    __shared__ float red[17];
    const int z = blockIdx.x, K = b.K[z], N = b.N[z];
    float* part = b.ws[z];
    float* u = part + spec_part_floats_dev(K, N);
    float* v = u + K;
    float* sigma = v + N + 1024;
    const int S = which ? 1 : SPEC_KS, n = which ? K : N;
    float* vec = which ? u : v;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[(size_t)s * n + i];
        vec[i] = a;
        ss += a * a;
    }
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    for (int i = threadIdx.x; i < n; i += blockDim.x) vec[i] = vec[i] / nrm;
    if (threadIdx.x == 0 && which) sigma[0] = nrm;
}
__global__ void spec_abs_partial_multi_kernel(SpecBatch b) {
    __shared__ float red[17];
    const int z = blockIdx.y, K = b.K[z], N = b.N[z];
    const long n = (long)K * N;
    const float* w = b.w[z];
    float* apart = b.ws[z] + spec_part_floats_dev(K, N) + K + N;
    float a = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += fabsf(w[i]);
    a = block_sum(a, red);
    if (threadIdx.x == 0) apart[blockIdx.x] = a;
}
__global__ void spec_final_multi_kernel(SpecBatch b, float alpha, float* __restrict__ loss, float* __restrict__ sgn) {
    __shared__ float red[17];
    const int z = blockIdx.x, K = b.K[z], N = b.N[z];
    const long n = (long)K * N;
    const float* apart = b.ws[z] + spec_part_floats_dev(K, N) + K + N;
    const float* sigma = apart + 1024;
    float a = 0.f;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) a += apart[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) {
        const float d = 1.f - 1.f / sigma[0];
        loss[z] = alpha * fabsf(d) * a / (float)n;
        sgn[z] = (alpha / (float)n) * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
}
struct SpecGradBatch { const float* w[SPEC_MAXP]; float* dw[SPEC_MAXP]; long n[SPEC_MAXP]; };
// dW[z] += scale * sgn[z] * sign(W[z])
__global__ void spec_grad_multi_kernel(SpecGradBatch b, const float* __restrict__ sgn, float scale) {
    const int z = blockIdx.y;
    const float s = sgn[z] * scale;
    const float* w = b.w[z];
    float* dw = b.dw[z];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < b.n[z]; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i];
        dw[i] += x > 0.f ? s : (x < 0.f ? -s : 0.f);
    }
}

extern "C" {

int mmseg_adam(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps, void* stream) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, lr_t, (const float*)nullptr, b1, b2, eps);
    return MMSEG_CHECK_LAUNCH();
}
// the same update with lr_t read from device memory (one float): the form a hipGraph capture of a training step records
int mmseg_adam_p(float* p, const float* g, float* m, float* v, long n, const float* lr_t, float b1, float b2, float eps, void* stream) {
    if (lr_t == nullptr) return (int)hipErrorInvalidValue;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, 0.f, lr_t, b1, b2, eps);
    return MMSEG_CHECK_LAUNCH();
}

// workspace floats: max(SPEC_KS*N, K) + K + N + 1024 + 2
static inline long spec_part_floats(int K, int N) { const long a = (long)SPEC_KS * N; return a > K ? a : K; }
long mmseg_spectral_workspace_floats(int K, int N) { return spec_part_floats(K, N) + K + N + 1024 + 2; }

// W: [K][N] (conv kernel reshaped (-1, Cout)), u0: [K].  Writes loss[0] and sgn[0] (see mmseg_spectral_grad).
int mmseg_spectral_fwd(const float* w, const float* u0, float* loss, float* sgn, float* ws, int K, int N, float alpha, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float* part = ws;                       // max(SPEC_KS * N, K)
    float* u = part + spec_part_floats(K, N);  // K
    float* v = u + K;                       // N
    float* apart = v + N;                   // 1024
    float* sigma = apart + 1024;            // 1
    const int kper = (K + SPEC_KS - 1) / SPEC_KS;
    const float* ucur = u0;
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(spec_gemv_t_kernel, dim3((N + 255) / 256, SPEC_KS), dim3(256), 0, st, w, ucur, part, K, N, kper);
        hipLaunchKernelGGL(spec_normalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)part, SPEC_KS, N, v, (float*)nullptr);
        hipLaunchKernelGGL(spec_gemv_n_kernel, dim3((K + 3) / 4), dim3(256), 0, st, w, (const float*)v, part, K, N);
        hipLaunchKernelGGL(spec_normalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)part, 1, K, u, sigma);
        ucur = u;
    }
    // sigma = u^T W v = ||W v|| of the last iteration
    const long n = (long)K * N;
    hipLaunchKernelGGL(spec_abs_partial_kernel, dim3(1024), dim3(256), 0, st, w, n, apart);
    hipLaunchKernelGGL(spec_final_kernel, dim3(1), dim3(256), 0, st, (const float*)apart, 1024, (const float*)sigma, n, alpha, loss, sgn);
    return MMSEG_CHECK_LAUNCH();
}
// The Spectral penalties of up to 4 matrices in one batch of launches (14 instead of 14 per matrix): loss[i], sgn[i] for
// matrix i.  ws = the matrices' workspaces back to back (mmseg_spectral_workspace_floats(K_i, N_i) floats each).
int mmseg_spectral_fwd4(const float* w0, const float* w1, const float* w2, const float* w3, const float* u0, const float* u1,
                        const float* u2, const float* u3, float* loss, float* sgn, float* ws, int n, int K0, int N0, int K1, int N1,
                        int K2, int N2, int K3, int N3, float alpha, void* stream) {
    if (n < 1 || n > SPEC_MAXP) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    SpecBatch b;
    const float* ws_[4] = {w0, w1, w2, w3};
    const float* us_[4] = {u0, u1, u2, u3};
    const int Ks[4] = {K0, K1, K2, K3}, Ns[4] = {N0, N1, N2, N3};
    int maxK = 0, maxN = 0;
    long off = 0;
    for (int i = 0; i < SPEC_MAXP; ++i) {
        const int j = i < n ? i : 0;
        b.w[i] = ws_[j]; b.u0[i] = us_[j]; b.K[i] = Ks[j]; b.N[i] = Ns[j];
        b.ws[i] = ws + (i < n ? off : 0);
        if (i < n) {
            if (Ks[i] < 1 || Ns[i] < 1) return (int)hipErrorInvalidValue;
            off += mmseg_spectral_workspace_floats(Ks[i], Ns[i]);
            if (Ks[i] > maxK) maxK = Ks[i];
            if (Ns[i] > maxN) maxN = Ns[i];
        }
    }
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(spec_gemv_t_multi_kernel, dim3((maxN + 255) / 256, SPEC_KS, n), dim3(256), 0, st, b, it == 0 ? 1 : 0);
        hipLaunchKernelGGL(spec_normalize_multi_kernel, dim3(n), dim3(1024), 0, st, b, 0);
        hipLaunchKernelGGL(spec_gemv_n_multi_kernel, dim3((maxK + 3) / 4, n), dim3(256), 0, st, b);
        hipLaunchKernelGGL(spec_normalize_multi_kernel, dim3(n), dim3(1024), 0, st, b, 1);
    }
    hipLaunchKernelGGL(spec_abs_partial_multi_kernel, dim3(1024, n), dim3(256), 0, st, b);
    hipLaunchKernelGGL(spec_final_multi_kernel, dim3(n), dim3(256), 0, st, b, alpha, loss, sgn);
    return MMSEG_CHECK_LAUNCH();
}
// dW_i += scale * sgn[i] * sign(W_i): the penalties' gradients accumulated straight into the gradient arena
int mmseg_spectral_grad4(const float* w0, const float* w1, const float* w2, const float* w3, const float* sgn, float* dw0, float* dw1,
                         float* dw2, float* dw3, int n, long n0, long n1, long n2, long n3, float scale, void* stream) {
    if (n < 1 || n > SPEC_MAXP) return (int)hipErrorInvalidValue;
    SpecGradBatch b;
    const float* ws_[4] = {w0, w1, w2, w3};
    float* ds_[4] = {dw0, dw1, dw2, dw3};
    const long ns[4] = {n0, n1, n2, n3};
    long mx = 0;
    for (int i = 0; i < SPEC_MAXP; ++i) {
        const int j = i < n ? i : 0;
        b.w[i] = ws_[j]; b.dw[i] = ds_[j]; b.n[i] = i < n ? ns[i] : 0;
        if (b.n[i] > mx) mx = b.n[i];
    }
    long blocks = (mx + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(spec_grad_multi_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, b, sgn, scale);
    return MMSEG_CHECK_LAUNCH();
}
int mmseg_spectral_grad(const float* w, const float* sgn, float scale, long n, float* dw, void* stream) {
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(spec_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, sgn, scale, n, dw);
    return MMSEG_CHECK_LAUNCH();
}

}  // extern "C"
