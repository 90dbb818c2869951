// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the DAFNet/MMSDNet
// training step.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MMSEG_ACT_NONE 0
#define MMSEG_ACT_RELU 1
#define MMSEG_ACT_LEAKY 2
#define MMSEG_ACT_TANH 3

#define MMSEG_CHECK_LAUNCH() (int)hipGetLastError()

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case MMSEG_ACT_RELU: return v > 0.f ? v : 0.f;
        case MMSEG_ACT_LEAKY: return v >= 0.f ? v : v * alpha;
        case MMSEG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT y (all four are invertible in sign)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float alpha) {
    switch (act) {
        case MMSEG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case MMSEG_ACT_LEAKY: return y >= 0.f ? 1.f : alpha;
        case MMSEG_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red /* >= 17 floats of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (wid == 0) {
        float t = lane < nw ? red[lane] : 0.f;
        t = wave_sum(t);
        if (lane == 0) red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// XCD-aware block remap (guide T1, bijective form): consecutive logical ids share an XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// ---- tensors stored as fp32 or as a 16-bit type (element code H: 0 = fp32, 1 = bf16, 2 = fp16; reduced-precision storage) -------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

// four consecutive elements number 4*i4 .. 4*i4+3 of a tensor stored with element code H
template <int H> __device__ __forceinline__ f32x4 ld4(const void* p, long i4) {
    if constexpr (H == 0) return reinterpret_cast<const f32x4*>(p)[i4];
    else if constexpr (H == 1) {
        const bf16x4_t v = reinterpret_cast<const bf16x4_t*>(p)[i4];
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        const f16x4_t v = reinterpret_cast<const f16x4_t*>(p)[i4];
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}
template <int H> __device__ __forceinline__ void st4(void* p, long i4, f32x4 v) {
    if constexpr (H == 0) reinterpret_cast<f32x4*>(p)[i4] = v;
    else if constexpr (H == 1) reinterpret_cast<bf16x4_t*>(p)[i4] = bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    else reinterpret_cast<f16x4_t*>(p)[i4] = f16x4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}
template <int H> __device__ __forceinline__ float ld1(const void* p, long i) {
    if constexpr (H == 0) return reinterpret_cast<const float*>(p)[i];
    else if constexpr (H == 1) return (float)reinterpret_cast<const __bf16*>(p)[i];
    else return (float)reinterpret_cast<const _Float16*>(p)[i];
}

// Sum the per-block partials part[blk][2][C] of 64 channels with FIN_LPC lanes per channel (blockDim FIN_NT, lane = threadIdx.x >> 6,
// channel c = blockIdx.x * 64 + (threadIdx.x & 63)); results valid where lane == 0.  Fixed order.  Used by every BatchNorm "final"
// kernel (norm.hip and the typed-I/O variants of act16.hip: same order -> same bits).  (Four lanes per channel walked up to 128
// partials each with dependent loads: ~10.7 us per launch, 104 launches per training iteration; 16 lanes: 5.3 us.)
constexpr int FIN_LPC = 16, FIN_NT = 64 * FIN_LPC;
__device__ __forceinline__ void reduce_partials_64(const float* __restrict__ part, int nblk, int C, int c, int lane,
                                                   float& s1, float& s2, float* sm /* 2 * FIN_NT floats */) {
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int b = lane; b < nblk; b += FIN_LPC) { a1 += part[((size_t)b * 2) * C + c]; a2 += part[((size_t)b * 2 + 1) * C + c]; }
    }
    sm[threadIdx.x] = a1; sm[FIN_NT + threadIdx.x] = a2;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int l = 0; l < FIN_LPC; ++l) { s1 += sm[l * 64 + cl]; s2 += sm[FIN_NT + l * 64 + cl]; }
}
