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
