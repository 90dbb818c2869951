// Probe of ds_read_b64_tr_b16 (gfx950): hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
// Checks, element by element, that two transposed reads with the lane -> (row, columns) address rule of csrc/wgrad32h.hpp (wgrad16h) give every lane the 8 pixels
// of ITS channel that v_mfma_f32_32x32x16_bf16 expects as the A / B operand.  Printed "0 mismatches" on MI355X (round 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
// LDS image [pixel 0..15][channel 0..63] shorts (128 B rows), value = pixel * 100 + channel.
// lane l: group g = l >> 4, i = l & 15, q = i >> 2, p = i & 3; cbase = 16 * (g & 1), h = g >> 1; pass rd = 0 / 1 -> pixels 8 h + 4 rd + q
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int cbase = 16 * (g & 1), h = g >> 1;
    for (int rd = 0; rd < 2; ++rd) {
        const int pix = 8 * h + 4 * rd + q, ch = cbase + 4 * p;
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + pix * 64 + ch));
        for (int e = 0; e < 4; ++e) out[(l * 2 + rd) * 4 + e] = v[e];
    }
}
int main() {
    short* d; hipMalloc(&d, 64 * 8 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, hh = l >> 5;      // MFMA A lane: channel r (of 32), k half hh
        for (int j = 0; j < 8; ++j) {
            const int want = (8 * hh + j) * 100 + r;      // x[pixel 8 hh + j][channel r]
            const int got = h[(l * 2 + (j >> 2)) * 4 + (j & 3)];
            if (want != got) { if (bad < 10) printf("lane %d j %d want %d got %d\n", l, j, want, got); ++bad; }
        }
    }
    printf("tr16 layout check: %d mismatches\n", bad);
    return bad != 0;
}
