/* Compiled as plain C by tests/test_native_abi.py: include/mmseg_hip.h must be a C header (no C++ types), and the entry
 * points that launch nothing (geometry / workspace queries, the precision switch) must be callable from C. */
#include <stdio.h>
#include "mmseg_hip.h"

int main(void) {
    int ok = 1;
    ok &= mmseg_conv2d_parity_taps(4, 2, 0) == 2 && mmseg_conv2d_parity_taps(4, 2, 1) == 2;
    ok &= mmseg_conv2d_parity_taps(3, 2, 0) == 2 && mmseg_conv2d_parity_taps(3, 2, 1) == 1;
    ok &= mmseg_conv2d_fast_path(64, 0, 64, 0) == 1 && mmseg_conv2d_fast_path(1, 0, 64, 0) == 0;
    ok &= mmseg_conv2d_fast_path(64, 64, 64, 0) == 1 && mmseg_conv2d_fast_path(64, 0, 64, 1) == 0;
    ok &= mmseg_conv2d_wgrad_workspace(8, 256, 256, 64, 64, 3, 3) >= (long)9 * 64 * 64;
    ok &= mmseg_colsum_workspace_floats(1000000L, 8) >= 65536;
    ok &= mmseg_spectral_workspace_floats(4096, 512) > 4096 + 512;
    ok &= mmseg_segpb_stats_floats(8) == 8 * 11 + 8 && mmseg_segpb_class_offset(8) == 88;
    ok &= mmseg_tps_workspace_floats(16) > 0 && mmseg_pairloss_workspace_floats(8) > 0;
    {
        int prev = mmseg_set_conv_precision(1);
        ok &= mmseg_get_conv_precision() == 1;
        ok &= mmseg_set_conv_precision(2) == 1 && mmseg_get_conv_precision() == 2;
        ok &= mmseg_set_conv_precision(7) == 2 && mmseg_get_conv_precision() == 0;      /* unknown modes fall back to fp32 */
        mmseg_set_conv_precision(prev);
    }
    /* invalid geometry is rejected without a launch (no GPU needed): hipErrorInvalidValue = 1 */
    ok &= mmseg_dense_fwd(0, 0, 0, 0, 0, 64, 10, 10, 0, 0.f, 0) != 0;
    ok &= mmseg_tps_warp_fwd(0, 0, 0, 0, 0, 1, 16, 16, 4, 0) != 0;
    printf(ok ? "C ABI OK\n" : "C ABI FAILED\n");
    return ok ? 0 : 1;
}
