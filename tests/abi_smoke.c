/* Plain-C consumer of include/mvster_hip.h: proves the header is valid C and the library links from C
 * (the boundary is a C ABI, not a C++/torch one).  No GPU is touched: only version, error strings and the
 * host-side argument checks are exercised.  Built and run by tests/test_abi_cpu.py. */
#include <stdio.h>
#include <string.h>

#include "mvster_hip.h"

int main(void) {
    float dummy[16] = {0};
    const float *srcs[1] = {0};
    if (mvster_abi_version() != MVSTER_ABI_VERSION) return 1;
    if (strcmp(mvster_error_string(MVSTER_OK), "ok") != 0) return 2;
    if (mvster_stage_poses(0, 0, 1, 2, 0) != MVSTER_ERR_NULL_POINTER) return 3;
    if (mvster_stage_poses(dummy, dummy, 0, 2, 0) != MVSTER_ERR_BAD_SHAPE) return 4;
    if (mvster_cost_volume_fwd(dummy, srcs, 1, dummy, dummy, dummy, dummy, 1, 24, 8, 8, 8, 8, 8, 8, 2.0f,
                               MVSTER_LAYOUT_NHWC, 0) != MVSTER_ERR_UNSUPPORTED) return 5;
    if (mvster_depth_regress_fwd(dummy, dummy, dummy, dummy, 0, 0, dummy, dummy, 1, 2, 4, 4, 1.0f, 0) != MVSTER_ERR_BAD_SHAPE) return 6;
    if (mvster_sinkhorn_fwd(dummy, dummy, dummy, (const uint8_t *)dummy, dummy, 0, 1, 5, 4, 4, 3, 1.0f, 0, 0) != MVSTER_ERR_UNSUPPORTED) return 7;
    printf("mvster C ABI v%d ok\n", mvster_abi_version());
    return 0;
}
