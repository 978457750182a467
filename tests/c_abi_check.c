/* The public header must be usable from plain C (C99): compile-only check plus a no-GPU smoke of the error paths. */
#include <stdio.h>
#include <string.h>
#include "cloudsky.h"

int main(void) {
    csky_ctx *ctx = NULL;
    csky_cloud_params p;
    csky_bands b = {8, 0, 1, 1};
    memset(&p, 0, sizeof p);
    if (sizeof(csky_cloud_params) != 112 || sizeof(csky_sky_params) != 32 || sizeof(csky_transmittance_params) != 16) return 10;
    if (csky_abi_version() != CSKY_ABI_VERSION) return 11;
    if (csky_create(NULL, 0) != CSKY_ERR_INVALID) return 12;
    if (csky_render_clouds_device(NULL, &p, 8, &b, NULL, 64, NULL) != CSKY_ERR_INVALID) return 13;
    csky_destroy(NULL);
    {
        int rc = csky_create(&ctx, 0);
        if (rc == CSKY_OK) { printf("device present\n"); csky_destroy(ctx); }
        else if (rc == CSKY_ERR_NO_DEVICE && strstr(csky_last_error(NULL), "no CPU fallback")) printf("no device: %s\n", csky_last_error(NULL));
        else return 14;
    }
    {   /* host-only asset entry points */
        unsigned char vol[8 * 8 * 8 * 4];
        if (csky_generate_shape_noise(3u, 8, vol) != CSKY_OK) return 15;
        if (csky_mip_offset(8, 1, 4) != 8u * 8u * 8u * 4u) return 16;
    }
    printf("c abi ok\n");
    return 0;
}
