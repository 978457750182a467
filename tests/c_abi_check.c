/* The public header must be usable from plain C (C99): compile check, a no-GPU smoke of the error paths, and -- when a GPU is
 * present and the program is given the asset directory and the committed fixture -- a 64x32 frame rendered through cloudsky.h
 * alone (no Python, no torch) and compared with tests/golden/clouds_64x32_deg45_rgba16f.bin.
 *   c_abi_check                         ABI / error paths only
 *   c_abi_check <asset_dir> <fixture>   + render and compare (needs a GPU: exit code 30 if there is none) */
#include "c_test_util.h"

static int render_and_compare(const char *asset_dir, const char *fixture) {
    csky_ctx *ctx = NULL;
    uint8_t *large, *small, *weather;
    uint16_t *ref = (uint16_t *)ctu_read_file(fixture, (size_t)64 * 32 * 8), *img = (uint16_t *)malloc((size_t)64 * 32 * 8);
    csky_cloud_params p;
    csky_sky_params sp;
    csky_transmittance_params tp;
    float pc[28];
    int bad;
    uint64_t inexact = 1;
    if (!ref || !img) return 20;
    if (ctu_default_noise(asset_dir, &large, &small, &weather) != 0) return 21;
    if (csky_create(&ctx, 0) != CSKY_OK) return 30;
    if (csky_set_noise(ctx, large, small, weather) != CSKY_OK) return 22;
    if (csky_noise_inexact_coeffs(ctx, &inexact) != CSKY_OK || inexact != 0) return 23;
    memset(&tp, 0, sizeof tp); tp.texture_size[0] = 256; tp.texture_size[1] = 64;                 /* transmittance_lut.gd:6 */
    if (csky_render_transmittance(ctx, &tp, NULL) != CSKY_OK) return 24;
    ctu_default_push_constant(pc, 64.0f, 32.0f);
    memcpy(&p, pc, sizeof p);
    memset(&sp, 0, sizeof sp); sp.texture_size[0] = 200; sp.texture_size[1] = 100;                /* sky_lut.gd:4 */
    sp.sun_direction[0] = p.LIGHT_DIRECTION[0]; sp.sun_direction[1] = p.LIGHT_DIRECTION[1]; sp.sun_direction[2] = p.LIGHT_DIRECTION[2];
    if (csky_render_sky_lut(ctx, &sp, NULL) != CSKY_OK) return 25;
    if (csky_render_clouds(ctx, &p, 64, 32, img, (size_t)64 * 8) != CSKY_OK) { fprintf(stderr, "%s\n", csky_last_error(ctx)); return 26; }
    bad = ctu_bad_pixels(img, ref, 64 * 32);
    printf("frame vs fixture: %d pixels beyond 2 fp16 ulp\n", bad);
    csky_destroy(ctx);
    return (bad >= 0 && bad <= 2) ? 0 : 27;                        /* the gate of tests/parity_metrics.py for a frame this small */
}

int main(int argc, char **argv) {
    csky_ctx *ctx = NULL;
    csky_cloud_params p;
    csky_bands b = {8, 0, 1, 1};
    memset(&p, 0, sizeof p);
    if (sizeof(csky_cloud_params) != 112 || sizeof(csky_sky_params) != 32 || sizeof(csky_transmittance_params) != 16) return 10;
    if (csky_abi_version() != CSKY_ABI_VERSION) return 11;
    if (csky_create(NULL, 0) != CSKY_ERR_INVALID) return 12;
    if (csky_render_clouds_device(NULL, &p, 8, &b, NULL, 64, NULL) != CSKY_ERR_INVALID) return 13;
    if (csky_multi_create(NULL, NULL, 0) != CSKY_ERR_INVALID || csky_multi_device_count(NULL) != 0) return 17;
    csky_destroy(NULL);
    csky_multi_destroy(NULL);
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
    if (argc >= 3) {
        const int rc = render_and_compare(argv[1], argv[2]);
        if (rc) return rc;
    }
    printf("c abi ok\n");
    return 0;
}
