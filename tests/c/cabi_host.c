/* A plain-C99 host of libirsde_hip.so: what a non-Python caller of the C ABI looks like (include/irsde_hip.h only, no
 * HIP or torch headers).  Host-only calls (no GPU needed): create the three engine kinds, walk the weight inventories,
 * exercise the error reporting.  Built and run by tests/test_cabi.py::test_plain_c_host with gcc -std=c99. */
#include <stdio.h>
#include <string.h>

#include "irsde_hip.h"

static int fail(const char* what) {
    fprintf(stderr, "FAIL %s: %s\n", what, irsde_last_error());
    return 1;
}

int main(void) {
    irsde_engine* e = NULL;
    irsde_config cfg;
    irsde_nafnet_config ncfg;
    irsde_latent_unet_config lcfg;
    int64_t shape[4];
    int nd = 0, n, i;
    size_t total = 0;

    if (irsde_version() != 107) return fail("version");

    memset(&cfg, 0, sizeof cfg);
    cfg.in_nc = 3; cfg.out_nc = 3; cfg.nf = 64; cfg.depth = 4;
    if (irsde_create(&cfg, &e) != IRSDE_OK) return fail("irsde_create");
    n = irsde_num_weights(e);
    for (i = 0; i < n; ++i) {
        size_t cnt = 1;
        int k;
        if (!irsde_weight_name(e, i) || irsde_weight_shape(e, i, shape, &nd) != IRSDE_OK) return fail("inventory");
        for (k = 0; k < nd; ++k) cnt *= (size_t)shape[k];
        total += cnt;
    }
    printf("unet %d tensors %zu parameters\n", n, total);
    if (n != 151 || total != 137147523u) return fail("ConditionalUNet inventory (SURVEY.md 8a: 151 tensors, 137 147 523 parameters)");
    if (irsde_finalize_weights(e) == IRSDE_OK) return fail("finalize without weights must fail");
    if (strlen(irsde_last_error()) == 0) return fail("error string");
    irsde_destroy(e);

    memset(&ncfg, 0, sizeof ncfg);
    ncfg.img_channel = 3; ncfg.width = 64; ncfg.middle_blk_num = 1; ncfg.n_enc = 4; ncfg.n_dec = 4;
    ncfg.enc_blk_nums[0] = 1; ncfg.enc_blk_nums[1] = 1; ncfg.enc_blk_nums[2] = 1; ncfg.enc_blk_nums[3] = 28;
    ncfg.dec_blk_nums[0] = 1; ncfg.dec_blk_nums[1] = 1; ncfg.dec_blk_nums[2] = 1; ncfg.dec_blk_nums[3] = 1;
    if (irsde_create_nafnet(&ncfg, &e) != IRSDE_OK) return fail("irsde_create_nafnet");
    printf("nafnet %d tensors\n", irsde_num_weights(e));
    if (irsde_num_weights(e) != 668) return fail("Refusion ConditionalNAFNet inventory");
    irsde_destroy(e);

    memset(&lcfg, 0, sizeof lcfg);
    lcfg.in_ch = 3; lcfg.out_ch = 3; lcfg.ch = 8; lcfg.n_mult = 4; lcfg.embed_dim = 8;
    lcfg.ch_mult[0] = 4; lcfg.ch_mult[1] = 8; lcfg.ch_mult[2] = 8; lcfg.ch_mult[3] = 16;
    if (irsde_create_latent_unet(&lcfg, &e) != IRSDE_OK) return fail("irsde_create_latent_unet");
    printf("latent unet %d tensors\n", irsde_num_weights(e));
    if (irsde_num_weights(e) != 69) return fail("latent UNet inventory");
    {
        int64_t lat[3], hid[27];
        int nh = 0;
        if (irsde_latent_shapes(e, 40, 52, lat, hid, &nh) != IRSDE_OK) return fail("irsde_latent_shapes");
        if (nh != 9 || lat[0] != 8 || lat[1] != 6 || lat[2] != 8) return fail("latent shapes of a 40x52 image (48x64 padded)");
    }
    irsde_destroy(e);

    cfg.nf = 48; /* not a multiple of 32 */
    if (irsde_create(&cfg, &e) == IRSDE_OK) return fail("bad config accepted");
    puts("c host ok");
    return 0;
}
