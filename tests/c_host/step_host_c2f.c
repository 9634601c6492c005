/* The torch-free C host of the coarse-to-fine path (BASELINE configs 4-5): the call sequence lua/adversarial_c2f_hip.lua makes for
 * one batch of adversarial_c2f.lua:40-187 through lua/facegen_hip.lua, in plain C99 against include/facegen_hip.h --
 *   models_c2f.lua:113-145 (create_G_d) and :237-278 (create_D_c) as fg_layer_spec lists (what FG.compile builds from the modules),
 *   FG.Gan(dnG, dnD, true, B): the step object in TABLE mode (G{noise, cond} through nn.JoinTable, D{x, cond} through nn.CAddTable),
 *   one D closure (real .diff / .coarse, new .coarse for the fake half, :126-143) and one G closure (:163-169), Adam, the c2f
 *   penalties of train_c2f.lua:27-34.
 * No torch, no Python, no other HIP user in the process.
 *
 *   step_host_c2f <dir> <B> <S>
 * reads  <dir>/{pG,pD,diff_real,cond_real,cond_fake,noise_d,cond_g,noise_g,pD_sync}.bin (host NCHW float32) and masks_{d,g}_<i>.bin
 *        (already in the device's element order), writes <dir>/out_*.npy for tests/test_gpu_c_host.py.
 */
#include "host_util.h"

static float* nhwc_from_file(const char* dir, const char* name, int n, int c, int s) {   /* FG.to_device_nhwc(FloatTensor) */
    const long long cnt = (long long)n * c * s * s;
    float* h = read_f32(dir, name, cnt);
    float* raw = dev_from_host(h, cnt);
    float* out = raw;
    if (c > 1) {
        out = dev_alloc(cnt);
        CHECK(fg_nchw_to_nhwc(ctx, raw, out, n, c, s, s));
        CHECK(fg_free(ctx, raw));
    }
    free(h);
    return out;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: step_host_c2f <dir> <batch> <size>\n"); return 1; }
    const char* dir = argv[1];
    const int B = atoi(argv[2]), S = atoi(argv[3]), H = B / 2, C = 3;
    printf("step_host_c2f: %s\n", fg_version());
    CHECK(fg_ctx_create(0, &ctx));                               /* FG.setDevice (train_c2f.lua:101-104, patched) */

    /* cudnn.SpatialConvolutionUpsample(nIn, nOut, k, k, 1): a 'same' convolution, factor 1 (fg_layer_spec.q) */
    const fg_layer_spec G_specs[] = {
        {FG_CONV, C + 1, 64, 3, 1, 1, 1}, {FG_PRELU, 0, 0, 0, 0, 0, 0}, {FG_CONV, 64, 64, 3, 1, 1, 1}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_CONV, 64, 128, 5, 2, 1, 1}, {FG_PRELU, 0, 0, 0, 0, 0, 0}, {FG_CONV, 128, 256, 5, 2, 1, 1}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_CONV, 256, C, 7, 3, 1, 1}, {FG_VIEW, C, S, S, 0, 0, 0}};
    const int nfeat = 256 * (S / 4) * (S / 4);
    const fg_layer_spec D_specs[] = {
        {FG_CONV, C, 64, 3, 1, 1, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0}, {FG_CONV, 64, 64, 3, 1, 1, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_MAXPOOL2, 0, 0, 0, 0, 0, 0},
        {FG_CONV, 64, 128, 3, 1, 1, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0}, {FG_CONV, 128, 256, 3, 1, 1, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_MAXPOOL2, 0, 0, 0, 0, 0, 0}, {FG_DROPOUT, 0, 0, 0, 0, 0.5f, 0},
        {FG_VIEW, nfeat, 0, 0, 0, 0, 0}, {FG_LINEAR, nfeat, 512, 0, 0, 0, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0}, {FG_DROPOUT, 0, 0, 0, 0, 0.5f, 0},
        {FG_LINEAR, 512, 1, 0, 0, 0, 0}, {FG_SIGMOID, 0, 0, 0, 0, 0, 0}};
    Net G = compile_from_file(G_specs, (int)(sizeof G_specs / sizeof G_specs[0]), C + 1, S, S, B, dir, "pG.bin");
    Net D = compile_from_file(D_specs, (int)(sizeof D_specs / sizeof D_specs[0]), C, S, S, B, dir, "pD.bin");
    const long long NPG = G.np, NPD = D.np;

    size_t gws_bytes = fg_gan_workspace_bytes(G.h, D.h, 1, B);
    void* gws = NULL;
    CHECK(fg_malloc(ctx, gws_bytes, &gws));
    fg_gan* gan = NULL;
    CHECK(fg_gan_create(ctx, G.h, D.h, 1, B, gws, gws_bytes, &gan));          /* FG.Gan(dnG, dnD, true, B) */
    CHECK(fg_gan_bind_workspaces(gan, G.ws, G.ws_bytes, D.ws, D.ws_bytes));
    CHECK(fg_gan_set_penalty(gan, 0, 1e-7f, 0.f, 1.f));                       /* D_L1, D_L2, D_clamp (train_c2f.lua:27-34) */
    CHECK(fg_gan_set_penalty(gan, 1, 0.f, 0.f, 5.f));
    CHECK(fg_gan_set_optimizer(gan, 0, 0, -1.0, 0.9, 0.999, 1e-8, 0, -1.0, 0, 0, 0));
    CHECK(fg_gan_set_optimizer(gan, 1, 0, -1.0, 0.9, 0.999, 1e-8, 0, -1.0, 0, 0, 0));

    float* diff_real = nhwc_from_file(dir, "diff_real.bin", H, C, S);
    float* cond_real = nhwc_from_file(dir, "cond_real.bin", H, C, S);
    float* cond_fake = nhwc_from_file(dir, "cond_fake.bin", H, C, S);
    float* noise_d = nhwc_from_file(dir, "noise_d.bin", H, 1, S);             /* [h][1][S][S] == [h][S][S][1] */
    float* cond_g = nhwc_from_file(dir, "cond_g.bin", B, C, S);
    float* noise_g = nhwc_from_file(dir, "noise_g.bin", B, 1, S);
    const int nm = fg_net_num_masks(D.h);
    if (nm != 2) { fprintf(stderr, "step_host_c2f: D has %d dropout layers, expected 2\n", nm); return 4; }
    const float* masks_d[2];
    const float* masks_g[2];
    for (int i = 0; i < nm; i++) {
        char name[64];
        const long long n = fg_net_mask_elems(D.h, i, B);
        snprintf(name, sizeof name, "masks_d_%d.bin", i);
        float* h = read_f32(dir, name, n);
        masks_d[i] = dev_from_host(h, n);
        free(h);
        snprintf(name, sizeof name, "masks_g_%d.bin", i);
        h = read_f32(dir, name, n);
        masks_g[i] = dev_from_host(h, n);
        free(h);
    }

    long long off, cnt;
    const float* gwsf = (const float*)gws;
    /* ---- D closure (adversarial_c2f.lua:123-160 + fevalD :40-88) ---- */
    CHECK(fg_step_D(gan, B, diff_real, cond_real, cond_fake, noise_d, masks_d, FG_STEP_NO_UPDATE));
    { long long s1[1] = {NPD}; dump(dir, "out_D_grad_raw.npy", D.grads, s1, 1); }
    CHECK(fg_gan_buffer(gan, FG_GAN_D_OUTPUT, &off, &cnt));
    { long long s1[1] = {B}; dump(dir, "out_D_prob.npy", (const float*)D.ws + off, s1, 1); }
    CHECK(fg_gan_buffer(gan, FG_GAN_LOSS, &off, &cnt));
    { long long s1[1] = {2}; dump(dir, "out_D_loss.npy", gwsf + off, s1, 1); }
    CHECK(fg_gan_buffer(gan, FG_GAN_CONFUSION, &off, &cnt));
    {
        int conf[8];
        CHECK(fg_d2h(ctx, conf, gwsf + off, sizeof conf));
        long long s1[1] = {8};
        write_npy(dir, "out_D_confusion.npy", "<i4", conf, 4, s1, 1);
    }
    CHECK(fg_gan_update(gan, 0));
    { long long s1[1] = {NPD}; dump(dir, "out_D_params.npy", D.params, s1, 1); }

    float* hsync = read_f32(dir, "pD_sync.bin", NPD);                          /* one common D for the G closure, like smoke() */
    CHECK(fg_h2d(ctx, D.params, hsync, (size_t)NPD * 4));
    CHECK(fg_net_params_changed(D.h));

    /* ---- G closure (adversarial_c2f.lua:163-187 + fevalG_on_D :92-113) ---- */
    CHECK(fg_step_G(gan, B, cond_g, noise_g, masks_g, FG_STEP_NO_UPDATE));
    { long long s1[1] = {NPG}; dump(dir, "out_G_grad_raw.npy", G.grads, s1, 1); }
    CHECK(fg_gan_buffer(gan, FG_GAN_D_INPUT, &off, &cnt));
    {
        float* nchw = dev_alloc((long long)B * C * S * S);
        CHECK(fg_nhwc_to_nchw(ctx, gwsf + off, nchw, B, C, S, S));
        long long s4[4] = {B, C, S, S};
        dump(dir, "out_G_samples.npy", nchw, s4, 4);
        CHECK(fg_free(ctx, nchw));
    }
    CHECK(fg_gan_buffer(gan, FG_GAN_D_OUTPUT, &off, &cnt));
    { long long s1[1] = {B}; dump(dir, "out_G_prob.npy", (const float*)D.ws + off, s1, 1); }
    CHECK(fg_gan_buffer(gan, FG_GAN_LOSS, &off, &cnt));
    { long long s1[1] = {2}; dump(dir, "out_G_loss.npy", gwsf + off, s1, 1); }
    CHECK(fg_gan_update(gan, 1));
    { long long s1[1] = {NPG}; dump(dir, "out_G_params.npy", G.params, s1, 1); }
    CHECK(fg_stream_sync(ctx));

    CHECK(fg_gan_destroy(gan));
    CHECK(fg_net_destroy(G.h));
    CHECK(fg_net_destroy(D.h));
    float* bufs[] = {G.params, G.grads, G.buffers, (float*)G.ws, D.params, D.grads, D.buffers, (float*)D.ws, (float*)gws,
                     diff_real, cond_real, cond_fake, noise_d, cond_g, noise_g};
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) CHECK(fg_free(ctx, bufs[i]));
    for (int i = 0; i < nm; i++) { CHECK(fg_free(ctx, (void*)masks_d[i])); CHECK(fg_free(ctx, (void*)masks_g[i])); }
    CHECK(fg_ctx_destroy(ctx));
    free(hsync);
    printf("step_host_c2f: OK\n");
    return 0;
}
