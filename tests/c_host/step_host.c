/* A torch-free, Python-free host of libfacegen_hip.so: the call sequence a LuaJIT host makes through lua/facegen_hip.lua
 * (train.lua:71-80 device + seed, :134-152 models + getParameters, nn_utils.lua:355-362 the Float <-> device copies,
 * adversarial.lua:240-288 one D closure + one G closure), written in plain C against include/facegen_hip.h.
 *
 * This process has NO other HIP user: device memory comes from fg_malloc, transfers are fg_h2d / fg_d2h, the context runs
 * on the default stream -- exactly the configuration LuaJIT gives the library (every GPU test in tests/ otherwise runs the
 * library inside a torch process on torch's allocator and stream).
 *
 *   step_host <dir> <B>
 * reads  <dir>/{pG,pD,real,noise_d,noise_g,pD_sync}.bin and masks_{d,g}_<i>.bin (raw little-endian float32, written by
 *        tests/test_gpu_c_host.py from the oracle's state), runs fg_step_D, fg_gan_update(D), [re-sync of D's parameters],
 *        fg_step_G, fg_gan_update(G) at batch B on the 32x32x3 nets of models.lua:57-81 / 382-416,
 * writes <dir>/out_*.npy; the pytest compares them with the oracle at the bars of __graft_entry__.smoke().
 */
#include "host_util.h"

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: step_host <dir> <batch>\n"); return 1; }
    const char* dir = argv[1];
    const int B = atoi(argv[2]), H = B / 2, C = 3, S = 32, ND = 100;
    printf("step_host: %s\n", fg_version());
    CHECK(fg_ctx_create(0, &ctx));                               /* cutorch.setDevice (train.lua:79) */

    /* MODELS.create_G_decoder_upsampling32 (models.lua:57-81) and create_D32b (models.lua:382-416) as fg_layer_spec lists */
    const fg_layer_spec G_specs[] = {
        {FG_LINEAR, ND, 128 * 8 * 8, 0, 0, 0, 0}, {FG_VIEW, 128, 8, 8, 0, 0, 0}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_UPSAMPLE2X, 0, 0, 0, 0, 0, 0}, {FG_CONV, 128, 256, 5, 2, 1, 0}, {FG_BATCHNORM, 256, 0, 0, 0, 1e-5f, 0.1f}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_UPSAMPLE2X, 0, 0, 0, 0, 0, 0}, {FG_CONV, 256, 128, 5, 2, 1, 0}, {FG_BATCHNORM, 128, 0, 0, 0, 1e-5f, 0.1f}, {FG_PRELU, 0, 0, 0, 0, 0, 0},
        {FG_CONV, 128, C, 3, 1, 1, 0}, {FG_SIGMOID, 0, 0, 0, 0, 0, 0}};
    fg_layer_spec D_specs[32];
    int nd = 0;
    const int chans[5] = {C, 64, 128, 256, 512};
    for (int i = 0; i < 4; i++) {
        D_specs[nd++] = (fg_layer_spec){FG_CONV, chans[i], chans[i + 1], 3, 1, 1, 0};
        D_specs[nd++] = (fg_layer_spec){FG_PRELU, 0, 0, 0, 0, 0, 0};
        D_specs[nd++] = (fg_layer_spec){FG_SPATIAL_DROPOUT, 0, 0, 0, 0, 0.2f, 0};
        D_specs[nd++] = (fg_layer_spec){FG_AVGPOOL2, 0, 0, 0, 0, 0, 0};
    }
    D_specs[nd++] = (fg_layer_spec){FG_VIEW, 2048, 0, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_LINEAR, 2048, 512, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_PRELU, 0, 0, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_DROPOUT, 0, 0, 0, 0, 0.5f, 0};
    D_specs[nd++] = (fg_layer_spec){FG_LINEAR, 512, 512, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_PRELU, 0, 0, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_DROPOUT, 0, 0, 0, 0, 0.5f, 0};
    D_specs[nd++] = (fg_layer_spec){FG_LINEAR, 512, 1, 0, 0, 0, 0};
    D_specs[nd++] = (fg_layer_spec){FG_SIGMOID, 0, 0, 0, 0, 0, 0};

    const long long NPG = 2470406, NPD = 2863239;                 /* getParameters() lengths (train.lua:151-152) */
    float* hpG = read_f32(dir, "pG.bin", NPG);
    float* hpD = read_f32(dir, "pD.bin", NPD);
    Net G = compile(G_specs, (int)(sizeof G_specs / sizeof G_specs[0]), ND, 1, 1, B, hpG, NPG);
    Net D = compile(D_specs, nd, C, S, S, B, hpD, NPD);

    size_t gws_bytes = fg_gan_workspace_bytes(G.h, D.h, 0, B);
    void* gws = NULL;
    CHECK(fg_malloc(ctx, gws_bytes, &gws));
    fg_gan* gan = NULL;
    CHECK(fg_gan_create(ctx, G.h, D.h, 0, B, gws, gws_bytes, &gan));
    CHECK(fg_gan_bind_workspaces(gan, G.ws, G.ws_bytes, D.ws, D.ws_bytes));
    CHECK(fg_gan_set_penalty(gan, 0, 0.f, 1e-4f, 1.f));           /* D_L1, D_L2, D_clamp (train.lua:29-37) */
    CHECK(fg_gan_set_penalty(gan, 1, 0.f, 0.f, 5.f));             /* G_L1, G_L2, G_clamp */
    CHECK(fg_gan_set_optimizer(gan, 0, 0, -1.0, 0.9, 0.999, 1e-8, 0, -1.0, 0, 0, 0));
    CHECK(fg_gan_set_optimizer(gan, 1, 0, -1.0, 0.9, 0.999, 1e-8, 0, -1.0, 0, 0, 0));

    /* inputs: host NCHW FloatTensors -> device NHWC (the nn.Copy of nn_utils.lua:355-362) */
    float* h_real = read_f32(dir, "real.bin", (long long)H * C * S * S);
    float* raw = dev_from_host(h_real, (long long)H * C * S * S);
    float* real = dev_alloc((long long)H * C * S * S);
    CHECK(fg_nchw_to_nhwc(ctx, raw, real, H, C, S, S));
    float* h_nd = read_f32(dir, "noise_d.bin", (long long)H * ND);
    float* h_ng = read_f32(dir, "noise_g.bin", (long long)B * ND);
    float* noise_d = dev_from_host(h_nd, (long long)H * ND);
    float* noise_g = dev_from_host(h_ng, (long long)B * ND);
    const int nm = fg_net_num_masks(D.h);
    if (nm != 6) { fprintf(stderr, "step_host: D has %d dropout layers, expected 6\n", nm); return 4; }
    const float* masks_d[6];
    const float* masks_g[6];
    for (int i = 0; i < nm; i++) {
        char name[64];
        long long n = fg_net_mask_elems(D.h, i, B);
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
    /* ---- D closure (adversarial.lua:240-268 + fevalD): gradients first (FG_STEP_NO_UPDATE), then the optimizer ---- */
    CHECK(fg_step_D(gan, B, real, NULL, NULL, noise_d, masks_d, FG_STEP_NO_UPDATE));
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
    if (fg_gan_optimizer_steps(gan, 0) != 1) { fprintf(stderr, "step_host: Adam's t for D is not 1\n"); return 5; }

    /* the G closure starts from ONE common D (the oracle's parameters after its own Adam step), like smoke() */
    float* hsync = read_f32(dir, "pD_sync.bin", NPD);
    CHECK(fg_h2d(ctx, D.params, hsync, (size_t)NPD * 4));
    CHECK(fg_net_params_changed(D.h));

    /* ---- G closure (adversarial.lua:275-288 + fevalG_on_D) ---- */
    CHECK(fg_step_G(gan, B, NULL, noise_g, masks_g, FG_STEP_NO_UPDATE));
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
                     raw, real, noise_d, noise_g};
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; i++) CHECK(fg_free(ctx, bufs[i]));
    for (int i = 0; i < nm; i++) { CHECK(fg_free(ctx, (void*)masks_d[i])); CHECK(fg_free(ctx, (void*)masks_g[i])); }
    CHECK(fg_ctx_destroy(ctx));
    free(hpG); free(hpD); free(h_real); free(h_nd); free(h_ng); free(hsync);
    printf("step_host: OK\n");
    return 0;
}
