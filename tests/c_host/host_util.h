/* Shared helpers of the torch-free C hosts (tests/c_host/step_host.c: the 32x32 nets of train.lua; step_host_c2f.c: the
 * coarse-to-fine nets of train_c2f.lua): raw float32 files in, NumPy .npy files out, device memory from fg_malloc, transfers through
 * fg_h2d / fg_d2h -- what lua/facegen_hip.lua's DeviceTensor / compile do, in plain C99 against include/facegen_hip.h. */
#ifndef FG_C_HOST_UTIL_H
#define FG_C_HOST_UTIL_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "facegen_hip.h"

static fg_ctx* ctx = NULL;

#define CHECK(call) do { int rc_ = (call); if (rc_ != FG_OK) { \
    fprintf(stderr, "c_host: %s -> %d: %s\n", #call, rc_, ctx ? fg_last_error(ctx) : "(no context)"); exit(2); } } while (0)

static float* read_f32(const char* dir, const char* name, long long expect) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "c_host: cannot open %s\n", path); exit(3); }
    float* p = (float*)malloc((size_t)expect * 4);
    if ((long long)fread(p, 4, (size_t)expect, f) != expect || fgetc(f) != EOF) {
        fprintf(stderr, "c_host: %s does not hold exactly %lld floats\n", path, expect); exit(3);
    }
    fclose(f);
    return p;
}

/* NumPy .npy v1.0, C order */
static void write_npy(const char* dir, const char* name, const char* descr, const void* data, size_t elem, const long long* shape, int nd) {
    char path[1024], dict[256], dims[128] = "";
    snprintf(path, sizeof path, "%s/%s", dir, name);
    size_t count = 1;
    for (int i = 0; i < nd; i++) {
        char one[32];
        snprintf(one, sizeof one, "%lld,", shape[i]);
        strcat(dims, one);
        count *= (size_t)shape[i];
    }
    int n = snprintf(dict, sizeof dict, "{'descr': '%s', 'fortran_order': False, 'shape': (%s), }", descr, dims);
    int total = 10 + n + 1;
    int pad = (64 - total % 64) % 64;
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "c_host: cannot write %s\n", path); exit(3); }
    unsigned short hlen = (unsigned short)(n + pad + 1);
    fwrite("\x93NUMPY\x01\x00", 1, 8, f);
    fwrite(&hlen, 2, 1, f);
    fwrite(dict, 1, (size_t)n, f);
    for (int i = 0; i < pad; i++) fputc(' ', f);
    fputc('\n', f);
    fwrite(data, elem, count, f);
    fclose(f);
}

static float* dev_alloc(long long n) {
    void* p = NULL;
    CHECK(fg_malloc(ctx, (size_t)n * 4, &p));
    return (float*)p;
}

static float* dev_from_host(const float* h, long long n) {       /* DeviceTensor(n):copy(FloatTensor) of lua/facegen_hip.lua */
    float* d = dev_alloc(n);
    CHECK(fg_h2d(ctx, d, h, (size_t)n * 4));
    return d;
}

static void dump(const char* dir, const char* name, const float* dev, const long long* shape, int nd) {
    long long n = 1;
    for (int i = 0; i < nd; i++) n *= shape[i];
    float* h = (float*)malloc((size_t)n * 4);
    CHECK(fg_d2h(ctx, h, dev, (size_t)n * 4));
    write_npy(dir, name, "<f4", h, 4, shape, nd);
    free(h);
}

typedef struct { fg_net* h; long long np, nb; float *params, *grads, *buffers; void* ws; size_t ws_bytes; } Net;

static Net compile(const fg_layer_spec* specs, int n, int c, int h, int w, int max_batch, const float* host_params, long long expect) {
    Net net;
    memset(&net, 0, sizeof net);
    CHECK(fg_net_create(ctx, specs, n, c, h, w, &net.h));
    net.np = fg_net_num_params(net.h);
    net.nb = fg_net_num_buffers(net.h);
    if (net.np != expect) { fprintf(stderr, "c_host: the plan has %lld parameters, the host vector %lld\n", net.np, expect); exit(4); }
    net.params = dev_from_host(host_params, net.np);
    net.grads = dev_alloc(net.np);
    CHECK(fg_fill(ctx, net.grads, 0.f, net.np));
    long long nb = net.nb > 0 ? net.nb : 1;
    float* hb = (float*)calloc((size_t)nb, 4);
    /* [running_mean | running_var] per BatchNorm layer in module order: fresh modules hold 0 | 1 */
    long long off = 0;
    for (int i = 0; i < n; i++)
        if (specs[i].type == FG_BATCHNORM) {
            for (int k = 0; k < specs[i].a; k++) hb[off + specs[i].a + k] = 1.f;
            off += 2 * specs[i].a;
        }
    net.buffers = dev_from_host(hb, nb);
    free(hb);
    net.ws_bytes = fg_net_workspace_bytes(net.h, max_batch);
    CHECK(fg_malloc(ctx, net.ws_bytes, &net.ws));
    CHECK(fg_net_bind(net.h, net.params, net.grads, net.buffers));
    return net;
}


/* the same with the parameter vector read from <dir>/<file> AFTER the plan says how long it is */
static __attribute__((unused)) Net compile_from_file(const fg_layer_spec* specs, int n, int c, int h, int w, int max_batch, const char* dir, const char* file) {
    fg_net* probe = NULL;
    CHECK(fg_net_create(ctx, specs, n, c, h, w, &probe));
    const long long np = fg_net_num_params(probe);
    CHECK(fg_net_destroy(probe));
    float* hp = read_f32(dir, file, np);
    Net net = compile(specs, n, c, h, w, max_batch, hp, np);
    free(hp);
    return net;
}
#endif
