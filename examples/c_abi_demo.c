/* Plain-C client of the C-ABI (include/fwi.h): no Python, no C++, no HIP headers.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo \
 *       -Lfull_waveform_inversion_amd -lfwi_hip -Wl,-rpath,$PWD/full_waveform_inversion_amd -lm
 *   ./examples/c_abi_demo            # needs an MI355X; prints checksums of d, F^T r and dJ/dc
 *
 * One 40 x 36 x 44 shot: forward(model, src, rec) with the forward term kept, adjoint(residual) with the
 * imaging condition, gradient().  tests/test_c_client.py builds it, runs it on the GPU and compares the
 * printed checksums with the same calls made through the Python binding.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fwi.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != FWI_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fwi_last_error(ctx)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double checksum(const float *a, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += (double)a[i] * (double)((i % 7) + 1);
    return s;
}

int main(void) {
    enum { NZ = 40, NY = 36, NX = 44, NT = 60, NREC = 5 };
    fwi_ctx *ctx = NULL;
    fwi_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.ndim = 3;
    cfg.nz = NZ; cfg.ny = NY; cfg.nx = NX;
    cfg.order = 8;
    cfg.nt_max = NT;
    cfg.npml = 6;
    cfg.device = 0;
    cfg.dtype = FWI_F32;
    cfg.kernel = FWI_KERNEL_AUTO;
    cfg.h = 10.0;
    cfg.dt = 1.0e-3;
    cfg.sigma_max = 400.0;
    if (fwi_abi_version() != FWI_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch\n");
        return 1;
    }
    int rc = fwi_create(&cfg, &ctx);
    if (rc != FWI_OK) {
        fprintf(stderr, "fwi_create failed (%d): %s\n", rc, fwi_last_error(NULL));
        return 1;
    }
    const size_t npts = (size_t)NZ * NY * NX;
    float *c = malloc(npts * sizeof *c), *g = malloc(npts * sizeof *g);
    for (size_t i = 0; i < npts; ++i) c[i] = 2000.0f + 500.0f * (float)((i * 2654435761u) % 1000u) / 1000.0f;
    int32_t src[3] = {20, 18, 22};
    int32_t rec[NREC * 3];
    for (int r = 0; r < NREC; ++r) { rec[3 * r] = 8; rec[3 * r + 1] = 6 + 5 * r; rec[3 * r + 2] = 7 + 6 * r; }
    float wav[NT], seis[NT * NREC], res[NT * NREC], adj[NT];
    for (int n = 0; n < NT; ++n) {  /* Ricker, 25 Hz */
        const double a = M_PI * 25.0 * (n * cfg.dt - 0.04);
        wav[n] = (float)((1.0 - 2.0 * a * a) * exp(-a * a));
    }
    CHECK(fwi_set_model(ctx, c));
    CHECK(fwi_forward(ctx, NT, 1, src, wav, NREC, rec, 1, seis));
    for (int i = 0; i < NT * NREC; ++i) res[i] = 0.5f * seis[i];
    CHECK(fwi_adjoint(ctx, res, 1, adj));
    CHECK(fwi_gradient(ctx, FWI_WRT_VELOCITY, g));
    printf("kernel %s\n", fwi_kernel_name(ctx));
    printf("seis %.9e\nadj %.9e\ngrad %.9e\n", checksum(seis, NT * NREC), checksum(adj, NT), checksum(g, npts));
    fwi_destroy(ctx);
    free(c);
    free(g);
    return 0;
}
