/* C restatement of oracle/fwi_oracle.py (fp64, OpenMP) -- TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference (/root/reference/full_waveform_inversion.py)
 * has no wave-propagation, adjoint or gradient code, so
 * there is no reference file:line to follow.  This file restates the
 * BUILD-DEFINED scheme documented at the top of oracle/fwi_oracle.py and is
 * itself checked against that NumPy oracle (tests/test_oracle.py).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * One entry point, fwi_oracle_propagate(), is the shared time loop of
 * Propagator._propagate(): forward = inject wavelets at sources, record at
 * receivers, optionally store q^n; adjoint = the same loop in reverse time
 * with the residual injected at receivers, recording at sources and
 * accumulating the zero-lag image  img += mu^{n+1} * q^n.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double COEF2[] = {-2.0, 1.0};
static const double COEF4[] = {-5.0 / 2.0, 4.0 / 3.0, -1.0 / 12.0};
static const double COEF8[] = {-205.0 / 72.0, 8.0 / 5.0, -1.0 / 5.0, 8.0 / 315.0, -1.0 / 560.0};

/* d = sigma dt / 2 along one axis (damping_profiles() of the NumPy oracle). */
static void profile(double *p, int n, int npml, double sigma_max, double dt) {
    for (int i = 0; i < n; ++i) {
        double dist = 0.0;
        if (npml > 0) {
            double a = (double)npml - i, b = (double)i - (n - 1 - npml);
            dist = a > b ? a : b;
            if (dist < 0.0) dist = 0.0;
            p[i] = 0.5 * dt * sigma_max * (dist / npml) * (dist / npml);
        } else {
            p[i] = 0.0;
        }
    }
}

/* Returns 0 on success, non-zero on bad arguments / allocation failure.
 *   c         velocity, nz*ny*nx (ny = 1 for 2-D), x fastest
 *   inj_idx   flat grid indices of the ninj injection points
 *   inj_amp   nt x ninj amplitudes, multiplied by inj_scale
 *   rec_idx   flat indices of the nrec recording points; rec_out nt x nrec, times rec_scale
 *   q_store   NULL, or nt x npts: written when save_q, read when image != NULL
 *   image     NULL, or npts accumulator (+= u_next * q_store[n])
 */
int fwi_oracle_propagate(int ndim, int nz, int ny, int nx, int order, const double *c, double h,
                         double dt, int npml, double sigma_max, int nt, int reverse, int ninj,
                         const int64_t *inj_idx, const double *inj_amp, double inj_scale, int nrec,
                         const int64_t *rec_idx, double rec_scale, double *rec_out, int save_q,
                         double *q_store, double *image) {
    const double *a;
    int r;
    if (order == 2) { a = COEF2; r = 1; }
    else if (order == 4) { a = COEF4; r = 2; }
    else if (order == 8) { a = COEF8; r = 4; }
    else return 1;
    if (ndim == 2) ny = 1; else if (ndim != 3) return 2;
    if ((save_q || image) && !q_store) return 3;

    const int64_t npts = (int64_t)nz * ny * nx;
    /* zero-padded working fields: halo r on every used axis */
    const int ry = (ndim == 3) ? r : 0;
    const int64_t px = nx + 2 * r, py = ny + 2 * ry, pz = nz + 2 * r;
    const int64_t sy = px, sz = px * py, ptot = px * py * pz;
    double *bufa = calloc(ptot, sizeof(double)), *bufb = calloc(ptot, sizeof(double));
    double *C = malloc(npts * sizeof(double));
    double *pzr = malloc(nz * sizeof(double)), *pyr = malloc(ny * sizeof(double)),
           *pxr = malloc(nx * sizeof(double));
    double *src = calloc(npts, sizeof(double));
    if (!bufa || !bufb || !C || !pzr || !pyr || !pxr || !src) return 4;
    profile(pzr, nz, npml, sigma_max, dt);
    profile(pxr, nx, npml, sigma_max, dt);
    if (ndim == 3) profile(pyr, ny, npml, sigma_max, dt); else pyr[0] = 0.0;
    for (int64_t i = 0; i < npts; ++i) C[i] = dt * dt * c[i] * c[i];
    double coef[5];
    for (int k = 0; k <= r; ++k) coef[k] = a[k] / (h * h);

    double *u_prev = bufa, *u_cur = bufb;
    for (int s = 0; s < nt; ++s) {
        const int n = reverse ? nt - 1 - s : s;
        for (int i = 0; i < ninj; ++i) src[inj_idx[i]] += inj_amp[(int64_t)n * ninj + i] * inj_scale;
        double *qn = (save_q || image) ? q_store + (int64_t)n * npts : NULL;
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 0; z < nz; ++z) {
            for (int y = 0; y < ny; ++y) {
                const int64_t g0 = ((int64_t)z * ny + y) * nx;
                const int64_t p0 = (z + r) * sz + (y + ry) * sy + r;
                const double dzy = pzr[z] + pyr[y];
                for (int x = 0; x < nx; ++x) {
                    const double *u = u_cur + p0 + x;
                    double lap = ndim * coef[0] * u[0];
                    for (int k = 1; k <= r; ++k) {
                        double acc = (u[-k] + u[k]) + (u[-k * sz] + u[k * sz]);
                        if (ndim == 3) acc += u[-k * sy] + u[k * sy];
                        lap += coef[k] * acc;
                    }
                    const double d = dzy + pxr[x];
                    const double q = C[g0 + x] * (lap + src[g0 + x]);
                    const double un = (2.0 * u[0] - (1.0 - d) * u_prev[p0 + x] + q) / (1.0 + d);
                    u_prev[p0 + x] = un; /* next field overwrites the oldest one */
                    if (save_q) qn[g0 + x] = q;
                    if (image) image[g0 + x] += un * qn[g0 + x];
                }
            }
        }
        for (int i = 0; i < ninj; ++i) src[inj_idx[i]] = 0.0;
        double *t = u_prev; u_prev = u_cur; u_cur = t;
        for (int i = 0; i < nrec; ++i) {
            const int64_t f = rec_idx[i];
            const int64_t x = f % nx, y = (f / nx) % ny, z = f / ((int64_t)nx * ny);
            rec_out[(int64_t)n * nrec + i] = u_cur[(z + r) * sz + (y + ry) * sy + r + x] * rec_scale;
        }
    }
    free(bufa); free(bufb); free(C); free(pzr); free(pyr); free(pxr); free(src);
    return 0;
}
