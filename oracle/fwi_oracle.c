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
 *
 * abc = 1 switches the border from the sponge to the convolutional PML stated at the top of
 * oracle/fwi_oracle.py (memory variables psi_d, zeta_d per axis; the adjoint sweep runs the exact
 * transpose).  The memory variables are kept as full padded fields here: this is the checker, not the product.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double COEF2[] = {-2.0, 1.0};
static const double COEF4[] = {-5.0 / 2.0, 4.0 / 3.0, -1.0 / 12.0};
static const double COEF8[] = {-205.0 / 72.0, 8.0 / 5.0, -1.0 / 5.0, 8.0 / 315.0, -1.0 / 560.0};
static const double DCOEF2[] = {1.0 / 2.0};
static const double DCOEF4[] = {2.0 / 3.0, -1.0 / 12.0};
static const double DCOEF8[] = {4.0 / 5.0, -1.0 / 5.0, 4.0 / 105.0, -1.0 / 280.0};

/* CPML coefficients along one axis (cpml_profiles() of the NumPy oracle) */
static void cpml_profile(double *a, double *b, int n, int npml, double sigma_max, double alpha_max, double dt) {
    for (int i = 0; i < n; ++i) {
        double dist = 0.0;
        if (npml > 0) {
            double lo = (double)npml - i, hi = (double)i - (n - 1 - npml);
            dist = lo > hi ? lo : hi;
            if (dist < 0.0) dist = 0.0;
        }
        const double x = dist / (npml > 0 ? npml : 1);
        const double sig = sigma_max * x * x, alp = alpha_max * (1.0 - x);
        b[i] = exp(-(sig + alp) * dt);
        a[i] = sig > 0.0 ? sig / (sig + alp) * (b[i] - 1.0) : 0.0;
    }
}

/* first / second centred differences along one axis (stride st) of a zero-padded field */
static inline double d1(const double *u, int64_t st, const double *dc, int r) {
    double v = 0.0;
    for (int k = 1; k <= r; ++k) v += dc[k - 1] * (u[k * st] - u[-k * st]);
    return v;
}
static inline double d2(const double *u, int64_t st, const double *co, int r) {
    double v = co[0] * u[0];
    for (int k = 1; k <= r; ++k) v += co[k] * (u[k * st] + u[-k * st]);
    return v;
}

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
int fwi_oracle_propagate_abc(int ndim, int nz, int ny, int nx, int order, const double *c, double h,
                             double dt, int npml, double sigma_max, int nt, int reverse, int ninj,
                             const int64_t *inj_idx, const double *inj_amp, double inj_scale, int nrec,
                             const int64_t *rec_idx, double rec_scale, double *rec_out, int save_q,
                             double *q_store, double *image, int abc, double alpha_max) {
    const double *a, *dc0;
    int r;
    if (order == 2) { a = COEF2; dc0 = DCOEF2; r = 1; }
    else if (order == 4) { a = COEF4; dc0 = DCOEF4; r = 2; }
    else if (order == 8) { a = COEF8; dc0 = DCOEF8; r = 4; }
    else return 1;
    const int cpml = (abc == 1 && npml > 0);
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
    profile(pzr, nz, cpml ? 0 : npml, sigma_max, dt);
    profile(pxr, nx, cpml ? 0 : npml, sigma_max, dt);
    if (ndim == 3) profile(pyr, ny, cpml ? 0 : npml, sigma_max, dt); else pyr[0] = 0.0;
    /* CPML: axis order z, y, x = 0, 1, 2; per axis the padded memory fields psi, zeta, the scratch field
     * (a * zeta~ or a * psi~ of the adjoint sweep) and the 1-D coefficients */
    double *psi[3] = {0, 0, 0}, *zet[3] = {0, 0, 0}, *tmp = NULL, *ca[3] = {0, 0, 0}, *cb[3] = {0, 0, 0};
    const int nax[3] = {nz, ny, nx};
    const int64_t strd[3] = {sz, sy, 1};
    double dcf[4];
    for (int k = 0; k < r; ++k) dcf[k] = dc0[k] / h;
    if (cpml) {
        tmp = calloc(ptot, sizeof(double));
        if (!tmp) return 4;
        for (int d = 0; d < 3; ++d) {
            if (d == 1 && ndim == 2) continue;
            psi[d] = calloc(ptot, sizeof(double));
            zet[d] = calloc(ptot, sizeof(double));
            ca[d] = malloc(nax[d] * sizeof(double));
            cb[d] = malloc(nax[d] * sizeof(double));
            if (!psi[d] || !zet[d] || !ca[d] || !cb[d]) return 4;
            cpml_profile(ca[d], cb[d], nax[d], npml, sigma_max, alpha_max, dt);
        }
    }
    for (int64_t i = 0; i < npts; ++i) C[i] = dt * dt * c[i] * c[i];
    double coef[5];
    for (int k = 0; k <= r; ++k) coef[k] = a[k] / (h * h);

    double *u_prev = bufa, *u_cur = bufb;
    for (int s = 0; s < nt; ++s) {
        const int n = reverse ? nt - 1 - s : s;
        for (int i = 0; i < ninj; ++i) src[inj_idx[i]] += inj_amp[(int64_t)n * ninj + i] * inj_scale;
        double *qn = (save_q || image) ? q_store + (int64_t)n * npts : NULL;
        if (cpml) {
            /* memory variables advance with the newest field; their contribution joins the bracket via src[] */
            for (int d = 0; d < 3; ++d) {
                if (!psi[d]) continue;
                const int64_t st = strd[d];
                const double *A_ = ca[d], *B_ = cb[d];
                /* pass 1: forward psi <- b psi + a D u      | adjoint zeta~ <- b zeta~ + mu,  tmp = a zeta~ */
#pragma omp parallel for collapse(2) schedule(static)
                for (int z = 0; z < nz; ++z)
                    for (int y = 0; y < ny; ++y)
                        for (int x = 0; x < nx; ++x) {
                            const int i = d == 0 ? z : d == 1 ? y : x;
                            const int64_t p = (z + r) * sz + (y + ry) * sy + r + x;
                            if (!reverse) {
                                if (A_[i] != 0.0) psi[d][p] = B_[i] * psi[d][p] + A_[i] * d1(u_cur + p, st, dcf, r);
                            } else {
                                if (A_[i] != 0.0) {
                                    zet[d][p] = B_[i] * zet[d][p] + u_cur[p];
                                    tmp[p] = A_[i] * zet[d][p];
                                } else {
                                    tmp[p] = 0.0;
                                }
                            }
                        }
                /* pass 2: forward zeta <- b zeta + a (E u + D psi) | adjoint psi~ <- b psi~ - D mu - D tmp */
#pragma omp parallel for collapse(2) schedule(static)
                for (int z = 0; z < nz; ++z)
                    for (int y = 0; y < ny; ++y)
                        for (int x = 0; x < nx; ++x) {
                            const int i = d == 0 ? z : d == 1 ? y : x;
                            if (A_[i] == 0.0) continue;
                            const int64_t p = (z + r) * sz + (y + ry) * sy + r + x;
                            if (!reverse)
                                zet[d][p] = B_[i] * zet[d][p] +
                                            A_[i] * (d2(u_cur + p, st, coef, r) + d1(psi[d] + p, st, dcf, r));
                            else
                                psi[d][p] = B_[i] * psi[d][p] - d1(u_cur + p, st, dcf, r) - d1(tmp + p, st, dcf, r);
                        }
                /* pass 3: the bracket term: forward D psi + zeta | adjoint E (a zeta~) - D (a psi~) */
                if (reverse) { /* E tmp first (tmp = a zeta~), then tmp := a psi~ for the D term */
#pragma omp parallel for collapse(2) schedule(static)
                    for (int z = 0; z < nz; ++z)
                        for (int y = 0; y < ny; ++y)
                            for (int x = 0; x < nx; ++x) {
                                const int64_t p = (z + r) * sz + (y + ry) * sy + r + x;
                                src[((int64_t)z * ny + y) * nx + x] += d2(tmp + p, st, coef, r);
                            }
#pragma omp parallel for collapse(2) schedule(static)
                    for (int z = 0; z < nz; ++z)
                        for (int y = 0; y < ny; ++y)
                            for (int x = 0; x < nx; ++x) {
                                const int i = d == 0 ? z : d == 1 ? y : x;
                                const int64_t p = (z + r) * sz + (y + ry) * sy + r + x;
                                tmp[p] = A_[i] * psi[d][p];
                            }
                }
#pragma omp parallel for collapse(2) schedule(static)
                for (int z = 0; z < nz; ++z)
                    for (int y = 0; y < ny; ++y)
                        for (int x = 0; x < nx; ++x) {
                            const int64_t p = (z + r) * sz + (y + ry) * sy + r + x;
                            const int64_t gi = ((int64_t)z * ny + y) * nx + x;
                            if (!reverse) src[gi] += d1(psi[d] + p, st, dcf, r) + zet[d][p];
                            else src[gi] -= d1(tmp + p, st, dcf, r);
                        }
            }
        }
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
        if (cpml) memset(src, 0, npts * sizeof(double));
        else for (int i = 0; i < ninj; ++i) src[inj_idx[i]] = 0.0;
        double *t = u_prev; u_prev = u_cur; u_cur = t;
        for (int i = 0; i < nrec; ++i) {
            const int64_t f = rec_idx[i];
            const int64_t x = f % nx, y = (f / nx) % ny, z = f / ((int64_t)nx * ny);
            rec_out[(int64_t)n * nrec + i] = u_cur[(z + r) * sz + (y + ry) * sy + r + x] * rec_scale;
        }
    }
    free(bufa); free(bufb); free(C); free(pzr); free(pyr); free(pxr); free(src);
    free(tmp);
    for (int d = 0; d < 3; ++d) { free(psi[d]); free(zet[d]); free(ca[d]); free(cb[d]); }
    return 0;
}

/* the sponge-only entry point (kept: the signature the first round's tests and baseline were built on) */
int fwi_oracle_propagate(int ndim, int nz, int ny, int nx, int order, const double *c, double h,
                         double dt, int npml, double sigma_max, int nt, int reverse, int ninj,
                         const int64_t *inj_idx, const double *inj_amp, double inj_scale, int nrec,
                         const int64_t *rec_idx, double rec_scale, double *rec_out, int save_q,
                         double *q_store, double *image) {
    return fwi_oracle_propagate_abc(ndim, nz, ny, nx, order, c, h, dt, npml, sigma_max, nt, reverse, ninj, inj_idx,
                                    inj_amp, inj_scale, nrec, rec_idx, rec_scale, rec_out, save_q, q_store, image,
                                    0, 0.0);
}
