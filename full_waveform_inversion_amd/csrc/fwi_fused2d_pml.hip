// 2-D temporal blocking WITH the convolutional PML inside the launch (fwi_config.abc = FWI_ABC_CPML).
//
// Same tiling as step2d_fused (fwi_fused2d.hip): a workgroup loads its FT x FT tile plus a KS r-cell halo of u^n
// and u^{n-1} once, advances it KS time steps in LDS and writes the tile of the last two time levels.  Here the
// memory variables of the border recursion (oracle/fwi_oracle.py, top; fwi_pml.hip for the slab form) travel with
// the tile: psi / zeta of the z border (rows) and of the x border (columns) sit in LDS beside the two field
// images, are advanced inside every sub-step and written back for the tile's own cells at the end.  That replaces
// 6 (forward) / 9 (adjoint) slab launches per TIME STEP around a one-step kernel by nothing: one launch per KS
// steps, as with the sponge (1024^2, npml 40: 19.5 -> see DESIGN.md s.4 CPML us/step).
//
// Dependencies.  Per axis d, forward:   psi' = b psi + a D u;  zeta' = b zeta + a (E u + D psi');
//                                        q = C (L u + sum_d (D psi' + zeta')),  u' = 2u - u_prev + q
// so u'(i) depends on u(i +- 2r) along a border axis (through psi'(i +- r)), not on u(i +- r) only.  The overlapped
// tiling budgets r cells of validity loss per sub-step, which is still enough here because of WHERE the border
// cells are: fused2d_cpml_supported() admits a grid only if every border cell a tile holds in its extended region
// lies in that tile's interior -- at least HL cells from every INWARD edge of the region, where the extra r cells
// are always valid -- with nothing but the outside of the grid (exact zeros) on its other side.  So that a narrow last
// tile does not disqualify a grid (it did for 5 of 6 grid sizes at npml 40), the tiles of this kernel are WHOLE with one
// overlap seam in the middle of each axis (fused2d_origin: the upper half of the tiles is anchored at the high end; the
// two tiles at the seam both compute the overlap, the lower one stores it, the host's injection / sampling tables
// follow the same ownership).  Grids that still do not qualify (border wider than FT - HL = 48, a single tile per axis,
// two tiles that both see both borders) keep the slab path.  The adjoint sweep runs the transposed recursion: zt' = b zt + mu; pt' = b pt - D mu - D (a zt');
// term = E (a zt') - D (a pt'), two dependent neighbour reads, hence two barriers before the update.
//
// C is read from global memory per group here (L2-resident; the third LDS image of step2d_fused is what the four
// memory-variable images take): LDS 74 KB (fields) + up to 77 KB (NPB = 48).
//
// No reference counterpart.  Parity: tests/test_gpu_cpml.py (vs the oracle, and vs the slab path).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "fwi_kernels.h"

namespace fwi {

namespace {

struct alignas(16) q4 {
    float v[4];
};
typedef float nt4 __attribute__((ext_vector_type(4)));

constexpr int cpml_halo(int r) { return (FUSED2D_STEPS * r + 3) / 4 * 4; }

}  // namespace

bool fused2d_cpml_supported(const GridDesc &g, int npml) {
    if (g.ndim != 2 || g.r != 4 || npml < 1) return false;  // (O(8) is what is instantiated)
    const int HL = cpml_halo(g.r), FT = FUSED2D_TILE;
    if (npml > FT - HL) return false;  // the memory-variable images hold 48 border cells
    // Every tile (whole tiles, one overlap seam per axis: fused2d_origin) must see each border either not at all or
    // from a side where nothing but the outside of the grid lies beyond it, and never both borders of an axis
    for (int n : {g.nz, g.nx}) {
        if (n <= FT) return false;
        const int nt = (n + FT - 1) / FT;
        for (int t = 0; t < nt; ++t) {
            const int lo = fused2d_origin(t, n, FT, 1) - HL, hi = lo + FT + 2 * HL;
            const bool low = lo < npml, high = hi > n - npml;  // border cells in the extended region
            if (low && high) return false;
            if (low && lo > 0) return false;     // the low border would begin inside this tile's halo
            if (high && hi < n) return false;    // ... the high border end in it
        }
    }
    return true;
}

// NPB: border width the memory-variable images are sized for (npml <= NPB).  REV: the adjoint recursion.
// Diagnostic builds only (-DFWI_CPML_ABLATE=bits, timing A/B; results are wrong): 1 = no neighbour pass, 2 = no border
// term in the update, 4 = constant C instead of the global load.
#ifndef FWI_CPML_ABLATE
#define FWI_CPML_ABLATE 0
#endif
template <int R, int KS, int FT, int NPB, bool SAVE_Q, bool IMAGE, bool REV>
__global__ __launch_bounds__(1024) void step2d_fused_cpml(Fused2dArgs a, GridDesc g) {
    constexpr int FNT = 1024;
    constexpr int HL = (KS * R + 3) / 4 * 4;
    constexpr int E = FT + 2 * HL, E4 = E / 4, NG = E * E4;
    constexpr int NI = FT * (FT / 4), IPT = (NI + FNT - 1) / FNT;
    constexpr int NXG = NPB / 4 + 1;  // 16-byte groups per row of the x images (one more: the high border need not
                                      // start on a group boundary)
    static_assert(E % 4 == 0 && HL % 4 == 0 && FT % 4 == 0 && NPB % 4 == 0, "float4 alignment");
    __shared__ q4 fa[E][E4];    // field A (starts as u^n)
    __shared__ q4 fb[E][E4];    // field B (starts as u^{n-1})
    __shared__ q4 pz[NPB][E4];  // psi (adjoint: pt) of the z border: image row r <-> extended row pbz + r
    __shared__ q4 zz[NPB][E4];  // zeta (adjoint: zt) of the z border
    __shared__ q4 px[E][NXG];   // psi / pt of the x border: image group gx <-> extended group pbx / 4 + gx
    __shared__ q4 zx[E][NXG];   // zeta / zt of the x border
    __shared__ float az[E], bz[E];             // a, b by extended row; 0 off the border and off the grid
    __shared__ __align__(16) float ax[E], bx[E];  // ... by extended column (read one 16-byte group at a time)

    const int tid = threadIdx.x;
    const int ntx = (g.nx + FT - 1) / FT;
    int tile = blockIdx.x;
    if (a.tile_order) {  // several rounds of tiles: the border tiles (about twice an interior tile's work) go first
        tile = a.tile_order[tile];
    } else if (a.xcd_remap) {  // XCD-contiguous tile runs, as in step2d_fused
        const int nblk = gridDim.x, x = tile & 7, q = nblk >> 3, r = nblk & 7;
        tile = x * q + min(x, r) + (tile >> 3);
    }
    const int tz = tile / ntx, tx = tile % ntx;
    // whole tiles with one overlap seam per axis (fused2d_origin); a tile stores from its first OWNED cell on
    const int z0 = fused2d_origin(tz, g.nz, FT, 1) - HL, x0 = fused2d_origin(tx, g.nx, FT, 1) - HL;  // extended (0, 0)
    const int own_lz = fused2d_own(tz, g.nz, FT, 1) - z0, own_l4 = (fused2d_own(tx, g.nx, FT, 1) - x0) >> 2;  // (>= HL, HL / 4)
    const int npml = a.pml_npml;
    const int xpad = (g.nx + 3) & ~3;

    // ---- fill: the two field images by LDS-DMA (zeros outside the grid come from the padded arrays' halo) ----
    {
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        constexpr int GPT = (NG + FNT - 1) / FNT;
        const int wave0 = tid & ~63;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int g0 = wave0 + i * FNT;
            if (g0 >= NG) break;
            const int gi = tid + i * FNT;
            if (gi < NG) {
                const int lz = gi / E4, l4 = gi % E4;
                const int zc = min(max(z0 + lz, -1), g.nz), xc = min(max(x0 + 4 * l4, -4), xpad);
                const int64_t p = g.off0 + (int64_t)zc * g.sz + xc;
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.u_cur + p), (lptr_t *)(&fa[0][0] + g0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.u_prev + p), (lptr_t *)(&fb[0][0] + g0), 16, 0, 0);
            }
        }
    }
    // ---- which border this tile holds per axis (never both ends of one axis: fused2d_cpml_supported) ----
    // pbz: extended row of image row 0 (-1: the tile sees no z border); pbx likewise, rounded down to a group
    int pbz = -1, pbx = -1;
    if (z0 < npml) pbz = -z0; else if (z0 + E > g.nz - npml) pbz = g.nz - npml - z0;
    if (x0 < npml) pbx = -x0; else if (x0 + E > g.nx - npml) pbx = (g.nx - npml - x0) & ~3;
    for (int i = tid; i < E; i += FNT) {
        const int z = z0 + i, x = x0 + i;
        const bool zin = z >= 0 && z < g.nz && (z < npml || z >= g.nz - npml);
        const bool xin = x >= 0 && x < g.nx && (x < npml || x >= g.nx - npml);
        az[i] = zin ? a.pml_a[0][z] : 0.f;
        bz[i] = zin ? a.pml_b[0][z] : 0.f;
        ax[i] = xin ? a.pml_a[1][x] : 0.f;
        bx[i] = xin ? a.pml_b[1][x] : 0.f;
    }
    // ---- the memory variables of every border cell in the extended region (halo cells too: each tile advances
    // its own copy of them, as it does with the halo of the fields) ----
    if (pbz >= 0) {
        for (int i = tid; i < npml * E4; i += FNT) {
            const int r = i / E4, l4 = i % E4;
            const int z = z0 + pbz + r, x = x0 + 4 * l4;  // z is a border row of the grid by construction
            if (x >= 0 && x < g.cx) {  // (rows of the compact arrays are padded to cx: whole vector)
                const int j = z < npml ? z : z - (g.nz - 2 * npml);  // slab plane (nz > 2 npml here)
                const int64_t o = (int64_t)j * g.cx + x;
                pz[r][l4] = *reinterpret_cast<const q4 *>(a.pml_psi[0] + o);
                zz[r][l4] = *reinterpret_cast<const q4 *>(a.pml_zeta[0] + o);
            } else {
                pz[r][l4] = q4{{0.f, 0.f, 0.f, 0.f}};
                zz[r][l4] = q4{{0.f, 0.f, 0.f, 0.f}};
            }
        }
    }
    if (pbx >= 0) {
        for (int i = tid; i < E * NXG; i += FNT) {
            const int lz = i / NXG, gx = i % NXG;
            const int z = z0 + lz;
            q4 p = {{0.f, 0.f, 0.f, 0.f}}, q = p;
            if (z >= 0 && z < g.nz) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int x = x0 + pbx + 4 * gx + c;
                    if (x >= 0 && x < g.nx && (x < npml || x >= g.nx - npml)) {
                        const int j = x < npml ? x : x - (g.nx - 2 * npml);
                        const int64_t o = (int64_t)z * (2 * npml) + j;
                        p.v[c] = a.pml_psi[1][o];
                        q.v[c] = a.pml_zeta[1][o];
                    }
                }
            }
            px[lz][gx] = p;
            zx[lz][gx] = q;
        }
    }
    q4 gacc[IPT];
    if (IMAGE) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) gacc[i] = {{0.f, 0.f, 0.f, 0.f}};
    }
    __syncthreads();

    q4(*cur)[E4] = fa;
    q4(*prv)[E4] = fb;
    const int s0 = a.inj_start ? a.inj_start[tile] : 0, s1 = a.inj_start ? a.inj_start[tile + 1] : 0;
    const int r0 = a.rec_start ? a.rec_start[tile] : 0, r1 = a.rec_start ? a.rec_start[tile + 1] : 0;
    const int pg = pbx >> 2;  // first extended group of the x images (when pbx >= 0)
    // groups of the x images that hold border cells (the high border may start up to 3 cells into its first group)
    const int ngx = pbx < 0 ? 0 : (((x0 < npml) ? 0 : (g.nx - npml - x0) - pbx) + npml + 3) >> 2;

    // 12 consecutive values around group l4 of a row: [l4 - 1][l4][l4 + 1]
    auto window = [&](const q4 *row, int l4, float *X) __attribute__((always_inline)) {
        const q4 l = row[max(l4 - 1, 0)], c = row[l4], r = row[min(l4 + 1, E4 - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            X[j] = l.v[j];
            X[4 + j] = c.v[j];
            X[8 + j] = r.v[j];
        }
    };
    // the same for an x image (groups outside the image are zero), optionally times the coefficient a(x)
    auto xwindow = [&](const q4 *row, int gx, bool times_a, int l4, float *X) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int gg = gx - 1 + t;
            q4 w = {{0.f, 0.f, 0.f, 0.f}}, c = {{1.f, 1.f, 1.f, 1.f}};
            if (gg >= 0 && gg < NXG) {
                w = row[gg];
                if (times_a) c = reinterpret_cast<const q4 *>(ax)[l4 - 1 + t];  // (an image group is inside the region)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) X[4 * t + j] = times_a ? c.v[j] * w.v[j] : w.v[j];
        }
    };

    auto substep = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const int n = a.n0 + s * a.dn;
        const bool qstep = (SAVE_Q || IMAGE) && (a.istride <= 1 || n % a.istride == 0);
        float *const qslot = a.q_base + (int64_t)(a.istride <= 1 ? n : n / a.istride) * g.npts;
        constexpr int lo = (s + 1) * R, hi = E - (s + 1) * R;
        constexpr int c_lo = lo / 4, w4 = (hi + 3) / 4 - c_lo, nact = (hi - lo) * w4;  // (active columns only, as in step2d_fused)
        constexpr int TRIPS = (nact + FNT - 1) / FNT;
        nt4 qv[IPT];
        if (IMAGE) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int ii = tid + i * FNT;
                const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
                const int z = z0 + lz, x = x0 + 4 * l4;
                qv[i] = nt4{0.f, 0.f, 0.f, 0.f};
                if (qstep && ii < NI && z < g.nz && x < g.nx)
                    qv[i] = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(qslot + (int64_t)z * g.cx + x));
            }
        }
        // ---- memory variables that the update reads at NEIGHBOURING cells: psi' (forward); zt', then pt' (adjoint).
        // Only over the transverse range this sub-step still updates: columns [lo, hi) of the z images, rows [lo, hi)
        // of the x images (one item per thread at npml = 40: 40 x 22 and 88 x 10 in the first sub-step) ----
        constexpr int c4 = lo / 4, nc4 = (hi - lo) / 4, nrw = hi - lo;
        static_assert(lo % 4 == 0 && hi % 4 == 0, "active columns are whole groups");
        if (FWI_CPML_ABLATE & 1) {
            __syncthreads();
        } else if (!REV) {
            if (pbz >= 0) {
                for (int i = tid; i < npml * nc4; i += FNT) {
                    const int r = i / nc4, l4 = c4 + i % nc4, lz = pbz + r;
                    q4 du = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        const q4 um = cur[lz - k][l4], up = cur[lz + k][l4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) du.v[j] = fmaf(a.pml_dk1[k], up.v[j] - um.v[j], du.v[j]);
                    }
                    q4 p = pz[r][l4];
                    const float av = az[lz], bv = bz[lz];
#pragma unroll
                    for (int j = 0; j < 4; ++j) p.v[j] = fmaf(bv, p.v[j], av * du.v[j]);
                    pz[r][l4] = p;
                }
            }
            if (pbx >= 0) {
                for (int i = tid; i < nrw * ngx; i += FNT) {
                    const int lz = lo + i / ngx, gx = i % ngx, l4 = pg + gx;
                    float X[12];
                    window(cur[lz], l4, X);
                    q4 p = px[lz][gx];
                    const q4 a4 = reinterpret_cast<const q4 *>(ax)[l4], b4 = reinterpret_cast<const q4 *>(bx)[l4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float du = 0.f;
#pragma unroll
                        for (int k = 1; k <= R; ++k) du = fmaf(a.pml_dk1[k], X[4 + j + k] - X[4 + j - k], du);
                        p.v[j] = fmaf(b4.v[j], p.v[j], a4.v[j] * du);
                    }
                    px[lz][gx] = p;
                }
            }
            __syncthreads();
        } else {
            if (pbz >= 0) {
                for (int i = tid; i < npml * nc4; i += FNT) {
                    const int r = i / nc4, l4 = c4 + i % nc4, lz = pbz + r;
                    const q4 u = cur[lz][l4];
                    q4 z = zz[r][l4];
                    const float bv = bz[lz];
#pragma unroll
                    for (int j = 0; j < 4; ++j) z.v[j] = fmaf(bv, z.v[j], u.v[j]);
                    zz[r][l4] = z;
                }
            }
            if (pbx >= 0) {
                for (int i = tid; i < nrw * ngx; i += FNT) {
                    const int lz = lo + i / ngx, gx = i % ngx, l4 = pg + gx;
                    const q4 u = cur[lz][l4];
                    q4 z = zx[lz][gx];
                    const q4 b4 = reinterpret_cast<const q4 *>(bx)[l4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) z.v[j] = fmaf(b4.v[j], z.v[j], u.v[j]);
                    zx[lz][gx] = z;
                }
            }
            __syncthreads();
            if (pbz >= 0) {
                for (int i = tid; i < npml * nc4; i += FNT) {
                    const int r = i / nc4, l4 = c4 + i % nc4, lz = pbz + r;
                    q4 d = {{0.f, 0.f, 0.f, 0.f}};  // D mu + D (a zt')
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        const q4 um = cur[lz - k][l4], up = cur[lz + k][l4];
                        q4 am = {{0.f, 0.f, 0.f, 0.f}}, ap = am;
                        if (r - k >= 0) am = zz[r - k][l4];
                        if (r + k < npml) ap = zz[r + k][l4];
                        const float cm = az[lz - k], cp = az[lz + k];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            d.v[j] = fmaf(a.pml_dk1[k], (up.v[j] - um.v[j]) + (cp * ap.v[j] - cm * am.v[j]), d.v[j]);
                    }
                    q4 p = pz[r][l4];
                    const float bv = bz[lz];
#pragma unroll
                    for (int j = 0; j < 4; ++j) p.v[j] = bv * p.v[j] - d.v[j];
                    pz[r][l4] = p;
                }
            }
            if (pbx >= 0) {
                for (int i = tid; i < nrw * ngx; i += FNT) {
                    const int lz = lo + i / ngx, gx = i % ngx, l4 = pg + gx;
                    float X[12], A[12];
                    window(cur[lz], l4, X);
                    xwindow(zx[lz], gx, true, l4, A);
                    q4 p = px[lz][gx];
                    const q4 b4 = reinterpret_cast<const q4 *>(bx)[l4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float d = 0.f;
#pragma unroll
                        for (int k = 1; k <= R; ++k)
                            d = fmaf(a.pml_dk1[k], (X[4 + j + k] - X[4 + j - k]) + (A[4 + j + k] - A[4 + j - k]), d);
                        p.v[j] = b4.v[j] * p.v[j] - d;
                    }
                    px[lz][gx] = p;
                }
            }
            __syncthreads();
        }
        // ---- stencil update of the active region ----
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int gi = tid + i * FNT;
            if (gi >= nact) break;
            const int lz = lo + gi / w4, l4 = c_lo + gi % w4;
            // C of this group from global memory (zero halo outside the grid), in flight under the LDS reads below
            const int zc = min(max(z0 + lz, -1), g.nz), xc = min(max(x0 + 4 * l4, -4), xpad);
            const q4 Cc = (FWI_CPML_ABLATE & 4) ? q4{{.1f, .1f, .1f, .1f}}
                                                : *reinterpret_cast<const q4 *>(a.C + g.off0 + (int64_t)zc * g.sz + xc);
            q4 term = {{0.f, 0.f, 0.f, 0.f}};
            const int rz = lz - pbz;
            const bool zreach = !(FWI_CPML_ABLATE & 2) && pbz >= 0 && rz >= -R && rz < npml + R;  // rows the z border reaches (D: +- R)
            const int gx = l4 - pg;
            const bool xreach = !(FWI_CPML_ABLATE & 2) && pbx >= 0 && gx >= -1 && gx <= ngx;      // groups the x border reaches
            if (REV) {
                // the adjoint term needs no second difference of the field: formed first, while little else is live
                if (zreach) {
                    q4 a0 = {{0.f, 0.f, 0.f, 0.f}};
                    if (rz >= 0 && rz < npml) {
                        a0 = zz[rz][l4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) a0.v[j] *= az[lz];
                    }
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        const bool inm = rz - k >= 0 && rz - k < npml, inp = rz + k >= 0 && rz + k < npml;
                        const float cm = inm ? az[lz - k] : 0.f, cp = inp ? az[lz + k] : 0.f;
                        q4 am = {{0.f, 0.f, 0.f, 0.f}}, ap = am, bm = am, bp = am;
                        if (inm) {
                            am = zz[rz - k][l4];
                            bm = pz[rz - k][l4];
                        }
                        if (inp) {
                            ap = zz[rz + k][l4];
                            bp = pz[rz + k][l4];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            term.v[j] = fmaf(a.ck[k], fmaf(-2.f, a0.v[j], cp * ap.v[j] + cm * am.v[j]), term.v[j]);
                            term.v[j] = fmaf(-a.pml_dk[k], cp * bp.v[j] - cm * bm.v[j], term.v[j]);
                        }
                    }
                }
                if (xreach) {
                    float A[12];
                    xwindow(zx[lz], gx, true, l4, A);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float e2 = 0.f;
#pragma unroll
                        for (int k = 1; k <= R; ++k)
                            e2 = fmaf(a.ck[k], fmaf(-2.f, A[4 + j], A[4 + j + k] + A[4 + j - k]), e2);
                        term.v[j] += e2;
                    }
                    xwindow(px[lz], gx, true, l4, A);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float db = 0.f;
#pragma unroll
                        for (int k = 1; k <= R; ++k) db = fmaf(a.pml_dk[k], A[4 + j + k] - A[4 + j - k], db);
                        term.v[j] -= db;
                    }
                }
            }
            float X[12];
            window(cur[lz], l4, X);
            q4 ez = {{0.f, 0.f, 0.f, 0.f}}, ex = ez;  // second differences along z and along x, in difference form
#pragma unroll
            for (int k = R; k >= 1; --k) {
                const q4 zm = cur[lz - k][l4], zp = cur[lz + k][l4];
                const float ck = a.ck[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ez.v[j] = fmaf(ck, fmaf(-2.f, X[4 + j], zm.v[j] + zp.v[j]), ez.v[j]);
                    ex.v[j] = fmaf(ck, fmaf(-2.f, X[4 + j], X[4 + j - k] + X[4 + j + k]), ex.v[j]);
                }
            }
            if (!REV) {
                if (zreach) {
                    q4 dp = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        q4 pm = {{0.f, 0.f, 0.f, 0.f}}, pp = pm;
                        if (rz - k >= 0 && rz - k < npml) pm = pz[rz - k][l4];
                        if (rz + k >= 0 && rz + k < npml) pp = pz[rz + k][l4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) dp.v[j] = fmaf(a.pml_dk[k], pp.v[j] - pm.v[j], dp.v[j]);
                    }
                    term = dp;
                    if (rz >= 0 && rz < npml) {
                        q4 z = zz[rz][l4];
                        const float av = az[lz], bv = bz[lz];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            z.v[j] = fmaf(bv, z.v[j], av * (ez.v[j] + dp.v[j]));
                            term.v[j] += z.v[j];
                        }
                        zz[rz][l4] = z;
                    }
                }
                if (xreach) {
                    float P[12];
                    xwindow(px[lz], gx, false, l4, P);
                    const bool own = gx >= 0 && gx < NXG;
                    q4 z = {{0.f, 0.f, 0.f, 0.f}};
                    if (own) z = zx[lz][gx];
                    const q4 a4 = reinterpret_cast<const q4 *>(ax)[l4], b4 = reinterpret_cast<const q4 *>(bx)[l4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float dp = 0.f;
#pragma unroll
                        for (int k = 1; k <= R; ++k) dp = fmaf(a.pml_dk[k], P[4 + j + k] - P[4 + j - k], dp);
                        z.v[j] = fmaf(b4.v[j], z.v[j], a4.v[j] * (ex.v[j] + dp));  // (a = b = 0 off the border)
                        term.v[j] += dp + z.v[j];
                    }
                    if (own) zx[lz][gx] = z;
                }
            }
            const q4 up = prv[lz][l4];
            q4 q, un;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q.v[j] = Cc.v[j] * ((ez.v[j] + ex.v[j]) + term.v[j]);
                un.v[j] = (2.f * X[4 + j] - up.v[j]) + q.v[j];
            }
            prv[lz][l4] = un;
            if (SAVE_Q && qstep) {
                const int z = z0 + lz, x = x0 + 4 * l4;
                const bool interior = lz >= own_lz && lz < HL + FT && l4 >= own_l4 && l4 < (HL + FT) / 4;
                if (interior && z < g.nz && x < g.nx) {
                    nt4 v = {q.v[0], q.v[1], q.v[2], q.v[3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<nt4 *>(qslot + (int64_t)z * g.cx + x));
                }
            }
        }
        __syncthreads();
        // ---- injection, sampling, imaging: as in step2d_fused ----
        if (s1 > s0) {
            for (int i = s0 + tid; i < s1; i += FNT) {
                const int run = a.inj_run[i];
                const int lz = a.inj_lz[i], lx = a.inj_lx[i];
                const float amp0 = a.inj_amp[(int64_t)n * a.ninj + a.inj_col[i]];
                float su = a.inj_cu[i] * amp0, sq = a.inj_cq[i] * amp0;
                if (run == 0) continue;
                for (int j = i + 1; j < i + run; ++j) {
                    const float amp = a.inj_amp[(int64_t)n * a.ninj + a.inj_col[j]];
                    su += a.inj_cu[j] * amp;
                    sq += a.inj_cq[j] * amp;
                }
                atomicAdd(&prv[lz][lx >> 2].v[lx & 3], su);
                if (SAVE_Q && qstep && a.inj_interior[i]) atomicAdd(qslot + a.inj_cidx[i], sq);
            }
            __syncthreads();
        }
        for (int i = r0 + tid; i < r1; i += FNT) {
            const int lz = a.rec_lz[i], lx = a.rec_lx[i];
            a.rec_out[(int64_t)n * a.nrec + a.rec_col[i]] = prv[lz][lx >> 2].v[lx & 3] * a.rec_scale;
        }
        if (IMAGE && qstep) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int ii = tid + i * FNT;
                if (ii >= NI) break;
                const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
                const q4 u = prv[lz][l4];
#pragma unroll
                for (int j = 0; j < 4; ++j) gacc[i].v[j] = fmaf(u.v[j], qv[i][j], gacc[i].v[j]);
            }
        }
        q4(*t)[E4] = cur;
        cur = prv;
        prv = t;
    };
    static_assert(KS == 4, "sub-steps are spelled out below");
    substep(std::integral_constant<int, 0>{});
    substep(std::integral_constant<int, 1>{});
    substep(std::integral_constant<int, 2>{});
    substep(std::integral_constant<int, 3>{});

    // ---- write the interior of the last two time levels, the gradient contribution, and the memory variables of
    // the tile's own border cells ----
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int ii = tid + i * FNT;
        if (ii >= NI) break;
        const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
        const int z = z0 + lz, x = x0 + 4 * l4;
        if (z >= g.nz || x >= g.nx || lz < own_lz || l4 < own_l4) continue;  // (below the seam's overlap: the other tile's)
        const int64_t p = g.off0 + (int64_t)z * g.sz + x;
        *reinterpret_cast<q4 *>(a.out_cur + p) = cur[lz][l4];
        *reinterpret_cast<q4 *>(a.out_prev + p) = prv[lz][l4];
        if (IMAGE) {
            float *gp = a.g + (int64_t)z * g.cx + x;
            q4 gv = *reinterpret_cast<const q4 *>(gp);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv.v[j] += gacc[i].v[j];
            *reinterpret_cast<q4 *>(gp) = gv;
        }
    }
    if (pbz >= 0) {  // (a border row this tile holds is one of its interior rows -- its own, or, in the tile just above
                     // the seam of a three-tile axis, one the tile below owns and stores)
        for (int i = tid; i < npml * (FT / 4); i += FNT) {
            const int r = i / (FT / 4), l4 = HL / 4 + i % (FT / 4);
            const int z = z0 + pbz + r, x = x0 + 4 * l4;
            if (x >= g.nx || l4 < own_l4 || pbz + r < own_lz || pbz + r >= HL + FT) continue;
            const int j = z < npml ? z : z - (g.nz - 2 * npml);
            const int64_t o = (int64_t)j * g.cx + x;
            *reinterpret_cast<q4 *>(a.pml_psi_out[0] + o) = pz[r][l4];
            *reinterpret_cast<q4 *>(a.pml_zeta_out[0] + o) = zz[r][l4];
        }
    }
    if (pbx >= 0) {
        for (int i = tid; i < FT * NXG; i += FNT) {
            const int lz = HL + i / NXG, gx = i % NXG;
            const int z = z0 + lz;
            if (z >= g.nz || lz < own_lz) continue;
            const q4 p = px[lz][gx], q = zx[lz][gx];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x = x0 + pbx + 4 * gx + c, lx = pbx + 4 * gx + c;
                if (x >= 0 && x < g.nx && (x < npml || x >= g.nx - npml) && lx >= 4 * own_l4 && lx < HL + FT) {
                    const int j = x < npml ? x : x - (g.nx - 2 * npml);
                    const int64_t o = (int64_t)z * (2 * npml) + j;
                    a.pml_psi_out[1][o] = p.v[c];
                    a.pml_zeta_out[1][o] = q.v[c];
                }
            }
        }
    }
}

template <int R, int NPB, bool REV>
static hipError_t launch_cpml_mode(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    constexpr int KS = FUSED2D_STEPS, FT = FUSED2D_TILE;
    const dim3 grid(fused2d_num_tiles(g));
    if (a.mode == 1)
        hipLaunchKernelGGL((step2d_fused_cpml<R, KS, FT, NPB, true, false, REV>), grid, dim3(1024), 0, s, a, g);
    else if (a.mode == 2)
        hipLaunchKernelGGL((step2d_fused_cpml<R, KS, FT, NPB, false, true, REV>), grid, dim3(1024), 0, s, a, g);
    else
        hipLaunchKernelGGL((step2d_fused_cpml<R, KS, FT, NPB, false, false, REV>), grid, dim3(1024), 0, s, a, g);
    return hipGetLastError();
}

template <int NPB>
static hipError_t launch_cpml_npb(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    return a.pml_rev ? launch_cpml_mode<4, NPB, true>(g, a, s) : launch_cpml_mode<4, NPB, false>(g, a, s);
}

hipError_t launch_fused2d_cpml(const GridDesc &g, const Fused2dArgs &a0, hipStream_t s) {
    static const bool no_remap = getenv("FWI_FUSED2D_NOREMAP") != nullptr;
    if (!fused2d_cpml_supported(g, a0.pml_npml) || a0.inc) return hipErrorInvalidValue;
    Fused2dArgs a = a0;
    a.xcd_remap = no_remap ? 0 : 1;
    if (a.pml_npml <= 16) return launch_cpml_npb<16>(g, a, s);
    if (a.pml_npml <= 32) return launch_cpml_npb<32>(g, a, s);
    return launch_cpml_npb<48>(g, a, s);
}

}  // namespace fwi
