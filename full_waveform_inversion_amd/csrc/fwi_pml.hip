// Convolutional PML (fwi_config.abc = FWI_ABC_CPML): the memory variables of the absorbing border and their
// contribution to the time step, as slab kernels around the unchanged (undamped) step kernels.
//
// No reference counterpart; the recursion and its exact transpose are stated at the top of
// oracle/fwi_oracle.py.  Per axis d the memory variables psi_d, zeta_d exist in that axis' border only
// (`nslab` = 2 npml planes of the grid, stored compactly), with 1-D coefficients a_d, b_d:
//
//   forward                                   adjoint sweep (mu = the newest adjoint field)
//   K1  psi  <- b psi  + a D u                zt <- b zt + mu
//   K2  zeta <- b zeta + a (E u + D psi)      pt <- b pt - D mu - D (a zt)
//   --- the step kernel: u' = 2u - u_prev + C (L u + injection) ---
//   K3  u' += C (D psi + zeta)  [q too]       u' += C (E (a zt) - D (a pt))
// (forward: K2 rides along in K3's launch -- both need D psi', zeta' only its own point -- so a step is 1 + 2 x axes
// launches forward and 1 + 3 x axes in the adjoint sweep, where pt' feeds K3 through its neighbours)
//
// D = centred first difference, E = second-difference star along d.  K3 covers the border widened by the stencil
// radius (D psi reaches that far).  These are bandwidth-trivial kernels over <= 2 (npml + r) planes per axis.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "fwi_kernels.h"

namespace fwi {

namespace {

// grid index of slab plane j, and slab plane of grid index i (-1: not in the border)
__device__ __forceinline__ int slab_to_grid(int j, int n, int npml) {
    return (n <= 2 * npml || j < npml) ? j : n - 2 * npml + j;
}
__device__ __forceinline__ int grid_to_slab(int i, int n, int npml) {
    if (i < 0 || i >= n) return -1;
    if (n <= 2 * npml || i < npml) return i;
    return i >= n - npml ? i - (n - 2 * npml) : -1;
}

template <typename T>
struct AxisView {
    int d;            // 0 = z, 1 = y, 2 = x
    int n, npml;      // grid extent along d, border width
    int64_t gst;      // stride of the padded fields along d
    int64_t ast;      // stride of the compact memory-variable arrays along d
    const T *a, *b;   // 1-D coefficients (length n)
};

// VL consecutive x values of one thread: 16 bytes (float4 / double2) for the z and y borders, whose rows run along
// x; one value for the x border.  Memory-variable rows are padded to cx (a multiple of 4) so the vectors stay aligned
// for any nx; the pad columns hold zeros (u = 0 and C = 0 there).
template <typename T, int VL>
struct alignas(sizeof(T) * VL) vecn {
    T v[VL];
};
template <typename T, int VL>
__device__ __forceinline__ vecn<T, VL> ldn(const T *p) { return *reinterpret_cast<const vecn<T, VL> *>(p); }
template <typename T, int VL>
__device__ __forceinline__ void stn(T *p, const vecn<T, VL> &x) { *reinterpret_cast<vecn<T, VL> *>(p) = x; }

// VL values of a memory variable at grid index i along d (0 outside the border), given the aux offset of the other
// two coordinates
template <typename T, int VL>
__device__ __forceinline__ vecn<T, VL> aux_at(const T *aux, const AxisView<T> &v, int64_t base, int i) {
    const int j = grid_to_slab(i, v.n, v.npml);
    vecn<T, VL> z;
#pragma unroll
    for (int q = 0; q < VL; ++q) z.v[q] = T(0);
    return j < 0 ? z : ldn<T, VL>(aux + base + (int64_t)j * v.ast);
}

}  // namespace

// One thread per VL points of the region: the border of axis D (phase 1, 2) or the border widened by R (phase 3).
// Launch geometry: x along threadIdx.x (and blockIdx.x), y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z --
// with the D extent replaced by the region's plane count -- so no thread divides anything; axis, phase and direction
// are compile-time (the first form decoded a linear index with two 64-bit divisions per thread and branched on all
// three at run time; see DESIGN.md s.4 CPML).
template <typename T, int R, int D, int PHASE, bool REV>
__global__ __launch_bounds__(256) void pml_kernel(PmlArgs<T> p, GridDesc g) {
    constexpr int VL = D == 2 ? 1 : (int)(16 / sizeof(T));
    using V = vecn<T, VL>;
    const int nd[3] = {g.nz, g.ny, g.nx};
    const int n = nd[D], npml = p.npml;
    const int nslab = min(n, 2 * npml);
    const int wide = min(n, 2 * (npml + R));  // planes of the widened region
    int e[3] = {g.nz, g.ny, D == 2 ? g.nx : g.cx};  // (x extent in padded-row terms for the vector lanes)
    e[D] = PHASE == 3 ? wide : nslab;
    int c[3];
    c[2] = VL * (blockIdx.x * blockDim.x + threadIdx.x);
    if (g.ndim == 3) {
        c[1] = blockIdx.y * blockDim.y + threadIdx.y;
        c[0] = blockIdx.z;
    } else {  // 2-D: the block's second dimension runs over z
        c[1] = 0;
        c[0] = blockIdx.z * blockDim.y + threadIdx.y;
    }
    if (c[2] >= e[2] || c[1] >= e[1] || c[0] >= e[0]) return;
    // grid coordinate along D
    const int jd = c[D];
    int i;
    if (PHASE == 3)
        i = (n <= 2 * (npml + R) || jd < npml + R) ? jd : n - 2 * (npml + R) + jd;
    else
        i = slab_to_grid(jd, n, npml);
    int gc[3] = {c[0], c[1], c[2]};
    gc[D] = i;
    const int64_t gs = D == 0 ? g.sz : D == 1 ? g.sy : 1;                              // field stride along D
    const int64_t pp = g.off0 + (int64_t)gc[0] * g.sz + (int64_t)gc[1] * g.sy + gc[2];  // padded index
    // compact aux arrays of axis D: extents (nz, ny, cx) with the D extent = nslab, x fastest
    int ae[3] = {g.nz, g.ny, g.cx};
    ae[D] = nslab;
    const int64_t astr[3] = {(int64_t)ae[1] * ae[2], ae[2], 1};
    int ac[3] = {gc[0], gc[1], gc[2]};
    ac[D] = 0;
    const int64_t abase = (int64_t)ac[0] * astr[0] + (int64_t)ac[1] * astr[1] + ac[2];  // aux offset at slab plane 0
    AxisView<T> v{D, n, npml, gs, astr[D], p.a[D], p.b[D]};
    T *psi = p.psi[D], *zet = p.zeta[D];
    const T *u = p.u_cur + pp;
    auto U = [&](int k) { return ldn<T, VL>(u + k * gs); };  // (aligned: pp, gs and k * gs keep x % VL == 0 for D != 2)

    if (PHASE == 1) {
        const int64_t ai = abase + (int64_t)jd * v.ast;
        V out = ldn<T, VL>((REV ? zet : psi) + ai);
        if (!REV) {
            V du;
#pragma unroll
            for (int q = 0; q < VL; ++q) du.v[q] = T(0);
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const V up = U(k), um = U(-k);
#pragma unroll
                for (int q = 0; q < VL; ++q) du.v[q] = fma(p.dk1[k], up.v[q] - um.v[q], du.v[q]);
            }
#pragma unroll
            for (int q = 0; q < VL; ++q) out.v[q] = fma(v.b[i], out.v[q], v.a[i] * du.v[q]);
            stn<T, VL>(psi + ai, out);
        } else {
            const V u0 = U(0);
#pragma unroll
            for (int q = 0; q < VL; ++q) out.v[q] = fma(v.b[i], out.v[q], u0.v[q]);
            stn<T, VL>(zet + ai, out);
        }
    } else if (PHASE == 2) {
        const int64_t ai = abase + (int64_t)jd * v.ast;
        if (!REV) {
            const V u0 = U(0);
            V e2, dp;
#pragma unroll
            for (int q = 0; q < VL; ++q) {
                e2.v[q] = p.ck[0] * u0.v[q];
                dp.v[q] = T(0);
            }
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const V up = U(k), um = U(-k);
                const V pp_ = aux_at<T, VL>(psi, v, abase, i + k), pm = aux_at<T, VL>(psi, v, abase, i - k);
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    e2.v[q] = fma(p.ck[k], up.v[q] + um.v[q], e2.v[q]);
                    dp.v[q] = fma(p.dk[k], pp_.v[q] - pm.v[q], dp.v[q]);
                }
            }
            V z = ldn<T, VL>(zet + ai);
#pragma unroll
            for (int q = 0; q < VL; ++q) z.v[q] = fma(v.b[i], z.v[q], v.a[i] * (e2.v[q] + dp.v[q]));
            stn<T, VL>(zet + ai, z);
        } else {
            V du, da;
#pragma unroll
            for (int q = 0; q < VL; ++q) du.v[q] = da.v[q] = T(0);
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const V up = U(k), um = U(-k);
                const int ip = i + k, im = i - k;
                const T ap = (ip < n) ? v.a[ip] : T(0), am = (im >= 0) ? v.a[im] : T(0);
                const V zp = aux_at<T, VL>(zet, v, abase, ip), zm = aux_at<T, VL>(zet, v, abase, im);
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    du.v[q] = fma(p.dk1[k], up.v[q] - um.v[q], du.v[q]);
                    da.v[q] = fma(p.dk1[k], ap * zp.v[q] - am * zm.v[q], da.v[q]);
                }
            }
            V ps = ldn<T, VL>(psi + ai);
#pragma unroll
            for (int q = 0; q < VL; ++q) ps.v[q] = v.b[i] * ps.v[q] - du.v[q] - da.v[q];
            stn<T, VL>(psi + ai, ps);
        }
    } else {
        V term;
        if (!REV) {
            // forward: phase 2 rides along (zeta' needs the same D psi' and only this point's own zeta)
            V dp;
#pragma unroll
            for (int q = 0; q < VL; ++q) dp.v[q] = T(0);
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const V pp_ = aux_at<T, VL>(psi, v, abase, i + k), pm = aux_at<T, VL>(psi, v, abase, i - k);
#pragma unroll
                for (int q = 0; q < VL; ++q) dp.v[q] = fma(p.dk[k], pp_.v[q] - pm.v[q], dp.v[q]);
            }
            term = dp;
            const int j0 = grid_to_slab(i, n, npml);
            if (j0 >= 0) {
                const V u0 = U(0);
                V e2;
#pragma unroll
                for (int q = 0; q < VL; ++q) e2.v[q] = p.ck[0] * u0.v[q];
#pragma unroll
                for (int k = 1; k <= R; ++k) {
                    const V up = U(k), um = U(-k);
#pragma unroll
                    for (int q = 0; q < VL; ++q) e2.v[q] = fma(p.ck[k], up.v[q] + um.v[q], e2.v[q]);
                }
                const int64_t ai = abase + (int64_t)j0 * v.ast;
                V z = ldn<T, VL>(zet + ai);
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    z.v[q] = fma(v.b[i], z.v[q], v.a[i] * (e2.v[q] + dp.v[q]));
                    term.v[q] += z.v[q];
                }
                stn<T, VL>(zet + ai, z);
            }
        } else {
            const T a0 = v.a[i];
            const V z0 = aux_at<T, VL>(zet, v, abase, i);
            V db;
#pragma unroll
            for (int q = 0; q < VL; ++q) {
                term.v[q] = p.ck[0] * a0 * z0.v[q];
                db.v[q] = T(0);
            }
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const int ip = i + k, im = i - k;
                const T ap = (ip < n) ? v.a[ip] : T(0), am = (im >= 0) ? v.a[im] : T(0);
                const V zp = aux_at<T, VL>(zet, v, abase, ip), zm = aux_at<T, VL>(zet, v, abase, im);
                const V qp = aux_at<T, VL>(psi, v, abase, ip), qm = aux_at<T, VL>(psi, v, abase, im);
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    term.v[q] = fma(p.ck[k], ap * zp.v[q] + am * zm.v[q], term.v[q]);
                    db.v[q] = fma(p.dk[k], ap * qp.v[q] - am * qm.v[q], db.v[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < VL; ++q) term.v[q] -= db.v[q];
        }
        const V Cc = ldn<T, VL>(p.C + pp);
        V un = ldn<T, VL>(p.u_next + pp), add;
#pragma unroll
        for (int q = 0; q < VL; ++q) {
            add.v[q] = Cc.v[q] * term.v[q];
            un.v[q] += add.v[q];
        }
        stn<T, VL>(p.u_next + pp, un);
        if (p.v) {  // increment form: the step's v' = u' - u moves with u'
            V vv = ldn<T, VL>(p.v + pp);
#pragma unroll
            for (int q = 0; q < VL; ++q) vv.v[q] += add.v[q];
            stn<T, VL>(p.v + pp, vv);
        }
        if (p.q_out) {
            T *qp = p.q_out + ((int64_t)gc[0] * g.ny + gc[1]) * g.cx + gc[2];
            V qq = ldn<T, VL>(qp);
#pragma unroll
            for (int q = 0; q < VL; ++q) qq.v[q] += add.v[q];
            stn<T, VL>(qp, qq);
        }
    }
}

// x border, fp32, r = 4, npml and nx multiples of 4: four consecutive x per thread (one 16-byte lane), the +-4
// neighbours from the two adjacent aligned vectors -- borders that are multiples of 4 wide start on vector boundaries,
// so a neighbour vector is either entirely inside the border or entirely outside it (zeros).
template <int PHASE, bool REV>
__global__ __launch_bounds__(256) void pml_kernel_x4(PmlArgs<float> p, GridDesc g) {
    constexpr int R = 4;
    using V = vecn<float, 4>;
    const int n = g.nx, npml = p.npml;
    const int nslab = 2 * npml, wide = 2 * (npml + R);
    const int ex = (PHASE == 3 ? wide : nslab) / 4;
    const int jv = blockIdx.x * blockDim.x + threadIdx.x;
    int y, z;
    if (g.ndim == 3) {
        y = blockIdx.y * blockDim.y + threadIdx.y;
        z = blockIdx.z;
    } else {
        y = 0;
        z = blockIdx.z * blockDim.y + threadIdx.y;
    }
    if (jv >= ex || y >= g.ny || z >= g.nz) return;
    const int jd = 4 * jv;
    const int half = PHASE == 3 ? npml + R : npml;
    const int i = jd < half ? jd : n - 2 * half + jd;              // grid x of this vector's first value
    const int64_t pp = g.off0 + (int64_t)z * g.sz + (int64_t)y * g.sy + i;
    const int64_t abase = ((int64_t)z * g.ny + y) * nslab;          // aux row (nz, ny, nslab)
    const float *A = p.a[2], *B = p.b[2];
    float *psi = p.psi[2], *zet = p.zeta[2];
    // 12 field values u[i-4 .. i+7] (the halo left of the row and the pad right of it hold zeros)
    auto field_win = [&](float *X) {
        const V l = ldn<float, 4>(p.u_cur + pp - 4), c = ldn<float, 4>(p.u_cur + pp), r = ldn<float, 4>(p.u_cur + pp + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { X[q] = l.v[q]; X[4 + q] = c.v[q]; X[8 + q] = r.v[q]; }
    };
    // 12 values of a memory variable at grid x = i-4 .. i+7, optionally times a(x); zeros outside the border
    auto aux_win = [&](const float *arr, bool times_a, float *X) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) {
            const int ig = i - 4 + 4 * gq;
            const int j = (ig >= 0 && ig < npml) ? ig : (ig >= n - npml && ig < n) ? ig - (n - nslab) : -1;
            V w = {{0.f, 0.f, 0.f, 0.f}};
            if (j >= 0) w = ldn<float, 4>(arr + abase + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) X[4 * gq + q] = (times_a && j >= 0) ? A[ig + q] * w.v[q] : w.v[q];
        }
    };
    const bool inb = (i >= 0 && i < npml) || (i >= n - npml && i < n);  // this vector lies in the border proper
    const int j0 = i < npml ? i : i - (n - nslab);
    if (PHASE == 1) {
        V out = ldn<float, 4>((REV ? zet : psi) + abase + j0);
        float X[12];
        field_win(X);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!REV) {
                float du = 0.f;
#pragma unroll
                for (int k = 1; k <= R; ++k) du = fmaf(p.dk1[k], X[4 + q + k] - X[4 + q - k], du);
                out.v[q] = fmaf(B[i + q], out.v[q], A[i + q] * du);
            } else {
                out.v[q] = fmaf(B[i + q], out.v[q], X[4 + q]);
            }
        }
        stn<float, 4>((REV ? zet : psi) + abase + j0, out);
    } else if (PHASE == 2) {
        float X[12], Y[12];
        field_win(X);
        aux_win(REV ? zet : psi, REV, Y);   // forward: psi'; adjoint: a zt'
        V out = ldn<float, 4>((REV ? psi : zet) + abase + j0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float du = 0.f, dy = 0.f, e2 = p.ck[0] * X[4 + q];
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                du = fmaf(p.dk1[k], X[4 + q + k] - X[4 + q - k], du);
                e2 = fmaf(p.ck[k], X[4 + q + k] + X[4 + q - k], e2);
                dy = fmaf(REV ? p.dk1[k] : p.dk[k], Y[4 + q + k] - Y[4 + q - k], dy);
            }
            out.v[q] = REV ? B[i + q] * out.v[q] - du - dy : fmaf(B[i + q], out.v[q], A[i + q] * (e2 + dy));
        }
        stn<float, 4>((REV ? psi : zet) + abase + j0, out);
    } else {
        float P[12], Z[12];
        aux_win(psi, REV, P);               // forward: psi'; adjoint: a pt'
        if (REV) aux_win(zet, true, Z);     // adjoint: a zt'
        else field_win(Z);                  // forward: u (phase 2 rides along: zeta' = b zeta + a (E u + D psi'))
        V zn = {{0.f, 0.f, 0.f, 0.f}};
        if (!REV && inb) zn = ldn<float, 4>(zet + abase + j0);
        const V Cc = ldn<float, 4>(p.C + pp);
        V un = ldn<float, 4>(p.u_next + pp), add;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dp = 0.f, e2 = p.ck[0] * Z[4 + q];
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                dp = fmaf(p.dk[k], P[4 + q + k] - P[4 + q - k], dp);
                e2 = fmaf(p.ck[k], Z[4 + q + k] + Z[4 + q - k], e2);
            }
            float term;
            if (REV) {
                term = e2 - dp;
            } else {
                if (inb) zn.v[q] = fmaf(B[i + q], zn.v[q], A[i + q] * (e2 + dp));
                term = dp + zn.v[q];
            }
            add.v[q] = Cc.v[q] * term;
            un.v[q] += add.v[q];
        }
        if (!REV && inb) stn<float, 4>(zet + abase + j0, zn);
        stn<float, 4>(p.u_next + pp, un);
        if (p.v) {
            V vv = ldn<float, 4>(p.v + pp);
#pragma unroll
            for (int q = 0; q < 4; ++q) vv.v[q] += add.v[q];
            stn<float, 4>(p.v + pp, vv);
        }
        if (p.q_out) {
            float *qp = p.q_out + ((int64_t)z * g.ny + y) * g.cx + i;
            V qq = ldn<float, 4>(qp);
#pragma unroll
            for (int q = 0; q < 4; ++q) qq.v[q] += add.v[q];
            stn<float, 4>(qp, qq);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LINE form of the z / y border (3-D): ONE launch per time step for both axes, BEFORE the step kernel.
//
// Along its own axis the border recursion is one-dimensional: psi'(i), zeta'(i) and the term of cell i need the field
// and the memory variables of the SAME (other two coordinates) line only.  So a thread owns one lane of x (two cells: 8
// bytes in fp32, 16 in fp64 -- measured best, DESIGN.md s.4 CPML) at a fixed other coordinate and marches along the
// axis through the border and the r cells its term reaches, in blocks of four cells: the field values of the line sit
// in a register window (every u is loaded once instead of 2r + 1 times), the new psi' (adjoint: alpha = a zt',
// beta = a pt') in a second one -- they are never re-read from memory, which is what forced the slab form into three
// passes -- and every memory variable is read once and written once per step, with streaming hints: at 256^3 the six
// of them (50 MB) are what pushed the wavefields out of the Infinity Cache.  The inputs of block k + 1 are in flight
// while block k is computed.  All accesses are coalesced along x.
// Off the border a = b = 0 make psi' = zeta' = 0 by themselves, so one loop body serves the low border, the high
// border and the case of the two (nearly) meeting (n < 2 npml + 3 r: one segment over the whole axis); loads are clamped
// instead of branched around, stores predicated.
//
// Round 4: the launch runs BEFORE the step kernel and HANDS THE TERM OVER instead of adding it to u' afterwards.  The
// term T_d = D psi_d' + zeta_d' (adjoint: E alpha_d - D beta_d) depends on the newest field and the memory variables
// only, so it is written to an array compact over the axis' shell (PmlArgs::tz / ty, 4 B per shell cell) and the step
// kernel forms q = C (L u + sum_d T_d + ...) in one go.  Against the round-3 form (after the step: re-read C, read-
// modify-write u' -- and v' in increment form, q in a store sweep) a shell cell costs 4 B written + 4 B read instead of
// 4 B (C) + 8 B (u') [+ 8 B (v') + 8 B (q)], the kernel has no optional operands left, and the z and the y lines share
// ONE launch (block ranges): 2 launches per time step instead of 3.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PML_LINE_MAXC = 192;  // rows of the coefficient tables (pml_line_axes checks)

template <typename T, int R, int D, bool REV, int VL>
__device__ __forceinline__ void pml_line_body(const PmlArgs<T> &p, const GridDesc &g, int bxi, int boi, int seg, T *ca,
                                              T *cb, int *cin) {
    // cells per block = the loads a thread keeps in flight (3 per cell).  More in flight was measured in round 4 and is
    // SLOWER (256^3 / npml 16 forward, us/step: this form 87.7; two blocks ahead 89.1; blocks of eight 92.9; 8-byte lanes
    // with blocks of eight, two ahead 98.0): the extra windows cost registers (up to 360 + AGPR copies), not latency
    constexpr int BS = 4;
    constexpr int W = (2 * R + BS - 1) / BS * BS;  // warm-up cells ahead of the segment (whole blocks)
    constexpr int NU = BS + 3 * R, NP = BS + 2 * R;
    using V = vecn<T, VL>;
    typedef T ntv_t __attribute__((ext_vector_type(VL > 1 ? VL : 2)));
    static_assert(D == 0 || D == 1, "z or y axis");

    const int n = D == 0 ? g.nz : g.ny, npml = p.npml;
    // (the low segment computes memory variables up to 2r rows past its end -- all zero off the border, but not if the
    // high border begins there: then one thread takes the whole axis, in order)
    const bool merged = n < 2 * npml + 3 * R;
    const int ib = (merged || seg == 0) ? 0 : n - npml - R;
    const int ie = (merged || seg == 1) ? n : npml + R;
    // First (warm-up) block: W cells ahead serve any case, but the blocks that would only form zeros are skipped when the
    // block size divides r (O(8)): the first psi' a segment needs is its border's first row (forward: formed r cells
    // ahead, adjoint alpha: 2 r), and at the high border everything before the border is zero by the table -- so the
    // forward march starts r cells ahead of the low border and AT the high segment's first cell, the adjoint 2 r / r ahead
    const bool high = !merged && seg == 1;
    const int is = (R % BS == 0) ? ib - (REV ? (high ? R : 2 * R) : (high ? 0 : R)) : ib - W;
    const int c0 = is - R;                // grid row of table entry 0
    const int nrows = (ie - is + BS - 1) / BS * BS + 3 * R + BS;
    for (int i = threadIdx.y * blockDim.x + threadIdx.x; i < nrows && i < PML_LINE_MAXC; i += blockDim.x * blockDim.y) {
        // (a segment sees its OWN border only: its last block may look a few rows into the other one, which the
        // other segment's threads are advancing at the same time)
        const int j = c0 + i;
        const bool in = j >= 0 && j < n && (merged ? (j < npml || j >= n - npml) : seg == 0 ? j < npml : j >= n - npml);
        ca[i] = in ? p.a[D][j] : T(0);
        cb[i] = in ? p.b[D][j] : T(0);
        cin[i] = in ? 1 : 0;
    }
    __syncthreads();

    const int xv = bxi * blockDim.x + threadIdx.x;
    const int o = boi * blockDim.y + threadIdx.y;  // the other coordinate: y (D = 0) or z (D = 1)
    const int no = D == 0 ? g.ny : g.nz;
    const bool act = VL * xv < g.cx && o < no;
    const int x = act ? VL * xv : 0, oc = act ? o : 0;
    const int nslab = min(n, 2 * npml);
    const int64_t gs = D == 0 ? g.sz : g.sy;                                          // field stride along the axis
    const int64_t fbase = g.off0 + (int64_t)oc * (D == 0 ? g.sy : g.sz) + x;          // padded index of row 0
    const int64_t ast = D == 0 ? (int64_t)g.ny * g.cx : g.cx;                         // memory-variable stride
    const int64_t abase = D == 0 ? (int64_t)oc * g.cx + x : (int64_t)oc * nslab * g.cx + x;
    // the term's array: shell row s of this line at tbase + s * tst
    const int nsh = pml_shell_rows(n, npml, R);
    const int64_t tst = D == 0 ? (int64_t)g.ny * g.cx : g.cx;
    const int64_t tbase = D == 0 ? (int64_t)oc * g.cx + x : (int64_t)oc * nsh * g.cx + x;
    T *const tout = D == 0 ? p.tz : p.ty;
    const int tshift = (merged || seg == 0) ? 0 : n - nsh;  // shell row of grid row i = i - tshift
    T *const m0 = REV ? p.zeta[D] : p.psi[D];   // the variable advanced first:  psi / zt
    T *const m1 = REV ? p.psi[D] : p.zeta[D];   // ... and second:               zeta / pt

    auto ldf = [&](const T *f, int j) {  // field row j, clamped into the zero halo
        return ldn<T, VL>(f + fbase + (int64_t)min(max(j, -HALO), n - 1 + HALO) * gs);
    };
    auto slab = [&](int j) {  // slab plane of grid row j (clamped: the value is multiplied by b = 0 off the border)
        const int jc = min(max(j, 0), n - 1);
        return (n <= 2 * npml || jc < npml) ? min(jc, nslab - 1) : max(jc - (n - nslab), 0);
    };
    auto ldm = [&](const T *m, int j) {
        const T *q = m + abase + (int64_t)slab(j) * ast;
        if constexpr (VL > 1) {
            const ntv_t v = __builtin_nontemporal_load(reinterpret_cast<const ntv_t *>(q));
            V r;
#pragma unroll
            for (int t = 0; t < VL; ++t) r.v[t] = v[t];
            return r;
        } else {
            V r;
            r.v[0] = __builtin_nontemporal_load(q);
            return r;
        }
    };
    auto stm = [&](T *m, int j, const V &v) {
        T *q = m + abase + (int64_t)slab(j) * ast;
        if constexpr (VL > 1) {
            ntv_t w;
#pragma unroll
            for (int t = 0; t < VL; ++t) w[t] = v.v[t];
            __builtin_nontemporal_store(w, reinterpret_cast<ntv_t *>(q));
        } else {
            __builtin_nontemporal_store(v.v[0], q);
        }
    };
    struct In {
        V un[BS];  // new field rows of the window
        V o0[BS];  // old values of the variable advanced first, at its new rows
        V o1[BS];  // ... of the second
    };
    // rows whose inputs a block at i0 needs (REV: zt sits 2r ahead like the field, pt r ahead; forward: psi r ahead,
    // zeta at the block's own rows)
    // (field rows past ie - 1 + r meet zero coefficients only -- psi', alpha vanish off the border and E reaches r rows --
    // so the window's far end, and the look-ahead of the last block, re-read that row instead of fetching planes nobody
    // uses: 8 of 32 planes per segment in O(8))
    const int jtop = ie - 1 + R;
    auto issue = [&](int i0, In &in) {
#pragma unroll
        for (int t = 0; t < BS; ++t) {
            in.un[t] = ldf(p.u_cur, min(i0 + 2 * R + t, jtop));
            in.o0[t] = ldm(m0, REV ? i0 + 2 * R + t : i0 + R + t);
            in.o1[t] = ldm(m1, REV ? i0 + R + t : i0 + t);
        }
    };

    V Uw[NU];  // forward: u rows i0 - r .. i0 + BS - 1 + 2r.   adjoint: alpha over the same rows
    V Pw[NP];  // forward: psi' rows i0 - r .. i0 + BS - 1 + r.  adjoint: beta
    V Mw[REV ? NP : 1];  // adjoint: mu rows i0 .. i0 + BS - 1 + 2r
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
        for (int t = 0; t < VL; ++t) Uw[k].v[t] = T(0);
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int t = 0; t < VL; ++t) Pw[k].v[t] = T(0);
    if (!REV) {
#pragma unroll
        for (int k = 0; k < 3 * R; ++k) Uw[k] = ldf(p.u_cur, is - R + k);
    } else {
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) Mw[k] = ldf(p.u_cur, is + k);
    }
    In nxt;
    issue(is, nxt);
    for (int i0 = is; i0 < ie; i0 += BS) {
        const In cur = nxt;
        issue(i0 + BS, nxt);
        const int tb = i0 - c0;  // table entry of row i0
        const bool out = i0 >= ib;
        if (!REV) {
#pragma unroll
            for (int t = 0; t < BS; ++t) Uw[3 * R + t] = cur.un[t];
            // psi'(j) = b psi + a D u, j = i0 + r + t
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int j = i0 + R + t;
                const T av = ca[tb + R + t], bv = cb[tb + R + t];
                V v;
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    T du = T(0);
#pragma unroll
                    for (int k = 1; k <= R; ++k) du = fma(p.dk1[k], Uw[2 * R + t + k].v[q] - Uw[2 * R + t - k].v[q], du);
                    v.v[q] = fma(bv, cur.o0[t].v[q], av * du);
                }
                Pw[2 * R + t] = v;
                if (act && cin[tb + R + t]) stm(m0, j, v);
            }
            // zeta'(i) = b zeta + a (E u + D psi');  term = D psi' + zeta'   (warm-up blocks: nothing to form)
            if (out)
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int i = i0 + t;
                const T av = ca[tb + t], bv = cb[tb + t];
                V zn, term;
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    T dp = T(0), e2 = T(0);
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        dp = fma(p.dk[k], Pw[R + t + k].v[q] - Pw[R + t - k].v[q], dp);
                        e2 = fma(p.ck[k], fma(T(-2), Uw[R + t].v[q], Uw[R + t + k].v[q] + Uw[R + t - k].v[q]), e2);
                    }
                    zn.v[q] = fma(bv, cur.o1[t].v[q], av * (e2 + dp));
                    term.v[q] = dp + zn.v[q];
                }
                const bool row = act && i < ie;
                if (row && cin[tb + t]) stm(m1, i, zn);
                if (row) stn<T, VL>(tout + tbase + (int64_t)(i - tshift) * tst, term);
            }
        } else {
            // zt'(j) = b zt + mu(j), alpha = a zt',  j = i0 + 2r + t
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int j = i0 + 2 * R + t;
                Mw[2 * R + t] = cur.un[t];
                const T av = ca[tb + 2 * R + t], bv = cb[tb + 2 * R + t];
                V zt;
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    zt.v[q] = fma(bv, cur.o0[t].v[q], cur.un[t].v[q]);
                    Uw[3 * R + t].v[q] = av * zt.v[q];
                }
                if (act && cin[tb + 2 * R + t]) stm(m0, j, zt);
            }
            // pt'(j) = b pt - D (mu + alpha), beta = a pt',  j = i0 + r + t
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int j = i0 + R + t;
                const T av = ca[tb + R + t], bv = cb[tb + R + t];
                V pt;
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    T d = T(0);
                    // (mu rows start at i0: row j - k is entry r + t - k >= 0)
#pragma unroll
                    for (int k = 1; k <= R; ++k)
                        d = fma(p.dk1[k], (Mw[R + t + k].v[q] - Mw[R + t - k].v[q]) +
                                              (Uw[2 * R + t + k].v[q] - Uw[2 * R + t - k].v[q]), d);
                    pt.v[q] = bv * cur.o1[t].v[q] - d;
                    Pw[2 * R + t].v[q] = av * pt.v[q];
                }
                if (act && cin[tb + R + t]) stm(m1, j, pt);
            }
            // term(i) = E alpha - D beta
            if (out)
#pragma unroll
            for (int t = 0; t < BS; ++t) {
                const int i = i0 + t;
                V term;
#pragma unroll
                for (int q = 0; q < VL; ++q) {
                    T e2 = T(0), db = T(0);
#pragma unroll
                    for (int k = 1; k <= R; ++k) {
                        e2 = fma(p.ck[k], fma(T(-2), Uw[R + t].v[q], Uw[R + t + k].v[q] + Uw[R + t - k].v[q]), e2);
                        db = fma(p.dk[k], Pw[R + t + k].v[q] - Pw[R + t - k].v[q], db);
                    }
                    term.v[q] = e2 - db;
                }
                if (act && i < ie) stn<T, VL>(tout + tbase + (int64_t)(i - tshift) * tst, term);
            }
#pragma unroll
            for (int k = 0; k < NP - BS; ++k) Mw[k] = Mw[k + BS];
        }
        // slide the windows by one block
#pragma unroll
        for (int k = 0; k < NU - BS; ++k) Uw[k] = Uw[k + BS];
#pragma unroll
        for (int k = 0; k < NP - BS; ++k) Pw[k] = Pw[k + BS];
    }
}

// One launch, both axes: blocks [0, nbz) march the z lines, the rest the y lines (each range: x blocks fastest, then
// the other coordinate, then the segment).  nbz = 0 / nby = 0: that axis is not taken.
template <typename T, int R, bool REV, int VL>
__global__ __launch_bounds__(256) void pml_line_t(PmlArgs<T> p, GridDesc g, int gx, int goz, int nbz, int goy) {
    __shared__ T ca[PML_LINE_MAXC], cb[PML_LINE_MAXC];
    __shared__ int cin[PML_LINE_MAXC];  // 1 = a border row this segment owns (and stores)
    int b = blockIdx.x;
    if (b < nbz) {
        const int bxi = b % gx, t = b / gx;
        pml_line_body<T, R, 0, REV, VL>(p, g, bxi, t % goz, t / goz, ca, cb, cin);
    } else {
        b -= nbz;
        const int bxi = b % gx, t = b / gx;
        pml_line_body<T, R, 1, REV, VL>(p, g, bxi, t % goy, t / goy, ca, cb, cin);
    }
}

// Axes (bit mask: z = 1, y = 2) the line form takes for this grid: both or none (the step kernels take the handed-over
// term of both axes together); the slab phases keep the rest.
int pml_line_axes(const GridDesc &g, int npml) {
    const bool off = getenv("FWI_NO_PML_LINES") != nullptr;  // (read per context: the tests switch it)
    if (off || g.ndim != 3 || npml < 1) return 0;
    const int nd[2] = {g.nz, g.ny};
    for (int d = 0; d < 2; ++d) {
        constexpr int B = 4;
        const int n = nd[d], r = g.r, W = (2 * r + B - 1) / B * B;
        const int len = n < 2 * npml + 3 * r ? n : npml + r;            // cells of a segment
        const int rows = (len + W + B - 1) / B * B + 3 * r + B;         // table rows the kernel fills
        if (rows > PML_LINE_MAXC || n < 1) return 0;
    }
    return 3;
}

// 16-byte lanes for fp64 and (round 4) fp32 O(8), 8-byte for fp32 O(2) / O(4).
template <typename T, int R, bool REV, int VL>
static void launch_pml_line_vl(const GridDesc &g, const PmlArgs<T> &p, hipStream_t s, int axes) {
    const int nxv = g.cx / VL;
    const int bx = nxv <= 16 ? 16 : nxv <= 32 ? 32 : 64, by = 256 / bx;
    const int gx = (nxv + bx - 1) / bx;
    const int goz = (g.ny + by - 1) / by, goy = (g.nz + by - 1) / by;
    const int nbz = (axes & 1) ? gx * goz * (g.nz < 2 * p.npml + 3 * R ? 1 : 2) : 0;
    const int nby = (axes & 2) ? gx * goy * (g.ny < 2 * p.npml + 3 * R ? 1 : 2) : 0;
    if (nbz + nby == 0) return;
    hipLaunchKernelGGL((pml_line_t<T, R, REV, VL>), dim3(nbz + nby), dim3(bx, by), 0, s, p, g, gx, goz, nbz, goy);
}

template <typename T, int R, bool REV>
static void launch_pml_line_rev(const GridDesc &g, const PmlArgs<T> &p, hipStream_t s, int axes) {
    // Lane width: 16 bytes for fp32 O(8) since round 4 (256^3 / npml 16, us/step forward / store / adjoint, two boxes:
    // 8-byte lanes 87.7 / 94.2 / 107.4 and 87.6 / 96.3 / 108.2, 16-byte 83.7 / 85.9 / 104.8 and 87.3 / 95.8 / 107.1 -- never
    // slower, up to 9 % faster; the round-3 form with its 5 - 7 operands per cell measured the opposite).
    // FWI_PML_LINE_VL=2 is the A/B hook.
    if constexpr (std::is_same<T, float>::value && R == 4) {
        static const bool narrow = getenv("FWI_PML_LINE_VL") && atoi(getenv("FWI_PML_LINE_VL")) == 2;
        if (!narrow) return launch_pml_line_vl<T, R, REV, 4>(g, p, s, axes);
    }
    launch_pml_line_vl<T, R, REV, 2>(g, p, s, axes);
}

template <typename T, int R>
static hipError_t launch_pml_lines_r(const GridDesc &g, const PmlArgs<T> &p, int reverse, hipStream_t s, int axes) {
    if (reverse) launch_pml_line_rev<T, R, true>(g, p, s, axes);
    else launch_pml_line_rev<T, R, false>(g, p, s, axes);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_pml_lines(const GridDesc &g, const PmlArgs<T> &p, int reverse, hipStream_t s, int axes) {
    if (((axes & 1) && !p.tz) || ((axes & 2) && !p.ty)) return hipErrorInvalidValue;
    switch (g.r) {
        case 1: return launch_pml_lines_r<T, 1>(g, p, reverse, s, axes);
        case 2: return launch_pml_lines_r<T, 2>(g, p, reverse, s, axes);
        default: return launch_pml_lines_r<T, 4>(g, p, reverse, s, axes);
    }
}
template hipError_t launch_pml_lines<float>(const GridDesc &, const PmlArgs<float> &, int, hipStream_t, int);
template hipError_t launch_pml_lines<double>(const GridDesc &, const PmlArgs<double> &, int, hipStream_t, int);

template <typename T, int R, int D, int PHASE>
static void launch_pml_axis(const GridDesc &g, const PmlArgs<T> &p, int reverse, hipStream_t s) {
    const int nd[3] = {g.nz, g.ny, g.nx};
    if constexpr (D == 2 && R == 4 && std::is_same<T, float>::value) {
        // 16-byte lanes along the x border too when everything starts on vector boundaries
        if (p.npml % 4 == 0 && g.nx % 4 == 0 && g.nx >= 2 * (p.npml + R)) {
            const int ex = (PHASE == 3 ? 2 * (p.npml + R) : 2 * p.npml) / 4;
            const int bx = ex <= 8 ? 8 : 16, by = 256 / bx;
            const dim3 block(bx, by);
            const dim3 grid((ex + bx - 1) / bx, g.ndim == 3 ? (g.ny + by - 1) / by : 1,
                            g.ndim == 3 ? g.nz : (g.nz + by - 1) / by);
            if (reverse)
                hipLaunchKernelGGL((pml_kernel_x4<PHASE, true>), grid, block, 0, s, p, g);
            else
                hipLaunchKernelGGL((pml_kernel_x4<PHASE, false>), grid, block, 0, s, p, g);
            return;
        }
    }
    constexpr int VL = D == 2 ? 1 : (int)(16 / sizeof(T));
    int e[3] = {g.nz, g.ny, D == 2 ? g.nx : g.cx / VL};  // threads along x: one per VL columns of the padded row
    e[D] = PHASE == 3 ? std::min(nd[D], 2 * (p.npml + R)) : std::min(nd[D], 2 * p.npml);
    if (e[0] <= 0 || e[1] <= 0 || e[2] <= 0) return;
    // threads along x: a 32-wide block for thin x slabs and narrow grids, 256-wide otherwise
    const int bx = e[2] <= 32 ? 32 : e[2] <= 96 ? 64 : 256, by = 256 / bx;
    const dim3 block(bx, by);
    const dim3 grid((e[2] + bx - 1) / bx, g.ndim == 3 ? (e[1] + by - 1) / by : 1, g.ndim == 3 ? e[0] : (e[0] + by - 1) / by);
    if (reverse)
        hipLaunchKernelGGL((pml_kernel<T, R, D, PHASE, true>), grid, block, 0, s, p, g);
    else
        hipLaunchKernelGGL((pml_kernel<T, R, D, PHASE, false>), grid, block, 0, s, p, g);
}

template <typename T, int R, int PHASE>
static hipError_t launch_pml_phase(const GridDesc &g, const PmlArgs<T> &p, int reverse, hipStream_t s, int axes) {
    if (axes & 1) launch_pml_axis<T, R, 0, PHASE>(g, p, reverse, s);
    if (g.ndim == 3 && (axes & 2)) launch_pml_axis<T, R, 1, PHASE>(g, p, reverse, s);
    if (axes & 4) launch_pml_axis<T, R, 2, PHASE>(g, p, reverse, s);
    return hipGetLastError();
}

template <typename T, int R>
static hipError_t launch_pml_r(const GridDesc &g, const PmlArgs<T> &p, int phase, int reverse, hipStream_t s, int axes) {
    switch (phase) {
        case 1: return launch_pml_phase<T, R, 1>(g, p, reverse, s, axes);
        case 2: return reverse ? launch_pml_phase<T, R, 2>(g, p, reverse, s, axes) : hipSuccess;  // forward: inside phase 3
        default: return launch_pml_phase<T, R, 3>(g, p, reverse, s, axes);
    }
}

template <typename T>
hipError_t launch_pml(const GridDesc &g, const PmlArgs<T> &p, int phase, int reverse, hipStream_t s, int axes) {
    switch (g.r) {
        case 1: return launch_pml_r<T, 1>(g, p, phase, reverse, s, axes);
        case 2: return launch_pml_r<T, 2>(g, p, phase, reverse, s, axes);
        default: return launch_pml_r<T, 4>(g, p, phase, reverse, s, axes);
    }
}

template hipError_t launch_pml<float>(const GridDesc &, const PmlArgs<float> &, int, int, hipStream_t, int);
template hipError_t launch_pml<double>(const GridDesc &, const PmlArgs<double> &, int, int, hipStream_t, int);

}  // namespace fwi
