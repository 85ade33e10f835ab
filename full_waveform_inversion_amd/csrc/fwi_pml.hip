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
