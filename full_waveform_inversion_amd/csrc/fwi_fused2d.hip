// 2-D temporal blocking: KS consecutive time steps per launch on an LDS-resident tile.
//
// A 2-D step kernel at BASELINE size (1024^2) is bounded by the ~1.5 us kernel boundary plus one
// dependent load -> barrier -> compute -> store chain (7 us per step measured), not by
// bandwidth.  Here a workgroup loads its TZ x TX tile plus a KS*r-cell halo of u^n, u^{n-1} and C
// once, advances it KS steps entirely in LDS (the region that is still exact shrinks by r cells
// per step: overlapped / trapezoidal tiling, halo work is redundant between neighbours), and
// writes the tile of the last two time levels.  Per step this amortises the boundary and the
// global round trip over KS steps at the price of (1 + 2 KS r / T)^2 more arithmetic.
//
// Source / residual injection, receiver sampling, the forward imaging term q (SAVE_Q) and the
// imaging condition (IMAGE) all happen inside the sub-steps, so -- unlike the single-step kernels --
// nothing is lagged: step n samples u^{n+1} and pairs mu^{n+1} with q^n directly, and the gradient
// accumulator is read-modified-written once per launch.
//
// No reference counterpart; arithmetic identical to step2d_tile / the oracle.
#include <hip/hip_runtime.h>

#include "fwi_kernels.h"

namespace fwi {

namespace {

struct alignas(16) q4 {
    float v[4];
};
typedef float nt4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp1(float a) {
    const float r = __builtin_amdgcn_rcpf(a);
    return r * fmaf(-a, r, 2.f);
}

}  // namespace

// Tile geometry: interior FT x FT points, KS fused steps, radius R  ->  extended edge FT + 2 KS R.
// Threads per workgroup (one workgroup per CU: the two LDS images take 74 KB).  PMC profile at 512:
// 57 % of the wave cycles are waits (LDS round trips, three barriers per sub-step, the initial
// global loads), the rest ~2400 instructions per wave.  Measured at 1024^2, us/step: 256 threads
// 5.5, 512 4.1, 1024 3.85 (forward); the imaging variant spills at the 128-VGPR cap of 1024
// threads and stays at 512 (5.5 vs 5.8).
template <int R, int KS, int FT, bool DAMP, bool SAVE_Q, bool IMAGE>
__global__ __launch_bounds__(IMAGE ? 512 : 1024) void step2d_fused(Fused2dArgs a, GridDesc g) {
    constexpr int FNT = IMAGE ? 512 : 1024;
    constexpr int HL = (KS * R + 3) / 4 * 4;  // halo cells per side (float4 aligned)
    constexpr int E = FT + 2 * HL;       // extended tile edge (rows and columns)
    constexpr int E4 = E / 4;            // float4 groups per row
    constexpr int NG = E * E4;           // groups in the extended tile
    constexpr int GPT = (NG + FNT - 1) / FNT;  // groups per thread
    static_assert(E % 4 == 0 && HL % 4 == 0, "tile edges must be float4 aligned");
    __shared__ q4 fa[E][E4];             // field A (starts as u^n)
    __shared__ q4 fb[E][E4];             // field B (starts as u^{n-1})
    __shared__ float dzs[E], dxs[E];     // damping profiles of the tile's rows / columns

    const int tid = threadIdx.x;
    const int ntx = (g.nx + FT - 1) / FT;
    const int tz = blockIdx.x / ntx, tx = blockIdx.x % ntx;
    const int z0 = tz * FT - HL, x0 = tx * FT - HL;  // grid coordinates of extended (0, 0)

    // ---- load: zero outside the grid (the padded arrays only carry 4 zero cells) -------------------
    q4 Cg[GPT];
    unsigned inside[GPT];  // 4-bit mask: which of the group's points lie inside the grid
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int gi = tid + i * FNT;
        const int lz = gi / E4, l4 = gi % E4;
        const int z = z0 + lz, x = x0 + 4 * l4;
        q4 va = {{0.f, 0.f, 0.f, 0.f}}, vb = va, vc = va;
        unsigned m = 0;
        if (gi < NG && z >= 0 && z < g.nz && x + 3 >= 0 && x < g.nx) {
            // groups are 16 B aligned in the padded layout (x0 and the 4-cell pad are multiples of 4);
            // a group straddling the right edge reads pad zeros / is masked element-wise
            const int64_t p = g.off0 + (int64_t)z * g.sz + x;
            if (x >= 0 && x + 3 < g.nx) {
                va = *reinterpret_cast<const q4 *>(a.u_cur + p);
                vb = *reinterpret_cast<const q4 *>(a.u_prev + p);
                vc = *reinterpret_cast<const q4 *>(a.C + p);
                m = 0xF;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j >= 0 && x + j < g.nx) {
                        va.v[j] = a.u_cur[p + j];
                        vb.v[j] = a.u_prev[p + j];
                        vc.v[j] = a.C[p + j];
                        m |= 1u << j;
                    }
            }
        }
        if (gi < NG) {
            fa[lz][l4] = va;
            fb[lz][l4] = vb;
        }
        Cg[i] = vc;
        inside[i] = m;
    }
    if (DAMP) {
        for (int i = tid; i < E; i += FNT) {
            const int z = z0 + i, x = x0 + i;
            dzs[i] = (z >= 0 && z < g.nz) ? a.dz[z] : 0.f;
            dxs[i] = (x >= 0 && x < g.nx) ? a.dx[x] : 0.f;
        }
    }
    q4 gacc[GPT];
    if (IMAGE) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) gacc[i] = {{0.f, 0.f, 0.f, 0.f}};
    }
    __syncthreads();

    q4(*cur)[E4] = fa;
    q4(*prv)[E4] = fb;
    const int s0 = a.inj_start ? a.inj_start[blockIdx.x] : 0, s1 = a.inj_start ? a.inj_start[blockIdx.x + 1] : 0;
    const int r0 = a.rec_start ? a.rec_start[blockIdx.x] : 0, r1 = a.rec_start ? a.rec_start[blockIdx.x + 1] : 0;

    for (int s = 0; s < KS; ++s) {
        const int n = a.n0 + s * a.dn;  // global time-step index of this sub-step
        // imaging stride: q^n is stored / correlated for n % istride == 0 only, in slot n / istride
        const bool qstep = (SAVE_Q || IMAGE) && (a.istride <= 1 || n % a.istride == 0);
        float *const qslot = a.q_base + (int64_t)(a.istride <= 1 ? n : n / a.istride) * g.npts;
        // Only the rows that must still be exact after this sub-step are updated: the exact region
        // shrinks by R per sub-step towards the interior (trapezoid).  Rows only: the test is (nearly)
        // wave-uniform; also cutting columns diverges lanes and measured slower (4.8 vs 4.1 us/step).
        const int lo = (s + 1) * R, hi = E - (s + 1) * R;
        // imaging: start fetching this step's q for the interior groups now, use it after the update
        nt4 qv[GPT];
        if (IMAGE) {
#pragma unroll
            for (int i = 0; i < GPT; ++i) {
                const int gi = tid + i * FNT;
                const int lz = gi / E4, l4 = gi % E4;
                const bool interior = gi < NG && lz >= HL && lz < HL + FT && l4 >= HL / 4 && l4 < (HL + FT) / 4;
                qv[i] = nt4{0.f, 0.f, 0.f, 0.f};
                if (qstep && interior && inside[i] == 0xF)
                    qv[i] = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(
                        qslot + (int64_t)(z0 + lz) * g.cx + (x0 + 4 * l4)));
            }
        }
        // ---- stencil update ---------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int gi = tid + i * FNT;
            if (gi >= NG) break;
            const int lz = gi / E4, l4 = gi % E4;
            if (lz < lo || lz >= hi) continue;
            const q4 c = cur[lz][l4];
            const q4 xl = cur[lz][max(l4 - 1, 0)], xr = cur[lz][min(l4 + 1, E4 - 1)];
            float X[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                X[j] = xl.v[j];
                X[4 + j] = c.v[j];
                X[8 + j] = xr.v[j];
            }
            q4 lap = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k = R; k >= 1; --k) {
                const q4 zm = cur[max(lz - k, 0)][l4], zp = cur[min(lz + k, E - 1)][l4];
                const float ck = a.ck[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = (X[4 + j - k] + X[4 + j + k]) + (zm.v[j] + zp.v[j]);
                    t = fmaf(-4.f, X[4 + j], t);
                    lap.v[j] = fmaf(ck, t, lap.v[j]);
                }
            }
            const q4 up = prv[lz][l4];
            q4 q, un;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q.v[j] = Cg[i].v[j] * lap.v[j];
                if (DAMP) {
                    const float d = dzs[lz] + dxs[4 * l4 + j];
                    un.v[j] = (fmaf(2.f, c.v[j], -(1.f - d) * up.v[j]) + q.v[j]) * rcp1(1.f + d);
                } else {
                    un.v[j] = (2.f * c.v[j] - up.v[j]) + q.v[j];
                }
                if (!((inside[i] >> j) & 1u)) un.v[j] = 0.f;  // Dirichlet: zero outside the grid
            }
            prv[lz][l4] = un;  // in place: only this thread reads prv[lz][l4]
            if (SAVE_Q && qstep) {
                const bool interior = lz >= HL && lz < HL + FT && l4 >= HL / 4 && l4 < (HL + FT) / 4;
                if (interior && inside[i]) {
                    float *qp = qslot + (int64_t)(z0 + lz) * g.cx + (x0 + 4 * l4);
                    if (inside[i] == 0xF) {
                        nt4 v = {q.v[0], q.v[1], q.v[2], q.v[3]};
                        __builtin_nontemporal_store(v, reinterpret_cast<nt4 *>(qp));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((inside[i] >> j) & 1u) qp[j] = q.v[j];
                    }
                }
            }
        }
        __syncthreads();  // u^{n+1} complete in prv; (also drains this wave's q stores)
        // ---- injection into the new field (entries of this tile's extended region) -------------------
        if (s1 > s0) {
            for (int i = s0 + tid; i < s1; i += FNT) {
                const float amp = a.inj_amp[(int64_t)n * a.ninj + a.inj_col[i]];
                const int lz = a.inj_lz[i], lx = a.inj_lx[i];
                atomicAdd(&prv[lz][lx >> 2].v[lx & 3], a.inj_cu[i] * amp);
                if (SAVE_Q && qstep && a.inj_interior[i])
                    atomicAdd(qslot + a.inj_cidx[i], a.inj_cq[i] * amp);
            }
            __syncthreads();
        }
        // ---- receiver sampling of the new field (entries inside this tile's interior) ----------------
        for (int i = r0 + tid; i < r1; i += FNT) {
            const int lz = a.rec_lz[i], lx = a.rec_lx[i];
            a.rec_out[(int64_t)n * a.nrec + a.rec_col[i]] = prv[lz][lx >> 2].v[lx & 3] * a.rec_scale;
        }
        // ---- imaging: g += mu^{n+1} * q^n on the interior ---------------------------------------------
        if (IMAGE && qstep) {
#pragma unroll
            for (int i = 0; i < GPT; ++i) {
                const int gi = tid + i * FNT;
                if (gi >= NG) break;
                const int lz = gi / E4, l4 = gi % E4;
                const bool interior = lz >= HL && lz < HL + FT && l4 >= HL / 4 && l4 < (HL + FT) / 4;
                if (interior && inside[i]) {
                    const float *qp = qslot + (int64_t)(z0 + lz) * g.cx + (x0 + 4 * l4);
                    const q4 u = prv[lz][l4];
                    if (inside[i] == 0xF) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) gacc[i].v[j] = fmaf(u.v[j], qv[i][j], gacc[i].v[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((inside[i] >> j) & 1u) gacc[i].v[j] = fmaf(u.v[j], qp[j], gacc[i].v[j]);
                    }
                }
            }
        }
        // roles swap: prv now holds the newest field
        q4(*t)[E4] = cur;
        cur = prv;
        prv = t;
    }

    // ---- write the interior of the last two time levels (and the gradient contribution) -------------
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int gi = tid + i * FNT;
        if (gi >= NG) break;
        const int lz = gi / E4, l4 = gi % E4;
        const bool interior = lz >= HL && lz < HL + FT && l4 >= HL / 4 && l4 < (HL + FT) / 4;
        if (!interior || !inside[i]) continue;
        const int z = z0 + lz, x = x0 + 4 * l4;
        const int64_t p = g.off0 + (int64_t)z * g.sz + x;
        const q4 vc = cur[lz][l4], vp = prv[lz][l4];
        if (inside[i] == 0xF) {
            *reinterpret_cast<q4 *>(a.out_cur + p) = vc;
            *reinterpret_cast<q4 *>(a.out_prev + p) = vp;
            if (IMAGE) {
                float *gp = a.g + (int64_t)z * g.cx + x;
                q4 gv = *reinterpret_cast<const q4 *>(gp);
#pragma unroll
                for (int j = 0; j < 4; ++j) gv.v[j] += gacc[i].v[j];
                *reinterpret_cast<q4 *>(gp) = gv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((inside[i] >> j) & 1u) {
                    a.out_cur[p + j] = vc.v[j];
                    a.out_prev[p + j] = vp.v[j];
                    if (IMAGE) a.g[(int64_t)z * g.cx + x + j] += gacc[i].v[j];
                }
        }
    }
}

int fused2d_num_tiles(const GridDesc &g) {
    return ((g.nx + FUSED2D_TILE - 1) / FUSED2D_TILE) * ((g.nz + FUSED2D_TILE - 1) / FUSED2D_TILE);
}

template <int R, bool DAMP>
static hipError_t launch_fused_r(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    constexpr int KS = FUSED2D_STEPS, FT = FUSED2D_TILE;
    const dim3 grid(fused2d_num_tiles(g));
    if (a.mode == 1)
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, true, false>), grid, dim3(1024), 0, s, a, g);
    else if (a.mode == 2)
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, true>), grid, dim3(512), 0, s, a, g);
    else
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, false>), grid, dim3(1024), 0, s, a, g);
    return hipGetLastError();
}

hipError_t launch_fused2d(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    if (g.r == 4) return a.damp ? launch_fused_r<4, true>(g, a, s) : launch_fused_r<4, false>(g, a, s);
    if (g.r == 2) return a.damp ? launch_fused_r<2, true>(g, a, s) : launch_fused_r<2, false>(g, a, s);
    return a.damp ? launch_fused_r<1, true>(g, a, s) : launch_fused_r<1, false>(g, a, s);
}

}  // namespace fwi
