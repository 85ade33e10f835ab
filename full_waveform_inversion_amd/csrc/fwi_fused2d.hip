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

#include <vector>

#include <cstdlib>
#include <type_traits>

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

// Diagnostic build only (make STAMPS=1 -> libfwi_hip_stamps.so, tools/stamp_fused2d.py): s_memtime at the phase
// boundaries of every workgroup, into a buffer nothing else reads.  The production library holds none of this.
#ifdef FWI_FUSED2D_STAMPS
__device__ unsigned long long fwi_fused2d_stamps[4096 * 8];
#define FWI_STAMP(k)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                                     \
            fwi_fused2d_stamps[blockIdx.x * 8 + (k)] = (k) == 7 ? __builtin_amdgcn_s_memrealtime()     \
                                                                : __builtin_amdgcn_s_memtime();        \
    } while (0)
#else
#define FWI_STAMP(k) do { } while (0)
#endif

// Tile geometry: interior FT x FT points, KS fused steps, radius R  ->  extended edge FT + 2 HL.
// One workgroup per CU (the three LDS images take 111 KB).
//
// The region that is still exact shrinks by R cells per sub-step on every side.  Each sub-step spreads the
// ROWS of its own region densely over the threads (the sub-steps are instantiated separately, so the divisions
// are by constants): 88, 80, 72, 64 rows of 24 groups for R = 4, i.e. 3 + 2 + 2 + 2 groups per thread instead
// of the 3 + 3 + 3 + 3 of a fixed group-to-thread map, which has to cover the whole 24 x 96 image every time
// and idles on the shrinking border (forward 3.85 -> 3.64, SAVE_Q 4.6 -> 3.8, IMAGE 5.6 -> 4.5 us/step at
// 1024^2).  Whole rows only: consecutive lanes then walk consecutive 16-byte LDS words, which is bank-conflict
// free; cutting the columns to the exact region too (2 + 2 + 2 + 1 groups) made half of all LDS cycles bank
// conflicts and was slower (4.2 us/step).  A dynamic map means no per-thread state may be tied to a group, so
// C sits in LDS beside the two fields and there are no inside-the-grid masks at all: outside the grid the
// loaded C is 0, and a cell with C = 0 and u = 0 stays 0 -- the Dirichlet wall.  The imaging accumulators and
// the final stores use a second, fixed map of the FT x FT interior.  Register pressure decides everything at
// 1024 threads (128 VGPRs): the update writes each group back at once (holding a thread's groups to write
// them together spilled, and a scratch reload waits for vmcnt(0), i.e. for every load in flight).
// INC: increment form (fwi_config.update_form): the second field is v = u - u_prev instead of u_prev; v' = A (B v + q),
// u' = u + v'.  The tile then holds u twice (ping-pong: u' cannot overwrite u while neighbours still read it), v and C
// = four LDS images (148 KB); HBM traffic is the same as in the standard form (u, v, C in; u, v out).
// SKIPD: grids of more than one round of tiles (> 256) -- see the note at the damping profiles below.
// Threads per workgroup by tile edge: 1024 for the 64- and 32-point tiles (24 x 96 / 16 x 64 groups), 512 for the
// 16-point tile (12 x 48 = 576 groups).
constexpr int fused2d_threads(int ft) { return ft >= 32 ? 1024 : 512; }

template <int R, int KS, int FT, bool DAMP, bool SAVE_Q, bool IMAGE, bool INC, bool SKIPD = false>
__global__ __launch_bounds__(fused2d_threads(FT)) void step2d_fused(Fused2dArgs a, GridDesc g) {
    constexpr int FNT = fused2d_threads(FT);  // all modes fit the 128-VGPR cap of 1024 threads (68 - 102 VGPRs, no scratch)
    constexpr int HL = (KS * R + 3) / 4 * 4;  // halo cells per side (float4 aligned)
    constexpr int E = FT + 2 * HL;       // extended tile edge (rows and columns)
    constexpr int E4 = E / 4;            // float4 groups per row
    constexpr int NG = E * E4;           // groups in the extended tile
    constexpr int NI = FT * (FT / 4);    // groups in the interior
    constexpr int IPT = (NI + FNT - 1) / FNT;  // interior groups per thread
    static_assert(E % 4 == 0 && HL % 4 == 0 && FT % 4 == 0, "tile edges must be float4 aligned");
    __shared__ q4 fa[E][E4];             // field A (starts as u^n)
    __shared__ q4 fb[E][E4];             // field B (starts as u^{n-1})
    __shared__ q4 fc[E][E4];             // C = dt^2 c^2, 0 outside the grid
    __shared__ q4 fv[INC ? E : 1][E4];   // increment form: v (fb is then only the second u image)
    __shared__ float dzs[E];             // damping profiles of the tile's rows ...
    __shared__ q4 dxs4[E4];              // ... and columns (read one 16-byte group at a time: conflict-free)

    const int tid = threadIdx.x;
    FWI_STAMP(0);
    const int ntx = (g.nx + FT - 1) / FT;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), each with a private L2.  Give
    // every XCD a contiguous run of tiles (row-major: whole tile rows at 1024^2), so that the halo a tile shares
    // with its neighbours -- 1.25 of the 2.25 tile areas each workgroup fetches -- is served by that L2 instead
    // of crossing the fabric once per tile (the fill phase is fabric-bound: 6.5k of the workgroup's 23.6k cycles
    // in the stamped build before, 4.5k after; 3.39 -> 2.69 us/step at 1024^2; tools/stamp_fused2d.py).  FWI_FUSED2D_NOREMAP (tuning hook): a.xcd_remap = 0.
    int tile = blockIdx.x;
    if (a.tile_order) {  // several rounds of tiles: border tiles first (fused2d_tile_order)
        tile = a.tile_order[tile];
    } else if (a.xcd_remap) {
        const int nblk = gridDim.x, x = tile & 7, q = nblk >> 3, r = nblk & 7;  // the first r XCDs hold q + 1 tiles
        tile = x * q + min(x, r) + (tile >> 3);
    }
    // (walking 4 x 4 blocks of tiles inside the run, so that an XCD owns a 4 x 8 patch instead of a 2 x 16 strip,
    // was measured and rejected: 2.83 vs 2.69 us/step)
    const int tz = tile / ntx, tx = tile % ntx;
    const int z0 = tz * FT - HL, x0 = tx * FT - HL;  // grid coordinates of extended (0, 0)

    // ---- load: zero outside the grid.  x0 and the 4-cell pad are multiples of 4, so a group is either left of
    // the grid, or starts inside it (a group straddling the right edge reads the pad: u = 0, C = 0 there) ------
    {
        // LDS-DMA (global_load_lds_dwordx4): each wave-instruction moves 64 consecutive groups = 1 KiB straight
        // into the (unpadded, group-ordered) LDS image -- destination = wave-uniform base + lane x 16 B, source
        // address per lane -- so the tile never passes through VGPRs: no 27 x ds_write_b128 per thread (13
        // issue cycles each, a third of this kernel's LDS-array cycles in the PMC pass of the register-staged
        // form) and no 108 staging registers.  Branch-free per lane: groups outside the grid are pointed at the
        // zero halo of the padded arrays -- row -1 / row nz, column -4 / the pad right of nx -- which holds
        // exactly the zeros (u and C alike) they must see.
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        constexpr int GPT = (NG + FNT - 1) / FNT;
        const int xpad = (g.nx + 3) & ~3;
        const int wave0 = tid & ~63;  // first thread of this wave
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int g0 = wave0 + i * FNT;  // first group of this wave-instruction (wave-uniform)
            if (g0 >= NG) break;
            const int gi = tid + i * FNT;
            if (gi < NG) {  // (only an image whose size is not a multiple of 64 groups has a partial last wave)
                const int lz = gi / E4, l4 = gi % E4;
                const int zc = min(max(z0 + lz, -1), g.nz), xc = min(max(x0 + 4 * l4, -4), xpad);
                const int64_t p = g.off0 + (int64_t)zc * g.sz + xc;
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.u_cur + p), (lptr_t *)(&fa[0][0] + g0), 16, 0, 0);
                // (increment form: the "u_prev" argument is the v field)
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.u_prev + p), (lptr_t *)((INC ? &fv[0][0] : &fb[0][0]) + g0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t *)(a.C + p), (lptr_t *)(&fc[0][0] + g0), 16, 0, 0);
            }
        }
    }
    // Most tiles of a large grid lie wholly inside the absorbing border's inner edge: their damping is identically
    // zero, and the damped update (two more multiplies, a reciprocal and its Newton step per cell: ~40 % of a
    // sub-step's arithmetic) reduces to the plain one.  Workgroup-uniform, decided from the profiles just loaded.
    // Only where it pays (SKIPD): a grid of one round of tiles (1024^2 = 256 tiles = 256 CUs) ends with its slowest,
    // i.e. border, workgroups whatever the interior ones save, and the branch costs those 3.5 % (10.9 -> 11.3 us
    // per launch); from two rounds on the interior tiles' saving is the launch's: 2048^2 417 -> 442, 3000^2 425 ->
    // 459, 8192^2 426 -> 444 Gpts/s.
    int any_damping = 0;
    if (DAMP) {
        for (int i = tid; i < E; i += FNT) {
            const int z = z0 + i, x = x0 + i;
            const float dzv = (z >= 0 && z < g.nz) ? a.dz[z] : 0.f, dxv = (x >= 0 && x < g.nx) ? a.dx[x] : 0.f;
            dzs[i] = dzv;
            dxs4[i >> 2].v[i & 3] = dxv;
            any_damping |= (dzv != 0.f) | (dxv != 0.f);
        }
    }
    // fixed map of the interior (imaging accumulators, q prefetch, final stores): group ii = tid + i FNT
    q4 gacc[IPT];
    if (IMAGE) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) gacc[i] = {{0.f, 0.f, 0.f, 0.f}};
    }
    bool damped = DAMP;
    if (DAMP && SKIPD)
        damped = __syncthreads_or(any_damping) != 0;  // (the barrier that ends the fill)
    else
        __syncthreads();
    FWI_STAMP(1);

    q4(*cur)[E4] = fa;
    q4(*prv)[E4] = fb;
    const int s0 = a.inj_start ? a.inj_start[tile] : 0, s1 = a.inj_start ? a.inj_start[tile + 1] : 0;
    const int r0 = a.rec_start ? a.rec_start[tile] : 0, r1 = a.rec_start ? a.rec_start[tile + 1] : 0;

    auto substep = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const int n = a.n0 + s * a.dn;  // global time-step index of this sub-step
        // imaging stride: q^n is stored / correlated for n % istride == 0 only, in slot n / istride
        const bool qstep = (SAVE_Q || IMAGE) && (a.istride <= 1 || n % a.istride == 0);
        float *const qslot = a.q_base + (int64_t)(a.istride <= 1 ? n : n / a.istride) * g.npts;
        // the region that must still be exact after this sub-step (compile-time: the loop is unrolled)
        constexpr int lo = (s + 1) * R, hi = E - (s + 1) * R;
        // (whole rows in x: restricting the sub-step to the 22 / 20 / 18 / 16 groups per row that hold active columns
        // makes 2 + 2 + 2 + 1 trips of the 1024 threads out of 3 + 2 + 2 + 2, but rows that are not whole 16-lane passes
        // of a ds_read_b128 conflict -- 1024^2 2.72 -> 2.96 us/step -- and narrowing the last sub-step alone, whose 16
        // groups are exactly one pass, changed nothing (2.79): the sub-steps are not bound by their trip count)
        constexpr int c_lo = 0, w4 = E4, nact = (hi - lo) * w4;
        constexpr int TRIPS = (nact + FNT - 1) / FNT;  // groups per thread in this sub-step
        // imaging: start fetching this step's q for the interior groups now, use it after the update
        nt4 qv[IPT];
        if (IMAGE) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int ii = tid + i * FNT;
                const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
                const int z = z0 + lz, x = x0 + 4 * l4;
                qv[i] = nt4{0.f, 0.f, 0.f, 0.f};
                if (qstep && ii < NI && z < g.nz && x < g.nx)
                    qv[i] = __builtin_nontemporal_load(
                        reinterpret_cast<const nt4 *>(qslot + (int64_t)z * g.cx + x));
            }
        }
        // ---- stencil update of the active region --------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int gi = tid + i * FNT;
            if (gi >= nact) break;
            const int lz = lo + gi / w4, l4 = c_lo + gi % w4;
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
                const q4 zm = cur[lz - k][l4], zp = cur[lz + k][l4];  // lo >= R, hi <= E - R: in range
                const float ck = a.ck[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = (X[4 + j - k] + X[4 + j + k]) + (zm.v[j] + zp.v[j]);
                    t = fmaf(-4.f, X[4 + j], t);
                    lap.v[j] = fmaf(ck, t, lap.v[j]);
                }
            }
            const q4 up = INC ? fv[INC ? lz : 0][l4] : prv[lz][l4], Cc = fc[lz][l4];
            // (q = C L u is formed inside either branch, next to the sum it feeds: hipcc then contracts multiply and
            // add the same way as in step2d_tile, and the kernels stay bit-identical -- runs whose step count is not
            // a multiple of 4 mix them, and a checkpointed run recomputes with the one what the other stored)
            q4 q, un, vn;
            if (DAMP && (!SKIPD || damped)) {
                const q4 dxv = dxs4[l4];
                const float dzv = dzs[lz];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q.v[j] = Cc.v[j] * lap.v[j];
                    const float d = dzv + dxv.v[j];
                    if (INC) {
                        vn.v[j] = fmaf(1.f - d, up.v[j], q.v[j]) * rcp1(1.f + d);
                        un.v[j] = c.v[j] + vn.v[j];
                    } else {
                        un.v[j] = (fmaf(2.f, c.v[j], -(1.f - d) * up.v[j]) + q.v[j]) * rcp1(1.f + d);
                    }
                }
            } else {  // d = 0: the same expressions with B = A = 1
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q.v[j] = Cc.v[j] * lap.v[j];
                    if (INC) {
                        vn.v[j] = up.v[j] + q.v[j];
                        un.v[j] = c.v[j] + vn.v[j];
                    } else {
                        un.v[j] = (2.f * c.v[j] - up.v[j]) + q.v[j];
                    }
                }
            }
            if (INC) fv[INC ? lz : 0][l4] = vn;  // in place: only this thread touches v here
            prv[lz][l4] = un;  // standard form: in place over u_prev (only this thread reads it); increment form: the
                               // other u image, which nobody reads in this sub-step
            if (SAVE_Q && qstep) {
                const int z = z0 + lz, x = x0 + 4 * l4;
                const bool interior = lz >= HL && lz < HL + FT && l4 >= HL / 4 && l4 < (HL + FT) / 4;
                if (interior && z < g.nz && x < g.nx) {  // (compact rows are padded to cx: whole vector)
                    nt4 v = {q.v[0], q.v[1], q.v[2], q.v[3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<nt4 *>(qslot + (int64_t)z * g.cx + x));
                }
            }
        }
        __syncthreads();  // u^{n+1} complete in prv; (also drains this wave's q stores)
        // ---- injection into the new field (entries of this tile's extended region) -------------------
        if (s1 > s0) {
            // the host sorts a tile's entries by node: the thread of a run's first entry adds the whole run in entry
            // order -- one add per node, so duplicate nodes sum reproducibly
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
                atomicAdd(&prv[lz][lx >> 2].v[lx & 3], su);  // (one ds_add per node: no read-back latency)
                if (INC) atomicAdd(&fv[INC ? lz : 0][lx >> 2].v[lx & 3], su);  // the source moves u' and v' alike
                if (SAVE_Q && qstep && a.inj_interior[i]) atomicAdd(qslot + a.inj_cidx[i], sq);  // (no round trip)
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
            for (int i = 0; i < IPT; ++i) {
                const int ii = tid + i * FNT;
                if (ii >= NI) break;
                const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
                const q4 u = prv[lz][l4];
#pragma unroll
                for (int j = 0; j < 4; ++j) gacc[i].v[j] = fmaf(u.v[j], qv[i][j], gacc[i].v[j]);
            }
        }
        // roles swap: prv now holds the newest field
        q4(*t)[E4] = cur;
        cur = prv;
        prv = t;
    };
    static_assert(KS == 2 || KS == 4, "sub-steps are spelled out below");
    substep(std::integral_constant<int, 0>{});
    FWI_STAMP(2);
    substep(std::integral_constant<int, 1>{});
    FWI_STAMP(3);
    if constexpr (KS == 4) {
        substep(std::integral_constant<int, 2>{});
        FWI_STAMP(4);
        substep(std::integral_constant<int, 3>{});
    }
    FWI_STAMP(5);

    // ---- write the interior of the last two time levels (and the gradient contribution) -------------
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int ii = tid + i * FNT;
        if (ii >= NI) break;
        const int lz = HL + ii / (FT / 4), l4 = HL / 4 + ii % (FT / 4);
        const int z = z0 + lz, x = x0 + 4 * l4;
        if (z >= g.nz || x >= g.nx) continue;  // (interior: z, x >= 0)
        // a group straddling the right edge writes zeros into the pad columns: they are zero by construction
        const int64_t p = g.off0 + (int64_t)z * g.sz + x;
        *reinterpret_cast<q4 *>(a.out_cur + p) = cur[lz][l4];
        *reinterpret_cast<q4 *>(a.out_prev + p) = INC ? fv[INC ? lz : 0][l4] : prv[lz][l4];  // u^{n+K-1}, or v^{n+K}
        if (IMAGE) {
            float *gp = a.g + (int64_t)z * g.cx + x;
            q4 gv = *reinterpret_cast<const q4 *>(gp);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv.v[j] += gacc[i].v[j];
            *reinterpret_cast<q4 *>(gp) = gv;
        }
    }
#ifdef FWI_FUSED2D_STAMPS
    __builtin_amdgcn_s_waitcnt(0);  // the stores have left the wave
    FWI_STAMP(6);
    FWI_STAMP(7);
#endif
}

#ifdef FWI_FUSED2D_STAMPS
extern "C" int fwi_debug_fused2d_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fwi_fused2d_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

int fused2d_num_tiles(const GridDesc &g, int ft) {
    return ((g.nx + ft - 1) / ft) * ((g.nz + ft - 1) / ft);
}

void fused2d_tile_order(const GridDesc &g, int ft, int npml, std::vector<int> &order, int seam) {
    order.clear();
    static const bool off = getenv("FWI_FUSED2D_NO_TILE_ORDER") != nullptr;  // tuning / A-B hook
    const int ntx = (g.nx + ft - 1) / ft, ntz = (g.nz + ft - 1) / ft, nt = ntx * ntz;
    if (off || nt <= 256 || npml <= 0) return;
    const int HL = (FUSED2D_STEPS * g.r + 3) / 4 * 4, E = ft + 2 * HL;
    auto heavy = [&](int t, int n) {  // the tile's extended region [t ft - HL, t ft - HL + E) meets a border of the axis
        const int lo = fused2d_origin(t, n, ft, seam) - HL;
        return (lo < npml || lo + E > n - npml) ? 1 : 0;
    };
    std::vector<int> first[2], rest;  // corners, edges, interior (row-major)
    for (int tz = 0; tz < ntz; ++tz)
        for (int tx = 0; tx < ntx; ++tx) {
            const int w = heavy(tz, g.nz) + heavy(tx, g.nx);
            (w == 2 ? first[0] : w == 1 ? first[1] : rest).push_back(tz * ntx + tx);
        }
    order.assign(nt, 0);
    int b = 0;
    for (auto &v : first)
        for (int t : v) order[b++] = t;
    // workgroup b runs on XCD b % 8: the interior tiles in eight contiguous runs, one per XCD
    const int nh = b;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, start[8], pos[8];
    for (int i = nh; i < nt; ++i) ++cnt[i & 7];
    for (int x = 0, acc = 0; x < 8; ++x) {
        start[x] = acc;
        pos[x] = 0;
        acc += cnt[x];
    }
    for (int i = nh; i < nt; ++i) order[i] = rest[start[i & 7] + pos[i & 7]++];
}

int fused2d_pick_tile(const GridDesc &g) {
    if (const char *e = getenv("FWI_FUSED2D_TILE")) {
        const int v = atoi(e);
        if (v == 64 || v == 32 || v == 16) return v;
    }
    // cost of a launch ~ rounds of 256 workgroups x extended tile area (the halo is 4 r = 16 cells whatever the tile:
    // a 32-point tile computes 4x its own area, a 16-point tile 9x -- worth it only while CUs would idle otherwise)
    const int HL = (FUSED2D_STEPS * g.r + 3) / 4 * 4;
    int best = FUSED2D_TILE;
    int64_t best_cost = -1;
    for (int ft : {64, 32, 16}) {
        const int64_t tiles = fused2d_num_tiles(g, ft), e = ft + 2 * HL;
        const int64_t cost = ((tiles + 255) / 256) * e * e;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = ft;
        }
    }
    return best;
}

template <int R, int FT, bool DAMP, bool SKIPD>
static hipError_t launch_fused_ft(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    constexpr int KS = FUSED2D_STEPS;
    const dim3 grid(fused2d_num_tiles(g, FT));
    if (a.inc) {
        if (a.mode == 1)
            hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, true, false, true, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
        else if (a.mode == 2)
            hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, true, true, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
        else
            hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, false, true, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
    } else if (a.mode == 1)
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, true, false, false, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
    else if (a.mode == 2)
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, true, false, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
    else
        hipLaunchKernelGGL((step2d_fused<R, KS, FT, DAMP, false, false, false, SKIPD>), grid, dim3(fused2d_threads(FT)), 0, s, a, g);
    return hipGetLastError();
}

template <int R, bool DAMP, bool SKIPD>
static hipError_t launch_fused_r(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    // (the increment form holds four LDS images: 64-point tiles only, the host never asks for less)
    if (a.ft == 32 && !a.inc) return launch_fused_ft<R, 32, DAMP, SKIPD>(g, a, s);
    if (a.ft == 16 && !a.inc) return launch_fused_ft<R, 16, DAMP, SKIPD>(g, a, s);
    return launch_fused_ft<R, 64, DAMP, SKIPD>(g, a, s);
}

template <int R>
static hipError_t launch_fused_d(const GridDesc &g, const Fused2dArgs &a, hipStream_t s) {
    if (!a.damp) return launch_fused_r<R, false, false>(g, a, s);
    // more than one round of workgroups: interior tiles skip the damped update (FWI_FUSED2D_SKIPD=0 / 1 forces)
    const bool skip = a.skipd >= 0 ? a.skipd != 0 : fused2d_num_tiles(g, a.ft ? a.ft : FUSED2D_TILE) > 256;
    return skip ? launch_fused_r<R, true, true>(g, a, s) : launch_fused_r<R, true, false>(g, a, s);
}

hipError_t launch_fused2d(const GridDesc &g, const Fused2dArgs &a0, hipStream_t s) {
    static const bool no_remap = getenv("FWI_FUSED2D_NOREMAP") != nullptr;  // tuning hook
    Fused2dArgs a = a0;
    a.xcd_remap = no_remap ? 0 : 1;
    if (g.r == 4) return launch_fused_d<4>(g, a, s);
    if (g.r == 2) return launch_fused_d<2>(g, a, s);
    return launch_fused_d<1>(g, a, s);
}

}  // namespace fwi
