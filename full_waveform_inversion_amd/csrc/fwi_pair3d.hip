// 3-D temporal blocking: TWO time steps per pass over the grid (fp32).
//
// The single-step stream kernel moves its algorithmic 16 B/update and sits at the fabric / HBM rate, so the only
// lever left is fewer bytes: advancing u^n -> u^{n+2} in one pass reads u^n, u^{n-1}, C and writes u^{n+1},
// u^{n+2} = 20 B per TWO updates (10 B/update).  Price: the first step has to be computed on a halo of r cells
// around every tile (overlapped tiling) and the on-chip state doubles.
//
// Workgroup = 8 waves; it owns TYI = 8 rows (y) x one x tile and marches z.  A wave = one row of 64 lanes x
// float4.  Per plane iteration s:
//   stage 1  u^{n+1}(plane s) on the tile's rows widened by r on each side (16 rows for r = 4: every wave one
//            interior row + one halo row).  z neighbours of u^n: register queue per row; y neighbours: the
//            plane's 8 + 4r rows staged in LDS (double buffered); x neighbours: the same LDS row.
//            The result goes into an LDS ring of r + 1 planes (all rows) and, for the interior row, into a
//            second register queue (and to HBM: u^{n+1} is an output).
//   stage 2  u^{n+2}(plane s - r) on the interior rows: z neighbours from that second queue, y / x neighbours
//            from the ring slot of plane s - r, u^n(s - r) from the tail of the first queue, C from a register
//            ring.  One barrier per iteration.
// Points outside the grid are loaded from the zero halo of the padded arrays (clamped row / plane / column
// indices) and stay exactly zero through stage 1 (C = 0 there): no masks in the arithmetic.
// x: a grid row of <= 256 columns is one tile without any x halo (the row ends are the grid boundary: zero edge
// vectors in LDS).  Wider grids are cut into tiles of `tw` interior columns + 2 x 8 halo columns.
// Source injection happens inside the pass (the source of step 1 must be in u^{n+1} before step 2 uses it) from a
// per-workgroup entry list; receiver sampling of both new fields rides on the next launch.
//
// No reference counterpart; arithmetic identical to step3d_stream (difference-form Laplacian).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "fwi_kernels.h"

namespace fwi {

namespace {

struct alignas(16) f4 {
    float v[4];
};
__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void st4(float *p, const f4 &v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ float rcp_nr(float a) {
    const float r = __builtin_amdgcn_rcpf(a);
    return r * fmaf(-a, r, 2.f);
}

}  // namespace

template <int R, bool DAMP>
__global__ __launch_bounds__(64 * PAIR3D_TY) void step3d_pair(Pair3dArgs a, GridDesc g, int zchunk, int nxt, int nyt,
                                                              int nblk, int tw) {
    constexpr int TYI = PAIR3D_TY;          // interior rows per workgroup = waves per workgroup
    constexpr int NQ = 2 * R + 2;           // z-queue slots: planes s-R .. s+R of u^n plus the one in flight
    constexpr int NRING = R + 1;            // planes of u^{n+1} kept in LDS (s-R .. s)
    constexpr int SROWS = TYI + 4 * R;      // staged rows of u^n (+ one trash row behind them)
    constexpr int RROWS = TYI + 2 * R;      // rows of u^{n+1}    (+ one trash row)
    constexpr int LROW = 66;                // LDS row in float4: [zero edge][64 lanes][zero edge]
    static_assert(NQ % NRING == 0 && NQ % 2 == 0, "ring indices must be static under the NQ-fold unrolling");
    // The loop body is free of divergent branches (hipcc then counts its s_waitcnt vmcnt(N) exactly and keeps the
    // next plane's loads in flight across the barrier): waves without a halo row (r < 4) repeat their interior row
    // and write to the trash rows instead.
    __shared__ f4 stg[2][SROWS + 1][LROW];
    __shared__ f4 ring[NRING][RROWS + 1][LROW];

    const int lane = threadIdx.x, w = threadIdx.y;
    int bid = blockIdx.x;
    if (bid >= nblk) {
        // Receiver sampling of the two fields the previous launch produced (read-only here): u_prev = the older.
        const int nth = 64 * TYI;
        for (int i = (bid - nblk) * nth + w * 64 + lane; i < a.nrec; i += (gridDim.x - nblk) * nth) {
            if (a.rec_out0) a.rec_out0[i] = a.u_prev[a.rec_pidx[i]] * a.rec_scale;
            if (a.rec_out1) a.rec_out1[i] = a.u_cur[a.rec_pidx[i]] * a.rec_scale;
        }
        return;
    }
    {  // XCD-contiguous runs of workgroups (as in step3d_stream): neighbours in y share the XCD's L2
        const int x = bid & 7, q = nblk >> 3, r = nblk & 7;
        bid = x * q + min(x, r) + (bid >> 3);
    }
    const int bx = bid % nxt;
    const int t2 = bid / nxt;
    const int by = t2 % nyt, bz = t2 / nyt;
    const bool fullrow = nxt == 1;          // the row is one tile: its ends are the grid boundary
    const int xi0 = bx * tw;                // first interior column of the tile
    const int xg = (fullrow ? 0 : xi0 - 2 * HALO) + 4 * lane;  // this lane's first column
    const int xend = fullrow ? g.nx : min(xi0 + tw, g.nx);
    const bool in_x = xg >= (fullrow ? 0 : xi0) && xg < xend;   // lane holds interior columns
    const int xpad = (g.nx + 3) & ~3;
    const int xc = min(max(xg, -HALO), xpad);  // clamped: the zero halo left of the row / the zero pad right of it
    const int y0 = by * TYI;
    const int z0 = bz * zchunk, z1 = min(g.nz, z0 + zchunk);
    const int64_t sz = g.sz, sy = g.sy;

    // rows of this wave: interior (offset w), halo of step 1 (only the first 2R waves), outer halo of u^n
    const int offI = w;
    const bool hasH = w < 2 * R;
    const int offH = (w < R) ? w - R : TYI + (w - R);
    const int offE = (w < R) ? w - 2 * R : TYI + R + (w - R);
    auto rowoff = [&](int off) -> int64_t {  // in-plane offset of (row y0 + off, column xc), row clamped to the halo
        const int y = min(max(y0 + off, -1), g.ny);
        return g.off0 + (int64_t)y * sy + xc;
    };
    const int64_t pI = rowoff(offI), pH = rowoff(hasH ? offH : offI), pE = rowoff(hasH ? offE : offI);
    const int srI = offI + 2 * R, srH = hasH ? offH + 2 * R : SROWS, srE = hasH ? offE + 2 * R : SROWS;  // staging rows
    const int rrI = offI + R, rrH = hasH ? offH + R : RROWS;                                              // ring rows
    const int lrH = hasH ? srH : srI;  // the row a halo-less wave recomputes (its own interior row, result discarded)
    const bool storeI = in_x && (y0 + offI) < g.ny;
    auto plane = [&](int p) -> int64_t { return (int64_t)min(max(p, -1), g.nz) * sz; };  // clamped to a zero plane

    // zero edge vectors of every LDS row (never written afterwards)
    for (int i = w * 64 + lane; i < 2 * (SROWS + 1) * 2 + NRING * (RROWS + 1) * 2; i += 64 * TYI) {
        const f4 zero = {{0.f, 0.f, 0.f, 0.f}};
        if (i < 2 * (SROWS + 1) * 2) {
            stg[i / ((SROWS + 1) * 2)][(i / 2) % (SROWS + 1)][(i & 1) ? LROW - 1 : 0] = zero;
        } else {
            const int k = i - 2 * (SROWS + 1) * 2;
            ring[k / ((RROWS + 1) * 2)][(k / 2) % (RROWS + 1)][(k & 1) ? LROW - 1 : 0] = zero;
        }
    }

    // xy part of the damping, per row
    f4 dxyI, dxyH;
    if (DAMP) {
        const int yI = y0 + offI, yH = y0 + offH;
        const float dyI = (yI >= 0 && yI < g.ny) ? a.dy[yI] : 0.f;
        const float dyH = (hasH && yH >= 0 && yH < g.ny) ? a.dy[yH] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dx = (xg + j >= 0 && xg + j < g.nx) ? a.dx[xg + j] : 0.f;
            dxyI.v[j] = dyI + dx;
            dxyH.v[j] = dyH + dx;
        }
    }
    auto dz_of = [&](int z) -> float {
        const int dist = max(0, max(a.npml - z, z - (g.nz - 1 - a.npml)));
        return a.dz_scale * (float)(dist * dist);
    };

    // injection entries of this workgroup (usually none)
    const int e0 = a.inj_start ? a.inj_start[bid] : 0, e1 = a.inj_start ? a.inj_start[bid + 1] : 0;

    // ---- prologue: u^n planes s0-R .. s0+R (s0 = z0 - R), operands of plane s0 ------------------------------
    const int s0 = z0 - R, s1 = z1 + R;
    f4 qnI[NQ], qnH[NQ], q1I[NQ];
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) {
        const int64_t o = plane(s0 - R + k);
        qnI[k] = ld4(a.u_cur + o + pI);
        qnH[k] = ld4(a.u_cur + o + pH);
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) q1I[k] = f4{{0.f, 0.f, 0.f, 0.f}};
    // (LDS-DMA for the outer halo rows, which are only ever staged, was tried: with a global_load_lds in flight hipcc
    // drains every outstanding load at the barrier -- 55 instead of 40 us per step)
    f4 e2[2], upI[2], upH[2], cH[2], cring[NRING], cnext;
    {
        const int64_t o = plane(s0);
        e2[0] = ld4(a.u_cur + o + pE);
        upI[0] = ld4(a.u_prev + o + pI);
        upH[0] = ld4(a.u_prev + o + pH);
        cH[0] = ld4(a.C + o + pH);
        cring[0] = ld4(a.C + o + pI);
    }
    __syncthreads();  // the zero edges are in place

    // one row of step 1 or step 2: Laplacian of `ctr` with z neighbours zq(k), y / x neighbours from LDS rows
    auto laplace = [&](const f4 &ctr, f4 (*L)[LROW], int lrow, auto zq) __attribute__((always_inline)) -> f4 {
        const f4 xl = L[lrow][lane], xr = L[lrow][lane + 2];
        float X[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            X[j] = xl.v[j];
            X[4 + j] = ctr.v[j];
            X[8 + j] = xr.v[j];
        }
        f4 lap = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = R; k >= 1; --k) {
            const f4 ym = L[lrow - k][1 + lane], yp = L[lrow + k][1 + lane];
            const f4 zm = zq(-k), zp = zq(k);
            const float c = a.ck[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = (X[4 + j - k] + X[4 + j + k]) + (ym.v[j] + yp.v[j]) + (zm.v[j] + zp.v[j]);
                t = fmaf(-6.f, X[4 + j], t);
                lap.v[j] = fmaf(c, t, lap.v[j]);
            }
        }
        return lap;
    };
    auto update = [&](const f4 &ctr, const f4 &up, const f4 &Cc, const f4 &lap, const f4 &dxy, float dzv) __attribute__((always_inline)) -> f4 {
        f4 un;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q = Cc.v[j] * lap.v[j];
            if (DAMP) {
                const float d = dxy.v[j] + dzv;
                un.v[j] = (fmaf(2.f, ctr.v[j], -(1.f - d) * up.v[j]) + q) * rcp_nr(1.f + d);
            } else {
                un.v[j] = (2.f * ctr.v[j] - up.v[j]) + q;
            }
        }
        return un;
    };
    // add the source terms of `step` (0 / 1) that fall on (plane z, row offset off) into v
    auto inject = [&](f4 &v, int z, int off, int step) __attribute__((always_inline)) {
        for (int e = e0; e < e1; ++e) {
            const Pair3dInj en = a.inj[e];
            if (en.z == z && en.yoff == off && (step == 0 || en.interior)) {
                const float amp = (step ? a.inj_amp1 : a.inj_amp0)[en.col] * en.cu;
                const int jj = ((en.xoff >> 2) == lane) ? (en.xoff & 3) : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) v.v[j] += (j == jj) ? amp : 0.f;  // (no dynamic index: v stays in registers)
            }
        }
    };

    for (int sb = s0; sb < s1; sb += NQ) {
#pragma unroll
        for (int ph = 0; ph < NQ; ++ph) {
            const int s = sb + ph;
            if (s >= s1) break;
            const int cur = ph & 1, nxt = cur ^ 1;
            f4(*S)[LROW] = stg[cur];
            // ---- stage the u^n rows of plane s --------------------------------------------------------------
            const f4 ctrI = qnI[(ph + R) % NQ], ctrH = qnH[(ph + R) % NQ];
            S[srI][1 + lane] = ctrI;
            S[srH][1 + lane] = ctrH;
            S[srE][1 + lane] = e2[cur];
            // ---- prefetch plane s + 1 (queue: plane s + R + 1) ------------------------------------------------
            {
                const int64_t oq = plane(s + R + 1), on = plane(s + 1);
                qnI[(ph + 2 * R + 1) % NQ] = ld4(a.u_cur + oq + pI);
                qnH[(ph + 2 * R + 1) % NQ] = ld4(a.u_cur + oq + pH);
                e2[nxt] = ld4(a.u_cur + on + pE);
                upI[nxt] = ld4(a.u_prev + on + pI);
                upH[nxt] = ld4(a.u_prev + on + pH);
                cH[nxt] = ld4(a.C + on + pH);
                cnext = ld4(a.C + on + pI);
            }
            __syncthreads();
            // ---- stage 1: u^{n+1}(s) on the interior row and the halo row ----------------------------------------
            const float dz1 = DAMP ? dz_of(s) : 0.f;
            f4(*Rg)[LROW] = ring[ph % NRING];
            {
                const f4 lap = laplace(ctrI, S, srI, [&](int k) -> const f4 & { return qnI[(ph + R + k) % NQ]; });
                f4 u1 = update(ctrI, upI[cur], cring[ph % NRING], lap, dxyI, dz1);
                if (e1 > e0) inject(u1, s, offI, 0);
                q1I[ph] = u1;
                Rg[rrI][1 + lane] = u1;
                if (storeI && s >= z0 && s < z1) st4(a.out1 + (int64_t)s * sz + pI, u1);
            }
            {
                const f4 lap = laplace(ctrH, S, lrH, [&](int k) -> const f4 & { return qnH[(ph + R + k) % NQ]; });
                f4 u1 = update(ctrH, upH[cur], cH[cur], lap, dxyH, dz1);
                if (e1 > e0) inject(u1, s, hasH ? offH : -1000, 0);
                Rg[rrH][1 + lane] = u1;
            }
            // ---- stage 2: u^{n+2}(z), z = s - R, on the interior row ------------------------------------------
            const int z = s - R;
            if (z >= z0) {  // (z < z1 by the loop bound)
                // (no barrier: plane z's rows were written R iterations = R barriers ago, and the slot is not
                // overwritten before the next iteration's barrier)
                f4(*Rz)[LROW] = ring[(ph + 1) % NRING];  // slot of plane s - R
                const f4 c1 = q1I[(ph + 2 + R) % NQ];      // u^{n+1}(z): planes s-2R .. s live in slots ph+2 .. ph+2+2R
                const f4 lap = laplace(c1, Rz, rrI, [&](int k) -> const f4 & { return q1I[(ph + 2 + R + k) % NQ]; });
                f4 u2 = update(c1, qnI[ph % NQ], cring[(ph + 1) % NRING], lap, dxyI, DAMP ? dz_of(z) : 0.f);
                if (e1 > e0) inject(u2, z, offI, 1);
                if (storeI) st4(a.out2 + (int64_t)z * sz + pI, u2);
            }
            cring[(ph + 1) % NRING] = cnext;  // C(s + 1) takes the slot C(s - R) has just left
        }
    }
}

int pair3d_num_tiles(const GridDesc &g, int zchunk, int tw) {
    const int nxt = g.nx <= 256 ? 1 : (g.nx + tw - 1) / tw;
    return nxt * ((g.ny + PAIR3D_TY - 1) / PAIR3D_TY) * ((g.nz + zchunk - 1) / zchunk);
}

void pair3d_default_tuning(const GridDesc &g, int *zchunk, int *tw) {
    // x: one tile for rows of <= 256 columns, else equal tiles of <= 240 interior columns (multiples of 4)
    int t = 256;
    if (g.nx > 256) {
        const int n = (g.nx + 239) / 240;
        t = ((g.nx + n - 1) / n + 3) & ~3;
    }
    *tw = t;
    const int nxt = g.nx <= 256 ? 1 : (g.nx + t - 1) / t;
    const int64_t cols = (int64_t)nxt * ((g.ny + PAIR3D_TY - 1) / PAIR3D_TY);
    // one resident round of workgroups (256 CUs): as few z chunks as that allows (each re-computes 2r planes)
    const int nzc = (int)std::max<int64_t>(1, std::min<int64_t>(g.nz, 256 / std::max<int64_t>(1, cols)));
    *zchunk = std::max((g.nz + nzc - 1) / nzc, std::min(g.nz, 16));
}

// workgroup (in the kernel's renumbered order) that owns interior point (z, y, x)
int pair3d_tile_of(const GridDesc &g, int zchunk, int tw, int z, int y, int x) {
    const int nxt = g.nx <= 256 ? 1 : (g.nx + tw - 1) / tw;
    const int nyt = (g.ny + PAIR3D_TY - 1) / PAIR3D_TY;
    return ((z / zchunk) * nyt + y / PAIR3D_TY) * nxt + (nxt == 1 ? 0 : x / tw);
}

template <int R>
static hipError_t launch_pair_r(const GridDesc &g, const Pair3dArgs &a, int zchunk, int tw, hipStream_t s) {
    const int nxt = g.nx <= 256 ? 1 : (g.nx + tw - 1) / tw;
    const int nyt = (g.ny + PAIR3D_TY - 1) / PAIR3D_TY;
    const int nblk = nxt * nyt * ((g.nz + zchunk - 1) / zchunk);
    const int nth = 64 * PAIR3D_TY;
    const int nrb = ((a.rec_out0 || a.rec_out1) && a.nrec > 0) ? (a.nrec + nth * 4 - 1) / (nth * 4) : 0;
    dim3 block(64, PAIR3D_TY), grid(nblk + nrb);
    // (undamped grids only since round 4: the damped O(8) instantiation spilled 70 registers -- 204 B of scratch per lane --
    // and the path is an opt-in negative result; fwi_create keeps sponge contexts on the single-step kernel)
    if (a.damp) return hipErrorInvalidValue;
    hipLaunchKernelGGL((step3d_pair<R, false>), grid, block, 0, s, a, g, zchunk, nxt, nyt, nblk, tw);
    return hipGetLastError();
}

hipError_t launch_pair3d(const GridDesc &g, const Pair3dArgs &a, int zchunk, int tw, hipStream_t s) {
    switch (g.r) {
        case 1: return launch_pair_r<1>(g, a, zchunk, tw, s);
        case 2: return launch_pair_r<2>(g, a, zchunk, tw, s);
        default: return launch_pair_r<4>(g, a, zchunk, tw, s);
    }
}

}  // namespace fwi
