// gfx950 (MI355X, CDNA4) kernels of the acoustic time-stepping path.
//
// No reference counterpart: the reference has no stencil / PML / adjoint /
// imaging code; the scheme is the build-defined one stated at
// the top of oracle/fwi_oracle.py.  All kernels are HBM-bandwidth bound
// (~2.4 flop/B), so no MFMA: the work here is coalesced 16 B/lane streams,
// an LDS-staged xy halo tile, a register queue along z and wave64 shuffles
// for the reductions.
//
// Numerics: the Laplacian is evaluated in difference form
//     L u = sum_k c_k * ( sum_axes (u[+k] + u[-k]) - 2 D u )
// which is algebraically the centred star but keeps fp32 round-off ~30x lower
// over thousands of steps than the coefficient form (the rounded coefficients
// of the latter do not sum to zero, which acts as a spurious mass term).
#include "fwi_device.h"

namespace fwi {

GridDesc make_grid(int ndim, int nz, int ny, int nx, int order, int xpitch_extra) {
    GridDesc g;
    g.ndim = ndim;
    g.nz = nz;
    g.ny = (ndim == 3) ? ny : 1;
    g.nx = nx;
    g.r = order / 2;
    const int hy = (ndim == 3) ? HALO : 0;
    // x pitch: TIGHT -- nx rounded to XALIGN (4: one 16-byte lane) plus ONE halo: the HALO zero cells right of a row are the HALO cells left
    // of the next (nothing writes them).  Every kernel clamps its lanes' addresses into [0, nx] + the halo, so no tile needs
    // slack, and the unused bytes between rows are expensive: until round 3 rows were 4 + roundup(nx, 256) + 4 floats, which
    // cost the 3-D sizes that are not multiples of 256 up to a quarter of their HBM-regime rate (512 x 512 x 384 280 -> 354
    // Gpts/s, 448^3 299 -> 351, 640^3 293 -> 333, 320^3 266 -> 325, 272^3 277 -> 339, 384^3 289 -> 323).  Pitch sweep at 512^3,
    // nx + 8 / 16 / 24 / 32 / 64 / 128 / 256 floats: 350 / 340 / 317 / 305 / 292 / 287 / 276 Gpts/s; nx + 4 against nx + 8: 256^3
    // 424 -> 434, 384^3 301 -> 305.  2-D (cache-resident up to ~4096^2): +1..2 % at every size from 1000^2 to 8192^2.
    // `xpitch_extra` (FWI_XPITCH_EXTRA floats, read once by fwi_create) is the A/B hook.
    g.sy = HALO + round_up(nx, XALIGN) + std::max(0, xpitch_extra / 4 * 4);
    // (rows whose interior starts on a 128-byte line -- pitch nx + 32 with the gap shared as right / left halo -- were
    // measured as well: 256^3 +2 %, 512^3 +1 %, 640^3 -8 %: not adopted)
    // (sharing the y-halo rows between consecutive planes the same way was measured: within noise, 512^3 349 vs 352)
    const int64_t py = (ndim == 3) ? HALO + round_up(g.ny, YALIGN) + HALO : 1;
    g.sz = g.sy * py;
    g.off0 = (int64_t)HALO * g.sz + (int64_t)hy * g.sy + HALO;
    // 2-D: rows are the tiled axis of step2d_tile, so they are rounded like y is in 3-D.
    // 3-D: LOOKAHEAD extra zero planes behind the far z halo, so the stream kernel's prefetches of
    // planes z + r + 1 ... need no clamping (affine addresses: the plane offsets strength-reduce).
    g.ptot = g.sz * (int64_t)(((ndim == 2) ? round_up(nz, YALIGN) : nz + LOOKAHEAD) + 2 * HALO) + HALO + 2 * MAX_TILE_X;  // (+ the last row's right halo, and
                                        // the edge loads of a 256-column tile whose row ends early: values never used)
    g.cx = (int)round_up(nx, 4);
    g.npts = (int64_t)nz * g.ny * g.cx;
    return g;
}

// ---------------------------------------------------------------------------
// POINT kernel: one thread per grid point, all 4r+... neighbours straight from
// global memory (L1/L2 absorb the re-reads).  Generic in dtype, order and
// dimension; the fallback for shapes the stream kernel does not take and the
// on-device cross-check of it.
// ---------------------------------------------------------------------------
template <typename T, int R, int NDIM, bool SAVE_Q, bool IMAGE, bool INC>
__global__ __launch_bounds__(256) void step_point(StepArgs<T> a, GridDesc g, int nbx, int nby, int nblk) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((int)blockIdx.x >= nblk) {
        // receiver sampling of the previous step's field (u_cur) rides along as extra workgroups
        for (int i = ((int)blockIdx.x - nblk) * 256 + tid; i < a.nrec; i += ((int)gridDim.x - nblk) * 256)
            a.rec_out[i] = a.u_cur[a.rec_pidx[i]] * a.rec_scale;
        return;
    }
    const int bx = blockIdx.x % nbx;
    const int t2 = blockIdx.x / nbx;
    const int x = bx * 64 + threadIdx.x;
    int y, z;
    if (NDIM == 3) {
        y = (t2 % nby) * 4 + threadIdx.y;
        z = t2 / nby;
    } else {
        y = 0;
        z = t2 * 4 + threadIdx.y;
    }
    if (x < g.nx && y < g.ny && z < g.nz) {
        const int64_t p = g.off0 + (int64_t)z * g.sz + (int64_t)y * g.sy + x;
        const T *u = a.u_cur + p;
        const T uc = u[0];
        T lap = T(0);
#pragma unroll
        for (int k = R; k >= 1; --k) {
            T t = (u[-k] + u[k]) + (u[-(int64_t)k * g.sz] + u[(int64_t)k * g.sz]);
            if (NDIM == 3) t += (u[-(int64_t)k * g.sy] + u[(int64_t)k * g.sy]);
            t = fma(T(-2 * NDIM), uc, t);
            lap = fma(a.ck[k], t, lap);
        }
        if (NDIM == 3 && a.pml_tz) {  // the z / y border's CPML term, handed over by the line launch before this step
            if (pml_in_shell(z, g.nz, a.npml, R))
                lap += a.pml_tz[((int64_t)pml_shell_index(z, g.nz, a.npml, R) * g.ny + y) * g.cx + x];
            if (pml_in_shell(y, g.ny, a.npml, R))
                lap += a.pml_ty[((int64_t)z * pml_shell_rows(g.ny, a.npml, R) + pml_shell_index(y, g.ny, a.npml, R)) * g.cx + x];
        }
        const T q = a.C[p] * lap;
        const T up = INC ? a.v[p] : a.u_prev[p];  // increment form: the operand is v^n = u^n - u^{n-1}
        T un;
        if (a.damp) {
            T d = a.dz[z] + a.dx[x];
            if (NDIM == 3) d += a.dy[y];
            if (INC) {
                const T vn = ((T(1) - d) * up + q) / (T(1) + d);
                a.v[p] = vn;
                un = uc + vn;
            } else {
                un = (T(2) * uc - (T(1) - d) * up + q) / (T(1) + d);
            }
        } else if (INC) {
            const T vn = up + q;
            a.v[p] = vn;
            un = uc + vn;
        } else {
            un = (T(2) * uc - up) + q;
        }
        a.u_prev[p] = un;
        const int64_t ci = ((int64_t)z * g.ny + y) * g.cx + x;
        if (SAVE_Q) a.q_out[ci] = q;
        // (paired imaging: the second pairing takes u_prev -- in increment form u - v, `up` being v there)
        if (IMAGE) a.g[ci] += a.q_in2 ? fma(INC ? uc - up : up, a.q_in2[ci], uc * a.q_in[ci]) : uc * a.q_in[ci];
    }
    // injection into the points this workgroup has just written (see the stream kernel)
    if (a.inj_start) {
        const int s0 = a.inj_start[blockIdx.x], s1 = a.inj_start[blockIdx.x + 1];
        if (s1 > s0) {
            __syncthreads();
            inject_runs<T, INC, SAVE_Q>(a, s0, s1, tid, 256);
        }
    }
}

constexpr int TILE_X = 256;                // 2-D tile kernel: floats per tile row = 64 lanes x float4
constexpr int LROW4 = TILE_X / 4 + 2;      // its LDS row in float4: [left edge][64][right edge]

// ---------------------------------------------------------------------------
// TILE kernel (2-D, fp32): the 2-D sibling of the stream kernel.  A workgroup of 64 x TY threads
// owns a 256 x TY (x, z) tile, one float4 per thread; the TY + 2r rows it needs are staged in LDS
// once (one barrier), x neighbours come from the same LDS row.  2-D grids of BASELINE size are
// L2-resident (1024^2: 4 MiB per field), so a time step is bounded by the ~1.5 us kernel boundary
// plus one load -> barrier -> compute -> store chain; there is no marching axis to pipeline.
// ---------------------------------------------------------------------------
template <int R, int TY, bool DAMP, bool SAVE_Q, int IMAGE>
__global__ __launch_bounds__(64 * TY) void step2d_tile(StepArgs<float> a, GridDesc g, int nxt, int nblk) {
    constexpr int NH = (2 * R + TY - 1) / TY;
    constexpr int TRASH = TY + 2 * R;
    constexpr int LROWS = TY + 2 * R + 1;
    __shared__ f4 L[LROWS][LROW4];

    const int lane = threadIdx.x, ty = threadIdx.y;
    int bid = blockIdx.x;
    if (bid >= nblk) {  // receiver sampling of the previous step's field
        for (int i = (bid - nblk) * (64 * TY) + ty * 64 + lane; i < a.nrec; i += (gridDim.x - nblk) * 64 * TY)
            a.rec_out[i] = a.u_cur[a.rec_pidx[i]] * a.rec_scale;
        return;
    }
    const int bx = bid % nxt, bz = bid / nxt;
    const int x0 = bx * TILE_X + 4 * lane;
    const int z0 = bz * TY, z = z0 + ty;
    const bool act = (x0 < g.nx) && (z < g.nz);
    const int64_t sz = g.sz;
    const int xa = min(x0, ((g.nx - 1) / 4 + 1) * 4);  // lanes right of the grid read one shared zero vector
    const int64_t poff = g.off0 + (int64_t)z * sz + xa;

    const f4 ctr = ld4(a.u_cur + poff);
    f4 halo[NH];
    int hrow[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int hr = ty + i * TY;
        const bool valid = hr < 2 * R;
        const int zh = !valid ? z : (hr < R) ? z0 - R + hr : z0 + TY + (hr - R);
        hrow[i] = !valid ? TRASH : (hr < R) ? hr : TY + hr;
        halo[i] = ld4(a.u_cur + g.off0 + (int64_t)zh * sz + xa);
    }
    const int64_t rowoff = g.off0 + (int64_t)z * sz + bx * TILE_X;
    const int64_t eoff = (lane < 4) ? rowoff - 4 + lane : (lane >= 60) ? rowoff + TILE_X + (lane - 60) : rowoff - 1;
    const int erow = (lane < 4 || lane >= 60) ? R + ty : TRASH;
    const int ecol = (lane < 4) ? lane : (lane >= 60) ? 4 * (LROW4 - 1) + (lane - 60) : lane;
    const float edge = a.u_cur[eoff];
    const f4 up = ld4(a.u_prev + poff), Cc = ld4(a.C + poff);
    const int64_t ci = (int64_t)z * g.cx + x0;
    f4 qi, qi2, gi;
    if (IMAGE) {
        qi = ld4_stream(a.q_in + (act ? ci : 0));
        if (IMAGE == 2) qi2 = ld4_stream(a.q_in2 + (act ? ci : 0));
        gi = ld4(a.g + (act ? ci : 0));
    }
    f4 A, B;
    if (DAMP) {
        const float dzv = (z < g.nz) ? a.dz[z] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = dzv + ((x0 + j < g.nx) ? a.dx[x0 + j] : 0.f);
            B.v[j] = 1.f - d;
            A.v[j] = rcp_nr(1.f + d);
        }
    }

    L[R + ty][1 + lane] = ctr;
#pragma unroll
    for (int i = 0; i < NH; ++i) L[hrow[i]][1 + lane] = halo[i];
    reinterpret_cast<float *>(&L[erow][0])[ecol] = edge;
    __syncthreads();

    const f4 xl = L[R + ty][lane], xr = L[R + ty][lane + 2];
    float X[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        X[j] = xl.v[j];
        X[4 + j] = ctr.v[j];
        X[8 + j] = xr.v[j];
    }
    f4 lap;
#pragma unroll
    for (int j = 0; j < 4; ++j) lap.v[j] = 0.f;
#pragma unroll
    for (int k = R; k >= 1; --k) {
        const f4 zm = L[R + ty - k][1 + lane], zp = L[R + ty + k][1 + lane];
        const float c = a.ck[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (X[4 + j - k] + X[4 + j + k]) + (zm.v[j] + zp.v[j]);
            t = fmaf(-4.f, X[4 + j], t);
            lap.v[j] = fmaf(c, t, lap.v[j]);
        }
    }
    f4 q, un;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q.v[j] = Cc.v[j] * lap.v[j];
        if (DAMP)
            un.v[j] = (fmaf(2.f, X[4 + j], -B.v[j] * up.v[j]) + q.v[j]) * A.v[j];
        else
            un.v[j] = (2.f * X[4 + j] - up.v[j]) + q.v[j];
    }
    if (act) {
        st4(a.u_prev + poff, un);
        if (SAVE_Q) st4_stream(a.q_out + ci, q);
        if (IMAGE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gi.v[j] = fmaf(X[4 + j], qi.v[j], gi.v[j]);
                if (IMAGE == 2) gi.v[j] = fmaf(up.v[j], qi2.v[j], gi.v[j]);
            }
            st4(a.g + ci, gi);
        }
    }
    if (a.inj_start) {  // injection into the points this workgroup has just written
        const int s0 = a.inj_start[bid], s1 = a.inj_start[bid + 1];
        if (s1 > s0) {
            __syncthreads();
            inject_runs<float, false, SAVE_Q>(a, s0, s1, ty * 64 + lane, 64 * TY);
        }
    }
}

int stream_tile_of(const GridDesc &g, const StreamTuning &t, int z, int y, int x) {
    const int nxt = stream_nxt(g, t.tile_x);
    if (g.ndim == 2) return (z / t.ty) * nxt + x / t.tile_x;
    const int nyt = (g.ny + t.ty - 1) / t.ty;
    return ((z / t.zchunk) * nyt + y / t.ty) * nxt + x / t.tile_x;
}

int stream_num_tiles(const GridDesc &g, const StreamTuning &t) {
    const int nxt = stream_nxt(g, t.tile_x);
    if (g.ndim == 2) return nxt * ((g.nz + t.ty - 1) / t.ty);
    return nxt * ((g.ny + t.ty - 1) / t.ty) * ((g.nz + t.zchunk - 1) / t.zchunk);
}

bool stream_supported(const GridDesc &g, bool is_f32) {
    if (g.nz < 1) return false;
    // any nx: the padded fields and the compact arrays (row stride cx = nx rounded up to 4) keep every
    // lane's 16-byte vector aligned; lanes straddling x = nx compute zeros there (C = 0 in the pad)
    if (is_f32) return true;                  // float4 lanes (3-D stream kernel, 2-D tile kernel)
    return g.ndim == 3;                       // double2 lanes: 3-D stream kernel only
}

bool stream_xpml_supported(const GridDesc &g, const StreamTuning &t, int npml, bool is_f32) {
    if (g.ndim != 3 || !is_f32 || g.r != 4 || npml < 4) return false;
    // (a lane's four cells lie wholly inside or outside the border when npml % 4 == 0; an even npml leaves one lane per
    // side astride the inner edge, whose stores are masked (stream_xpml_partial); the high side's lanes stay 16-byte
    // aligned in the slab row because nx - 2 npml is a multiple of 4)
    if (npml % 2 || g.nx % 4) return false;
    if (g.nx < 2 * (npml + g.r)) return false;         // the two borders (and their reach) do not meet
    const int tw = t.tile_x, nxt = stream_nxt(g, tw);
    // each border and the r cells it reaches into lie inside ONE tile row, i.e. inside one wave
    return tw >= npml + g.r && g.nx - (nxt - 1) * tw >= npml + g.r;
}

StreamTuning stream_default_tuning(const GridDesc &g, bool is_f32, double extra_bytes) {
    const int full_x = is_f32 ? 256 : 128, vl = is_f32 ? 4 : 2;
    // 3-D: split nx into equal x tiles (multiples of the lane vector) rather than full ones plus a remainder
    const int nxt0 = (g.nx + full_x - 1) / full_x;
    const int tile_x = (g.ndim == 2) ? full_x : (int)round_up((g.nx + nxt0 - 1) / nxt0, vl);
    // 2-D: rows per workgroup.  16 is 7 % faster for a lone shot (6.98 vs 7.46 us/step at 1024^2) but
    // 8 co-schedules better when several shots share the GPU (4 concurrent: 3.5 vs 4.4 us/step/shot).
    if (g.ndim == 2) return StreamTuning{8, 1, 1, tile_x};
    // Measured on MI355X (tools/tune_stream.py, tools/size_sweep.py; profiles/r02_stream_tuning_sweep.txt).  Two
    // things decide: every z chunk re-reads 2r halo planes and pays a prologue (~10 planes' worth), so chunks should
    // be long; and workgroups run in rounds of one per CU (256), so the last round should not be a sliver -- a grid
    // whose tiles make 288 columns (768^3 with 8-row tiles) spends its second round on 32 workgroups (235 Gpts/s;
    // 305 with 96-plane chunks = 9 full rounds).  Cost of a candidate (rows per tile, chunks), relative to ideal,
    // with W = the workgroups that saturate the memory system:
    //   (1 + 10 / zchunk) x sum over rounds of max(1, workgroups in the round / W), over (workgroups / W)
    // Two regimes: while the three fields fit the Infinity Cache every CU counts (W = 256) and 4-row tiles cost
    // nothing extra -- their third read of a y row hits in L2; from HBM ~200 workgroups already saturate (W = 200;
    // 384^3 is fastest with 192 workgroups of 192 planes, not with 768 of 48) and the 4-row tile's extra row reads
    // show (200 x 400 x 400: TY 8 x 100 planes 283 Gpts/s, TY 4 x 200 planes 240).  The model reproduces the
    // hand-tuned choices of round 1 (256^3: TY 4 x 64 planes, 407 Gpts/s vs TY 8 x 32 347; 512^3: TY 8 x 256, 328 vs
    // TY 4 x 256 302) and the measured ranking of the chunk lengths at 300^3, 384^3 and 768^3; against the round-1
    // rule (one round, then stop; FWI_STREAM_TUNING=legacy) it changes 700^3 (+39 %) and 768^3 (+24 %) and nothing
    // else from 96^3 to 1024^3.  What it cannot fix: a grid whose columns make 192 - 240 workgroups runs at about
    // that share of the 512^3 rate (384^3: 280 vs 352 Gpts/s) -- per-CU throughput is bounded (neither a deeper
    // prefetch, PF = 2, nor fuller waves, 32-lane rows for 100 % lane use at 384^3, moved it: both measured, both
    // dropped), and shorter chunks to fill more rounds cost more in halo planes than the idle CUs do.
    static const bool legacy = getenv("FWI_STREAM_TUNING") && !strcmp(getenv("FWI_STREAM_TUNING"), "legacy");
    if (legacy) {
        StreamTuning best{4, g.nz, 1, tile_x};
        for (int ty : {8, 4}) {
            const int64_t tiles_xy = stream_nxt(g, tile_x) * (round_up(g.ny, ty) / ty);
            const int nzc = (int)std::max<int64_t>(1, std::min<int64_t>(g.nz, 256 / std::max<int64_t>(1, tiles_xy)));
            const int zc = std::max((g.nz + nzc - 1) / nzc, std::min(g.nz, 16));
            best = StreamTuning{ty, zc, 1, tile_x};
            if (zc >= 64 || zc >= g.nz) break;  // long enough chunks with 8-row tiles: keep them
        }
        return best;
    }
    constexpr double CUS = 256.0, PROLOGUE = 10.0;
    // (200 MiB: 256^3 = 192 MiB is resident, 272^3 = 230 MiB measurably is not -- it runs 12 % faster tuned as HBM)
    // Round 4: everything the step touches counts, not the three fields alone -- 256^3 with a 16-cell CPML (fields + 50 MB of
    // memory variables + 21 MB of handed-over terms) or in increment form (a fourth field) is an HBM-regime problem and
    // wants the HBM-regime shape: 8-row tiles x 32 planes instead of 4 x 64 (CPML forward 83.3 -> 71.1, adjoint 102.7 -> 93.6
    // us/step on one box; increment form 67.8 / 81.3 / 77.3 -> 63.7 / 76.7 / 75.4) -- once the 8-row x-border variants
    // stopped spilling, that is: in round 3 the same choice measured 116 against 100.
    const bool cache_resident = 3.0 * (double)g.npts * (is_f32 ? 4 : 8) + extra_bytes <= 200.0 * 1024 * 1024;
    // (the kernels that carry extras -- the x border's recursion, the handed-over terms, the increment field -- have a
    // longer dependency chain per plane: every CU counts for them even from HBM.  Measured at 256^3 with the CPML: 8 rows
    // x 32 planes = 256 workgroups 71.6 / 79.8 / 93.7 us/step forward / store / adjoint, x 37 planes = 224 workgroups -- what
    // W = 200 picks -- 81.6 / 97.1 / 100.4, round 3's 4 rows x 64 planes 84.1 / 95.6 / 103.3; at 320^3 one round of 240
    // workgroups 188 against 248 for 320 or 160 of them)
    const double WSAT = (cache_resident || extra_bytes > 0.0) ? 256.0 : 200.0, TY4_PENALTY = cache_resident ? 1.0 : 1.15;
    StreamTuning best{8, g.nz, 1, tile_x};
    double best_cost = 1e30;
    const int zc_min = std::min(g.nz, 16);
    for (int ty : {8, 4}) {
        const int64_t tiles_xy = stream_nxt(g, tile_x) * (round_up(g.ny, ty) / ty);
        int last_zc = 0;
        for (int nzc = 1; nzc <= g.nz; ++nzc) {
            const int zc = (g.nz + nzc - 1) / nzc;
            if (zc < zc_min) break;
            if (zc == last_zc) continue;
            last_zc = zc;
            const double nblk = (double)tiles_xy * ((g.nz + zc - 1) / zc);
            const double full = std::floor(nblk / CUS), rem = nblk - full * CUS;
            const double rounds = full * (CUS / WSAT) + (rem > 0 ? std::max(1.0, rem / WSAT) : 0.0);
            const double cost = (1.0 + PROLOGUE / zc) * rounds / (nblk / WSAT) * (ty == 4 ? TY4_PENALTY : 1.0);
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best = StreamTuning{ty, zc, 1, tile_x};
            }
        }
    }
    return best;
}

template <typename T>
static hipError_t launch_stream(const GridDesc &g, const StepArgs<T> &a, const StreamTuning &t, hipStream_t s) {
    switch (g.r) {
        case 1: return launch_stream_r<T, 1>(g, a, t, s);
        case 2: return launch_stream_r<T, 2>(g, a, t, s);
        default: return launch_stream_r<T, 4>(g, a, t, s);
    }
}

template <int R, int TY, bool DAMP>
static hipError_t launch_tile2d_mode(const GridDesc &g, const StepArgs<float> &a, hipStream_t s) {
    const int nxt = stream_nxt(g, TILE_X);
    const int nblk = nxt * ((g.nz + TY - 1) / TY);
    const int nrb = (a.rec_out && a.nrec > 0) ? (a.nrec + 64 * TY * 4 - 1) / (64 * TY * 4) : 0;
    dim3 block(64, TY), grid(nblk + nrb);
    if (a.q_out)
        hipLaunchKernelGGL((step2d_tile<R, TY, DAMP, true, 0>), grid, block, 0, s, a, g, nxt, nblk);
    else if (a.q_in && a.q_in2)
        hipLaunchKernelGGL((step2d_tile<R, TY, DAMP, false, 2>), grid, block, 0, s, a, g, nxt, nblk);
    else if (a.q_in)
        hipLaunchKernelGGL((step2d_tile<R, TY, DAMP, false, 1>), grid, block, 0, s, a, g, nxt, nblk);
    else
        hipLaunchKernelGGL((step2d_tile<R, TY, DAMP, false, 0>), grid, block, 0, s, a, g, nxt, nblk);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_tile2d_r(const GridDesc &g, const StepArgs<float> &a, const StreamTuning &t,
                                  hipStream_t s) {
    if (a.damp) {
        switch (t.ty) {
            case 4: return launch_tile2d_mode<R, 4, true>(g, a, s);
            case 16: return launch_tile2d_mode<R, 16, true>(g, a, s);
            default: return launch_tile2d_mode<R, 8, true>(g, a, s);
        }
    }
    switch (t.ty) {
        case 4: return launch_tile2d_mode<R, 4, false>(g, a, s);
        case 16: return launch_tile2d_mode<R, 16, false>(g, a, s);
        default: return launch_tile2d_mode<R, 8, false>(g, a, s);
    }
}

static inline void point_blocks(const GridDesc &g, int &nbx, int &nby, int &nblk) {
    nbx = (g.nx + 63) / 64;
    nby = (g.ndim == 3) ? (g.ny + 3) / 4 : (g.nz + 3) / 4;
    nblk = nbx * nby * ((g.ndim == 3) ? g.nz : 1);
}

int point_tile_of(const GridDesc &g, int z, int y, int x) {
    int nbx, nby, nblk;
    point_blocks(g, nbx, nby, nblk);
    return (g.ndim == 3) ? (z * nby + y / 4) * nbx + x / 64 : (z / 4) * nbx + x / 64;
}

int point_num_tiles(const GridDesc &g) {
    int nbx, nby, nblk;
    point_blocks(g, nbx, nby, nblk);
    return nblk;
}

template <typename T, int R, int NDIM>
static hipError_t launch_point_mode(const GridDesc &g, const StepArgs<T> &a, hipStream_t s) {
    int nbx, nby, nblk;
    point_blocks(g, nbx, nby, nblk);
    const int nrb = (a.rec_out && a.nrec > 0) ? (a.nrec + 1023) / 1024 : 0;
    dim3 block(64, 4), grid(nblk + nrb);
    if (a.v) {
        if (a.q_out)
            hipLaunchKernelGGL((step_point<T, R, NDIM, true, false, true>), grid, block, 0, s, a, g, nbx, nby, nblk);
        else if (a.q_in)
            hipLaunchKernelGGL((step_point<T, R, NDIM, false, true, true>), grid, block, 0, s, a, g, nbx, nby, nblk);
        else
            hipLaunchKernelGGL((step_point<T, R, NDIM, false, false, true>), grid, block, 0, s, a, g, nbx, nby, nblk);
    } else if (a.q_out)
        hipLaunchKernelGGL((step_point<T, R, NDIM, true, false, false>), grid, block, 0, s, a, g, nbx, nby, nblk);
    else if (a.q_in)
        hipLaunchKernelGGL((step_point<T, R, NDIM, false, true, false>), grid, block, 0, s, a, g, nbx, nby, nblk);
    else
        hipLaunchKernelGGL((step_point<T, R, NDIM, false, false, false>), grid, block, 0, s, a, g, nbx, nby, nblk);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_point(const GridDesc &g, const StepArgs<T> &a, hipStream_t s) {
    if (g.ndim == 3) {
        switch (g.r) {
            case 1: return launch_point_mode<T, 1, 3>(g, a, s);
            case 2: return launch_point_mode<T, 2, 3>(g, a, s);
            default: return launch_point_mode<T, 4, 3>(g, a, s);
        }
    }
    switch (g.r) {
        case 1: return launch_point_mode<T, 1, 2>(g, a, s);
        case 2: return launch_point_mode<T, 2, 2>(g, a, s);
        default: return launch_point_mode<T, 4, 2>(g, a, s);
    }
}

template <>
hipError_t launch_step<float>(int kernel, const GridDesc &g, const StepArgs<float> &a,
                              const StreamTuning &t, hipStream_t s) {
    if (kernel == K_STREAM && g.ndim == 2) {
        switch (g.r) {
            case 1: return launch_tile2d_r<1>(g, a, t, s);
            case 2: return launch_tile2d_r<2>(g, a, t, s);
            default: return launch_tile2d_r<4>(g, a, t, s);
        }
    }
    if (kernel == K_STREAM) return launch_stream<float>(g, a, t, s);
    return launch_point<float>(g, a, s);
}

template <>
hipError_t launch_step<double>(int kernel, const GridDesc &g, const StepArgs<double> &a, const StreamTuning &t,
                               hipStream_t s) {
    if (kernel == K_STREAM && g.ndim == 3) return launch_stream<double>(g, a, t, s);
    return launch_point<double>(g, a, s);
}

// ---------------------------------------------------------------------------
// Receiver sampling of the final step (earlier steps are sampled inside the step launches).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void record_kernel(const T *u, const int64_t *pidx, T *out, T scale, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = u[pidx[i]] * scale;
}

template <typename T>
hipError_t launch_record(const T *u, const int64_t *pidx, T *out, T scale, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(record_kernel<T>, dim3((n + 63) / 64), dim3(64), 0, s, u, pidx, out, scale, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Imaging tail, gradient scaling, dot product.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T q_elem(const T *q, int64_t i, int bf16) {
    if (bf16) return (T)__uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(q)[i] << 16);
    return q[i];
}

// g[x_s] += sum_n mu^{n+1}(x_s) * cq_s * w^n_s: the part of the imaging term that is the source itself (bf16 store)
template <typename T>
__global__ void source_image_kernel(const T *adj_series, const T *wav, const int64_t *cidx, const T *cq, T *gacc,
                                    int nt, int nsrc, int stride, T inv_rs) {
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nsrc) return;
    double acc = 0.0;
    for (int n = 0; n < nt; n += stride)
        acc += (double)adj_series[(int64_t)n * nsrc + sidx] * (double)wav[(int64_t)n * nsrc + sidx];
    atomicAdd(gacc + cidx[sidx], (T)(acc * (double)inv_rs * (double)cq[sidx]));
}

template <typename T>
hipError_t launch_source_image(const T *adj_series, const T *wav, const int64_t *cidx, const T *cq, T *gacc, int nt,
                               int nsrc, int stride, T inv_rs, hipStream_t s) {
    if (nsrc <= 0) return hipSuccess;
    hipLaunchKernelGGL(source_image_kernel<T>, dim3((nsrc + 63) / 64), dim3(64), 0, s, adj_series, wav, cidx, cq, gacc,
                       nt, nsrc, stride, inv_rs);
    return hipGetLastError();
}

template <typename T>
__global__ void image_kernel(GridDesc g, const T *u, const T *q, T *gacc, int q_bf16) {
    const int64_t rowlen = g.cx;
    for (int64_t ci = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ci < g.npts;
         ci += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = ci / rowlen;  // z * ny + y
        const int x = (int)(ci - row * rowlen);
        if (x >= g.nx) continue;  // pad column of the compact layout
        const int z = (int)(row / g.ny), y = (int)(row % g.ny);
        const int64_t p = g.off0 + (int64_t)z * g.sz + (int64_t)y * g.sy + x;
        gacc[ci] += u[p] * q_elem<T>(q, ci, q_bf16);
    }
}

template <typename T>
hipError_t launch_image(const GridDesc &g, const T *u, const T *q, T *gacc, int q_bf16, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (g.npts + 255) / 256));
    hipLaunchKernelGGL(image_kernel<T>, dim3(blocks), dim3(256), 0, s, g, u, q, gacc, q_bf16);
    return hipGetLastError();
}

// padded C = dt^2 c^2 from the compact velocity; counts non-finite / non-positive entries in *bad
template <typename T>
__global__ void build_model_kernel(GridDesc g, const T *c, T *Cpad, double dt2, int *bad) {
    for (int64_t ci = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ci < g.npts;
         ci += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = ci / g.cx;
        const int x = (int)(ci - row * g.cx);
        if (x >= g.nx) continue;  // pad column: C stays 0 there, which freezes u at 0 (a Dirichlet wall)
        const int z = (int)(row / g.ny), y = (int)(row % g.ny);
        const double cv = (double)c[ci];
        if (!(cv > 0.0) || !isfinite(cv)) atomicAdd(bad, 1);
        Cpad[g.off0 + (int64_t)z * g.sz + (int64_t)y * g.sy + x] = (T)(dt2 * cv * cv);
    }
}

template <typename T>
hipError_t launch_build_model(const GridDesc &g, const T *c, T *Cpad, double dt2, int *bad, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (g.npts + 255) / 256));
    hipLaunchKernelGGL(build_model_kernel<T>, dim3(blocks), dim3(256), 0, s, g, c, Cpad, dt2, bad);
    return hipGetLastError();
}

template <typename T>
__global__ void finalize_gradient_kernel(const T *gacc, const T *c, T *out, double scale,
                                         int wrt_velocity, int64_t n, int nx, int cx) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double v = (double)gacc[i] * scale;
        if (wrt_velocity) {
            const double cc = (double)c[i];
            v *= -2.0 / (cc * cc * cc);
        }
        if (cx != nx && (int)(i % cx) >= nx) v = 0.0;  // pad column (c = 0 there)
        out[i] = (T)v;
    }
}

template <typename T>
hipError_t launch_finalize_gradient(const GridDesc &g, const T *gacc, const T *c, T *out, double scale,
                                    int wrt_velocity, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (g.npts + 255) / 256));
    hipLaunchKernelGGL(finalize_gradient_kernel<T>, dim3(blocks), dim3(256), 0, s, gacc, c, out, scale,
                       wrt_velocity, g.npts, g.nx, g.cx);
    return hipGetLastError();
}

// compact (stride cx, pad columns zeroed) <-> contiguous (stride nx)
template <typename T>
__global__ void repack_kernel(T *dst, const T *src, int64_t rows, int nx, int cx, int to_compact) {
    const int64_t n = rows * cx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cx;
        const int x = (int)(i - row * cx);
        if (to_compact)
            dst[i] = x < nx ? src[row * nx + x] : T(0);
        else if (x < nx)
            dst[row * nx + x] = src[i];
    }
}

template <typename T>
hipError_t launch_repack(const GridDesc &g, T *dst, const T *src, int to_compact, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (g.npts + 255) / 256));
    hipLaunchKernelGGL(repack_kernel<T>, dim3(blocks), dim3(256), 0, s, dst, src, (int64_t)g.nz * g.ny, g.nx, g.cx,
                       to_compact);
    return hipGetLastError();
}

// cells of a padded field outside the interior that are not exactly zero (NaN counts)
template <typename T>
__global__ void dirty_padding_kernel(GridDesc g, const T *f, unsigned long long *bad) {
    const int hy = g.ndim == 3 ? HALO : 0;
    unsigned long long n = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < g.ptot; p += (int64_t)gridDim.x * blockDim.x) {
        // interior cell (z, y, x) lives at off0 + z sz + y sy + x with off0 = HALO sz + hy sy + HALO: shift by the x halo
        const int64_t q = p - HALO;
        bool interior = false;
        if (q >= 0) {
            const int64_t zz = q / g.sz, r = q - zz * g.sz, yy = r / g.sy, xx = r - yy * g.sy;
            interior = zz >= HALO && zz < HALO + g.nz && yy >= hy && yy < hy + g.ny && xx < g.nx;
        }
        if (!interior && !(f[p] == T(0))) ++n;
    }
    if (n) atomicAdd(bad, n);
}

template <typename T>
hipError_t launch_count_dirty_padding(const GridDesc &g, const T *f, unsigned long long *bad, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (g.ptot + 255) / 256));
    hipLaunchKernelGGL(dirty_padding_kernel<T>, dim3(blocks), dim3(256), 0, s, g, f, bad);
    return hipGetLastError();
}

// wave64 shuffle reduction -> LDS across the 4 waves -> one fp64 atomic per block
template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(const T *a, const T *b, int64_t n, double *out) {
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc += (double)a[i] * (double)b[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

template <typename T>
hipError_t launch_dot(const T *a, const T *b, int64_t n, double *out, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256));
    hipLaunchKernelGGL(dot_kernel<T>, dim3(blocks), dim3(256), 0, s, a, b, n, out);
    return hipGetLastError();
}

// Least-squares residual and misfit on the device: r = syn - obs (written over obs, where the adjoint
// sweep reads its injection amplitudes), *out += sum r^2 -- same reduction shape as dot_kernel.
template <typename T>
__global__ __launch_bounds__(256) void residual_l2_kernel(const T *syn, T *obs_inout, int64_t n, double *out) {
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T r = syn[i] - obs_inout[i];
        obs_inout[i] = r;
        acc += (double)r * (double)r;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

template <typename T>
hipError_t launch_residual_l2(const T *syn, T *obs_inout, int64_t n, double *out, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256));
    hipLaunchKernelGGL(residual_l2_kernel<T>, dim3(blocks), dim3(256), 0, s, syn, obs_inout, n, out);
    return hipGetLastError();
}

// Off-grid points (multilinear interpolation): per-point time series <-> per-node time series on the device.
//   scatter  node[n, m] = w[m] * point[n, owner[m]]                      (injection amplitudes)
//   gather   point[n, p] = sum_{m in nodes(p)} w[m] * node[n, m]         (sampling; the transpose)
// gather: a point has at most 2^D <= 8 nodes, so 8 lanes own one (time sample, point) pair and reduce with
// __shfl_down within their group of 8 -- the "wavefront reduction for the receiver gather" of north_star.
template <typename T>
__global__ void scatter_series_kernel(const T *pt, T *node, const int *owner, const T *w, int nt, int npts, int nnodes) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)nt * nnodes) return;
    const int n = (int)(i / nnodes), m = (int)(i - (int64_t)n * nnodes);
    node[i] = w[m] * pt[(int64_t)n * npts + owner[m]];
}

template <typename T>
__global__ __launch_bounds__(256) void gather_series_kernel(const T *node, T *pt, const int *pt_start, const T *w, int nt,
                                                            int npts, int nnodes) {
    const int64_t pair = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;  // (n, p), 8 lanes each
    const int sub = threadIdx.x & 7;
    T v = T(0);
    const bool live = pair < (int64_t)nt * npts;
    int n = 0, p = 0;
    if (live) {
        n = (int)(pair / npts);
        p = (int)(pair - (int64_t)n * npts);
        const int m = pt_start[p] + sub;
        if (m < pt_start[p + 1]) v = w[m] * node[(int64_t)n * nnodes + m];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) v += __shfl_down(v, off, 8);
    if (live && sub == 0) pt[pair] = v;
}

template <typename T>
hipError_t launch_scatter_series(const T *pt, T *node, const int *owner, const T *w, int nt, int npts, int nnodes,
                                 hipStream_t s) {
    const int64_t tot = (int64_t)nt * nnodes;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_series_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, pt, node, owner, w,
                       nt, npts, nnodes);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_gather_series(const T *node, T *pt, const int *pt_start, const T *w, int nt, int npts, int nnodes,
                                hipStream_t s) {
    const int64_t tot = (int64_t)nt * npts * 8;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_series_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, node, pt, pt_start,
                       w, nt, npts, nnodes);
    return hipGetLastError();
}

// Model-sized vector algebra for the optimiser (device-resident L-BFGS state).
template <typename T>
__global__ void axpby_kernel(T *y, double a, const T *x, double b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (T)(a * (double)x[i] + b * (double)y[i]);
}

template <typename T>
__global__ void clip_kernel(T *x, double lo, double hi, int64_t n, int nx, int cx) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (cx == nx || (int)(i % cx) < nx) x[i] = (T)fmin(fmax((double)x[i], lo), hi);
}

// *out = max(*out, max |x|): wave64 shuffle max, LDS across waves, one atomic per block.  The
// bit pattern of a non-negative double orders like an unsigned integer.
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T *x, int64_t n, unsigned long long *out) {
    double m = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, fabs((double)x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}

static inline int vec_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n + 255) / 256)); }

template <typename T>
hipError_t launch_axpby(T *y, double a, const T *x, double b, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(axpby_kernel<T>, dim3(vec_blocks(n)), dim3(256), 0, s, y, a, x, b, n);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_clip(const GridDesc &g, T *x, double lo, double hi, hipStream_t s) {
    hipLaunchKernelGGL(clip_kernel<T>, dim3(vec_blocks(g.npts)), dim3(256), 0, s, x, lo, hi, g.npts, g.nx, g.cx);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_absmax(const T *x, int64_t n, double *out, hipStream_t s) {
    hipLaunchKernelGGL(absmax_kernel<T>, dim3(vec_blocks(n)), dim3(256), 0, s, x, n, (unsigned long long *)out);
    return hipGetLastError();
}

#define FWI_INSTANTIATE(T)                                                                          \
    template hipError_t launch_record<T>(const T *, const int64_t *, T *, T, int, hipStream_t);    \
    template hipError_t launch_image<T>(const GridDesc &, const T *, const T *, T *, int, hipStream_t); \
    template hipError_t launch_source_image<T>(const T *, const T *, const int64_t *, const T *, T *, int, int, int, T, \
                                               hipStream_t);                                        \
    template hipError_t launch_finalize_gradient<T>(const GridDesc &, const T *, const T *, T *, double, int, \
                                                    hipStream_t);                                  \
    template hipError_t launch_repack<T>(const GridDesc &, T *, const T *, int, hipStream_t);       \
    template hipError_t launch_dot<T>(const T *, const T *, int64_t, double *, hipStream_t);         \
    template hipError_t launch_count_dirty_padding<T>(const GridDesc &, const T *, unsigned long long *, hipStream_t); \
    template hipError_t launch_residual_l2<T>(const T *, T *, int64_t, double *, hipStream_t);       \
    template hipError_t launch_scatter_series<T>(const T *, T *, const int *, const T *, int, int, int, hipStream_t); \
    template hipError_t launch_gather_series<T>(const T *, T *, const int *, const T *, int, int, int, hipStream_t);  \
    template hipError_t launch_build_model<T>(const GridDesc &, const T *, T *, double, int *, hipStream_t); \
    template hipError_t launch_axpby<T>(T *, double, const T *, double, int64_t, hipStream_t);         \
    template hipError_t launch_clip<T>(const GridDesc &, T *, double, double, hipStream_t);           \
    template hipError_t launch_absmax<T>(const T *, int64_t, double *, hipStream_t);
FWI_INSTANTIATE(float)
FWI_INSTANTIATE(double)

}  // namespace fwi
