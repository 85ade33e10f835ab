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
#include "fwi_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace fwi {

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

GridDesc make_grid(int ndim, int nz, int ny, int nx, int order) {
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
    // FWI_XPITCH_EXTRA (floats) is the A/B hook.
    g.sy = HALO + round_up(nx, XALIGN);
    if (const char *e = getenv("FWI_XPITCH_EXTRA")) g.sy += std::max(0, atoi(e) / 4 * 4);
    // (rows whose interior starts on a 128-byte line -- pitch nx + 32 with the gap shared as right / left halo -- were
    // measured as well: 256^3 +2 %, 512^3 +1 %, 640^3 -8 %: not adopted)
    // (sharing the y-halo rows between consecutive planes the same way was measured: within noise, 512^3 349 vs 352)
    const int64_t py = (ndim == 3) ? HALO + round_up(g.ny, YALIGN) + HALO : 1;
    g.sz = g.sy * py;
    g.off0 = (int64_t)HALO * g.sz + (int64_t)hy * g.sy + HALO;
    // 2-D: rows are the tiled axis of step2d_tile, so they are rounded like y is in 3-D.
    // 3-D: LOOKAHEAD extra zero planes behind the far z halo, so the stream kernel's prefetches of
    // planes z + r + 1 ... need no clamping (affine addresses: the plane offsets strength-reduce).
    g.ptot = g.sz * (int64_t)(((ndim == 2) ? round_up(nz, YALIGN) : nz + LOOKAHEAD) + 2 * HALO) + HALO + 2 * 256;  // (+ the last row's right halo, and
                                        // the edge loads of a 256-column tile whose row ends early: values never used)
    g.cx = (int)round_up(nx, 4);
    g.npts = (int64_t)nz * g.ny * g.cx;
    return g;
}

// Source / residual injection into the points a workgroup has just written.  The host sorts a tile's entries by
// node, so the entries of one node are consecutive: the thread of a run's FIRST entry adds the whole run, in entry
// order, and issues ONE add per node -- duplicate nodes (two sources on one node) sum in a fixed order:
// bit-reproducible.  A node belongs to exactly one tile, so no other workgroup touches it; the caller's barrier has
// drained this workgroup's own stores of u'.
template <typename T, bool INC, bool Q>
__device__ __forceinline__ void inject_runs(const StepArgs<T> &a, int s0, int s1, int tid, int nth) {
    for (int i = s0 + tid; i < s1; i += nth) {
        // entry i's operands all at once (one level of loads, then the amplitude): the common run of 1 costs what
        // the plain per-entry form did
        const int run = a.inj_run[i];  // host-made: length of the node's run at its first entry, 0 at the others
        const int64_t p = a.inj_pidx[i];
        const T amp0 = a.inj_amp[a.inj_col[i]];
        T su = a.inj_cu[i] * amp0, sq = Q ? a.inj_cq[i] * amp0 : T(0);
        if (run == 0) continue;  // a later entry of its node's run: the first one's thread adds it
        for (int j = i + 1; j < i + run; ++j) {
            const T amp = a.inj_amp[a.inj_col[j]];
            su += a.inj_cu[j] * amp;
            if (Q) sq += a.inj_cq[j] * amp;
        }
        // ONE add per node and field: atomics only because they need no round trip (a plain read-modify-write would
        // put an L2 latency at the very end of the kernel); with one add per address the result is order-free
        atomicAdd(a.u_prev + p, su);
        if (INC) atomicAdd(a.v + p, su);  // the source moves u' and v' alike
        if (Q) atomicAdd(a.q_out + a.inj_cidx[i], sq);
    }
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
        if (IMAGE) a.g[ci] += a.q_in2 ? fma(up, a.q_in2[ci], uc * a.q_in[ci]) : uc * a.q_in[ci];
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

// ---------------------------------------------------------------------------
// STREAM kernel (3-D, fp32 and fp64): a workgroup of 64 x TY threads owns a
// (64 VL) x TY (x, y) tile -- VL = 4 floats or 2 doubles per lane -- and marches
// `zchunk` planes in z.  Each thread owns one 16-byte vector (VL consecutive x)
// per plane: every global access is a coalesced 16 B/lane stream, a wave
// covers 1 KiB of one row.
//   z neighbours : register queue of 2r+2 vectors, the next plane fetched
//                  straight into it one plane ahead; z loop unrolled by the
//                  queue length so all indices are static (no rotation).
//   y neighbours : the current plane's TY + 2r rows staged in LDS (double
//                  buffered, so ONE barrier per plane); the 2r halo rows are
//                  fetched r planes ahead (right behind the neighbouring tile
//                  streaming them) by the waves themselves.
//   x neighbours : the row's left/right vectors read back from the same LDS
//                  row (lanes 0-3 / 60-63 add the 4 edge elements per side).
// Algorithmic traffic 16 B/point in fp32 (u_cur, u_prev, C in; u_next out);
// the halo re-reads ((TY+2r)/TY in y, (zchunk+2r)/zchunk in z) are L2 /
// Infinity Cache traffic.  Blocks are renumbered so each XCD (private 4 MiB
// L2) owns a contiguous slab of tiles and shares those halo rows on chip.
// ---------------------------------------------------------------------------
// 16-byte vector of the field type: float4 / double2.  One per lane = 1 KiB per wave-instruction.
template <typename T> struct VecOf;
template <> struct VecOf<float> {
    static constexpr int VL = 4;
    typedef float nt_t __attribute__((ext_vector_type(4)));
};
template <> struct VecOf<double> {
    static constexpr int VL = 2;
    typedef double nt_t __attribute__((ext_vector_type(2)));
};
template <typename T>
struct alignas(16) vec {
    T v[VecOf<T>::VL];
};
using f4 = vec<float>;

template <typename T>
__device__ __forceinline__ vec<T> ldv(const T *p) { return *reinterpret_cast<const vec<T> *>(p); }
template <typename T>
__device__ __forceinline__ void stv(T *p, const vec<T> &v) { *reinterpret_cast<vec<T> *>(p) = v; }

// Streaming (non-temporal) forms for the once-per-step traffic of the imaging term q: it is
// written once in the forward pass and read once in the adjoint pass, tens of GiB per shot, and
// must not evict the wavefields from L2 / Infinity Cache.
template <typename T>
__device__ __forceinline__ vec<T> ldv_stream(const T *p) {
    typedef typename VecOf<T>::nt_t nt_t;
    const nt_t v = __builtin_nontemporal_load(reinterpret_cast<const nt_t *>(p));
    vec<T> r;
#pragma unroll
    for (int j = 0; j < VecOf<T>::VL; ++j) r.v[j] = v[j];
    return r;
}
template <typename T>
__device__ __forceinline__ void stv_stream(T *p, const vec<T> &f) {
    typedef typename VecOf<T>::nt_t nt_t;
    nt_t v;
#pragma unroll
    for (int j = 0; j < VecOf<T>::VL; ++j) v[j] = f.v[j];
    __builtin_nontemporal_store(v, reinterpret_cast<nt_t *>(p));
}
// Forward-term store in bf16 (fwi_config.store_dtype): 4 values = 8 bytes per lane, round to nearest even on the
// way out, exact on the way in.  bf16 keeps fp32's exponent, so the term needs no scaling.  The conversion is the
// plain cast (v_cvt_pk_bf16_f32): the integer-rounding form (u + 0x7fff + lsb) >> 16 turns some NaNs into 0 or
// infinity (MI355X_MICROARCH.md, correctness boundaries), which would launder a blown-up forward run into a
// finite-looking gradient; the cast keeps every NaN a NaN and rounds finite values identically.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void st_bf16x4_stream(void *base, int64_t elem, const vec<float> &q) {
    u32x2 w;
    w[0] = bf16_pack2(q.v[0], q.v[1]);
    w[1] = bf16_pack2(q.v[2], q.v[3]);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(base) + elem));
}
__device__ __forceinline__ vec<float> ld_bf16x4_stream(const void *base, int64_t elem) {
    const u32x2 w = __builtin_nontemporal_load(
        reinterpret_cast<const u32x2 *>(reinterpret_cast<const unsigned short *>(base) + elem));
    vec<float> q;
    q.v[0] = __uint_as_float(w[0] << 16);
    q.v[1] = __uint_as_float(w[0] & 0xffff0000u);
    q.v[2] = __uint_as_float(w[1] << 16);
    q.v[3] = __uint_as_float(w[1] & 0xffff0000u);
    return q;
}
template <typename T>
__device__ __forceinline__ void st_q(T *base, int64_t elem, const vec<T> &q, bool) { stv_stream<T>(base + elem, q); }
template <typename T>
__device__ __forceinline__ vec<T> ld_q(const T *base, int64_t elem, bool) { return ldv_stream<T>(base + elem); }
template <bool QB>
__device__ __forceinline__ void st_qf(float *base, int64_t elem, const vec<float> &q) {
    if (QB) st_bf16x4_stream(base, elem, q); else stv_stream<float>(base + elem, q);
}
template <bool QB>
__device__ __forceinline__ vec<float> ld_qf(const float *base, int64_t elem) {
    return QB ? ld_bf16x4_stream(base, elem) : ldv_stream<float>(base + elem);
}
template <bool QB>
__device__ __forceinline__ void st_qf(double *base, int64_t elem, const vec<double> &q) { stv_stream<double>(base + elem, q); }
template <bool QB>
__device__ __forceinline__ vec<double> ld_qf(const double *base, int64_t elem) { return ldv_stream<double>(base + elem); }

__device__ __forceinline__ f4 ld4(const float *p) { return ldv<float>(p); }
__device__ __forceinline__ void st4(float *p, const f4 &v) { stv<float>(p, v); }
__device__ __forceinline__ f4 ld4_stream(const float *p) { return ldv_stream<float>(p); }
__device__ __forceinline__ void st4_stream(float *p, const f4 &v) { stv_stream<float>(p, v); }

// 1 / a for a in [1, 2): hardware reciprocal (1 ulp) + one Newton step, ~0.5 ulp; replaces the
// ~12-instruction IEEE division in the damping factor A = 1 / (1 + d).
__device__ __forceinline__ float rcp_nr(float a) {
    const float r = __builtin_amdgcn_rcpf(a);
    return r * fmaf(-a, r, 2.f);
}
__device__ __forceinline__ double rcp_nr(double a) { return 1.0 / a; }

constexpr int TILE_X = 256;                // 2-D tile kernel: floats per tile row = 64 lanes x float4
constexpr int LROW4 = TILE_X / 4 + 2;      // its LDS row in float4: [left edge][64][right edge]

// IMAGE: 0 = off, 1 = g += u_cur * q_in, 2 = additionally g += u_prev * q_in2 (two time levels per
// read-modify-write of g: the adjoint sweep is HBM-bound, this takes it from 28 to 24 B/update).
// XP: the convolutional PML of the x border carried in the lanes (1 = forward recursion, 2 = its transpose): the
// border cells of a row are the first / last npml / 4 lanes of the wave that owns it, psi' of the neighbouring cells
// comes through __shfl_up / __shfl_down (the 2 r-wide dependence of the border recursion never leaves the wave),
// D u and E_x u from the x window the stencil has in registers anyway.  The memory variables are read and written
// once per step by the lanes that own them -- no slab launches for this axis, no second pass over u' and q.
// ZP (with XP: the same direction): the z border's recursion rides on the z march.  The queue runs r planes further
// ahead (planes z - r .. z + 2r), so psi' of plane z + r -- which needs u(z .. z + 2r) -- is formed r planes before the
// update of plane z reads psi'(z - r .. z + r); those 2r + 1 planes of psi' wait in a per-thread LDS ring (no thread
// reads another's slot: no barrier).  The adjoint recursion keeps a zt' = alpha / a ring over z - r .. z + 2r and a
// beta = a pt' ring over z - r .. z + r.  All of it sits under one wave-uniform test of the plane index -- planes
// further than 2r from a border pay nothing but two clamped loads -- and holds no global LOAD (the pipelined loop keeps
// its counted vmcnt).  Chunk seams must stay 2r planes clear of the borders (stream_zpml_supported).
template <typename T, int R, int TY, bool DAMP, bool SAVE_Q, int IMAGE, bool FULL, int PF, bool INC = false,
          bool QB = false, int XPM = 0, bool ZP = false>
__global__ __launch_bounds__(64 * TY) void step3d_stream(StepArgs<T> a, GridDesc g, int zchunk,
                                                         int nxt, int nyt, int nblk, int tw) {
    // XPM 1 / 2: the x border in the lanes, forward / adjoint; 3 / 4: the same for a border whose width is not a
    // multiple of the lane vector (npml = 10, 14, ...): the one lane per side that straddles the border's inner edge
    // stores its memory variables cell by cell (the slab row holds the other side's cells right behind)
    constexpr int XP = XPM == 0 ? 0 : ((XPM - 1) & 1) + 1;
    constexpr bool XMASK = XPM > 2;
    constexpr int NH = (2 * R + TY - 1) / TY;  // halo rows each wave fetches per plane
    constexpr int TRASH = TY + 2 * R;          // LDS row that absorbs the writes of idle slots
    constexpr int LROWS = TY + 2 * R + 1;
    constexpr int VL = VecOf<T>::VL;       // elements per lane: float4 / double2
    constexpr int HV = HALO / VL;          // vectors per x-halo side (1 / 2)
    constexpr int TX = 64 * VL;            // tile row in elements (256 / 128)
    constexpr int LROWV = 64 + 2 * HV;     // LDS row in vectors: [left edge][64 lanes][right edge]
    using V = vec<T>;
    __shared__ V lds[2][LROWS][LROWV];

    const int lane = threadIdx.x, ty = threadIdx.y;
    int bid = blockIdx.x;
    if (bid >= nblk) {
        // Receiver sampling rides along as extra workgroups: u_cur is read-only in this
        // launch and already holds the previous step's field including its injection.
        for (int i = (bid - nblk) * (64 * TY) + ty * 64 + lane; i < a.nrec; i += (gridDim.x - nblk) * 64 * TY)
            a.rec_out[i] = a.u_cur[a.rec_pidx[i]] * a.rec_scale;
        return;
    }
    if (zchunk > 0) {  // XCD-contiguous slabs: XCD x runs blocks x, x + 8, ... -> give it one contiguous range
        const int x = bid & 7, q = nblk >> 3, r = nblk & 7;  // the first r XCDs hold q + 1 blocks
        bid = x * q + min(x, r) + (bid >> 3);
    }
    zchunk = abs(zchunk);  // (a negative zchunk is the tuning hook that switches the renumbering off)
    const int bx = bid % nxt;
    const int t2 = bid / nxt;
    const int by = t2 % nyt, bz = t2 / nyt;
    // x tiles are `tw` columns wide (a multiple of VL, <= TX): the host splits nx into equal tiles, so
    // a grid of 384 columns runs as 2 x 192 (48 lanes each) instead of 256 + 128 -- same idle lanes,
    // but every workgroup then moves the same bytes and none is the straggler (384^3: 224 -> 269
    // Gpts/s).  FULL: tw == TX.
    const int twid = FULL ? TX : tw;
    const int x0 = bx * twid + VL * lane;
    const int y0 = by * TY, y = y0 + ty;
    const int z0 = bz * zchunk;
    const int z1 = min(g.nz, z0 + zchunk);
    const bool act = FULL || ((VL * lane < twid) && (x0 < g.nx) && (y < g.ny));
    const int64_t sz = g.sz, sy = g.sy;
    // Lanes that own no points store nothing, but the loop is branch-free, so they load.  The first
    // HALO columns right of the tile are the x halo of its last lanes: those lanes load the real field
    // there (which makes the right-edge scalars below redundant unless tw == TX).  Lanes further right
    // repeat the last halo vector, and lanes right of the GRID all read the one all-zero vector just
    // behind the row's last data vector (pad, never written): the same cache lines as real data instead
    // of a stream of pad lines, and exactly the zeros the neighbouring lanes' x stencil must see.
    const int xa = FULL ? x0 : min(min(x0, bx * twid + twid + HALO - VL), ((g.nx - 1) / VL + 1) * VL);

    // Addressing: wave-uniform 64-bit plane base (SGPRs) + per-thread 32-bit
    // in-plane offset (one VGPR shared by u_cur, u_prev and C, which have the
    // same padded layout).
    const unsigned poff = (unsigned)(g.off0 + (int64_t)y * sy + xa);  // (z = 0, y, x0)
    const unsigned coff = (unsigned)((int64_t)y * g.cx + x0);         // compact, z = 0
    const int64_t cplane = (int64_t)g.ny * g.cx;

    // The loop below is free of divergent branches so that hipcc can count its
    // s_waitcnt vmcnt(N) exactly and keep the prefetches in flight across
    // iterations: slots with nothing to fetch re-load their own address and
    // write to the TRASH row instead of branching.
    // Halo rows: hr in [0, 2R), below (hr < R) or above the tile.
    unsigned hoff[NH];
    int hrow[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int hr = ty + i * TY;
        const bool valid = hr < 2 * R;
        const int yh = !valid ? y : (hr < R) ? y0 - R + hr : y0 + TY + (hr - R);
        hrow[i] = !valid ? TRASH : (hr < R) ? hr : TY + hr;
        hoff[i] = (unsigned)(g.off0 + (int64_t)yh * sy + xa);
    }
    // x edges: lanes 0-3 fetch the 4 elements left of the tile row, lanes 60-63
    // the 4 elements right of it (one scalar load each).
    const unsigned rowoff = (unsigned)(g.off0 + (int64_t)y * sy + bx * twid);
    // (idle lanes 4..59 all repeat lane 3's address: one cache line per wave instead of the eight their own
    // vectors span -- those re-loads were a quarter of the kernel's L2 requests)
    const unsigned eoff = (lane < 4) ? rowoff - 4 + lane : (lane >= 60) ? rowoff + TX + (lane - 60) : rowoff - 1;
    const int erow = (lane < 4 || lane >= 60) ? R + ty : TRASH;
    // idle lanes write consecutive words of the trash row (a 16 B stride would be a 4-way bank conflict)
    const int ecol = (lane < 4) ? lane : (lane >= 60) ? (HV + 64) * VL + (lane - 60) : lane;

    // loop-invariant xy part of the damping
    V Axy, Bxy, dxy;
    if (DAMP) {
        const T dyv = (y < g.ny) ? a.dy[y] : T(0);
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const T d = dyv + ((x0 + j < g.nx) ? a.dx[x0 + j] : T(0));
            dxy.v[j] = d;
            Bxy.v[j] = T(1) - d;
            Axy.v[j] = rcp_nr(T(1) + d);
        }
    }

    // x-border CPML in the lanes: a(x), b(x) of this lane's cells (0 off the border, so that psi and zeta vanish
    // there whatever was loaded), the lane's offset in a row of the memory-variable arrays (nz, ny, 2 npml)
    V xca, xcb;
    unsigned xld = 0;
    bool xin = false;
    unsigned xcells = 0;  // (XMASK) which of the lane's cells are border cells
    int64_t xplane = 0;
    if constexpr (XP != 0) {
        const int nsl = 2 * a.npml;
        xplane = (int64_t)g.ny * nsl;
        bool in = false;
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int x = x0 + j;
            const bool b = x < g.nx && (x < a.npml || x >= g.nx - a.npml);
            xca.v[j] = b ? a.xp_a[x] : T(0);
            xcb.v[j] = b ? a.xp_b[x] : T(0);
            in |= b;
            xcells |= b ? 1u << j : 0u;
        }
        xin = act && in;  // (XMASK off: npml and nx are multiples of VL, a lane's cells are all in the border or all out)
        const int jx = x0 < a.npml ? x0 : x0 - (g.nx - nsl);
        xld = (unsigned)((int64_t)min(y, g.ny - 1) * nsl + (xin ? jx : 0));  // lanes off the border re-read column 0
    }

    // z register queue: plane p lives in slot (p - z0 + R) % NQ.  PF planes are fetched ahead of
    // use straight into the queue; the pointwise operands (u_prev, C), halo rows and edge pieces
    // sit in rings of PF + 1 slots.  NQ is a multiple of PF + 1 and the z loop is unrolled NQ
    // times, so every queue / ring index below is a compile-time constant: no register
    // rotations, and hipcc keeps the prefetches of planes z+1 .. z+PF in flight while plane z is
    // computed (exact vmcnt counts).  PF >= 2 matters at one wave per SIMD: one plane takes
    // ~0.6 us, less than a loaded L2-miss round trip.
    // The y-halo rows are fetched R planes ahead (ring of R + 1 slots, which divides NQ): that is
    // (almost) when the neighbouring tile, whose interior rows they are, streams the same plane
    // into its own z queue, so the second request hits in the XCD's L2 instead of going back to
    // HBM ~4 planes later (measured at 512^3: traffic 1.235x algorithmic before).
    constexpr int NR = PF + 1;
    constexpr int QE = ZP ? R : 0;  // planes the queue runs further ahead for the z border's recursion
    constexpr int NQ = (2 * R + 1 + QE + PF + NR - 1) / NR * NR;
    constexpr int HPF = (NQ % (R + 1) == 0) ? R : PF;  // halo prefetch distance
    constexpr int NRH = HPF + 1;
    V zq[NQ];
    // (ZP: the last prefetches would run one plane past the zero planes behind the grid: clamped onto the last one)
    const int zlast = g.nz + HALO + LOOKAHEAD - 1;
#pragma unroll
    for (int k = 0; k < 2 * R + QE + PF; ++k)
        zq[k] = ldv<T>(a.u_cur + (int64_t)(ZP ? min(z0 - R + k, zlast) : z0 - R + k) * sz + poff);
    V up[NR], Cc[NR], halo[NRH][NH];
    V xps[XP ? NR : 1], xzt[XP ? NR : 1];  // psi / zeta (adjoint: pt / zt) of the x border, fetched like up / Cc
    V zps[ZP ? NR : 1], zzt[ZP ? NR : 1];  // ... of the z border: psi(z + r) / zeta(z); adjoint pt(z + r) / zt(z + 2r)
    T edge[NR];
    // increment form: the pointwise operand is v^n (its own padded field) instead of u^{n-1}
    const T *const pw = INC ? a.v : a.u_prev;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int64_t o = (int64_t)(z0 + p) * sz;
        up[p] = ldv<T>(pw + o + poff);
        Cc[p] = ldv<T>(a.C + o + poff);
        edge[p] = a.u_cur[o + eoff];
        if constexpr (XP != 0) {
            xps[p] = ldv<T>(a.xp_psi + (int64_t)min(z0 + p, g.nz - 1) * xplane + xld);
            xzt[p] = ldv<T>(a.xp_zeta + (int64_t)min(z0 + p, g.nz - 1) * xplane + xld);
        }
    }
#pragma unroll
    for (int p = 0; p < HPF; ++p) {
        const int64_t o = (int64_t)(z0 + p) * sz;
#pragma unroll
        for (int i = 0; i < NH; ++i) halo[p][i] = ldv<T>(a.u_cur + o + hoff[i]);
    }


    // ---- z border in the march (ZP) ----------------------------------------------------------------------------
    // rings: forward psi' over planes z - r .. z + r; adjoint alpha = a zt' over z - r .. z + 2r, beta = a pt' over
    // z - r .. z + r.  Slot of plane p = (p - z0 + r) mod ring length; a thread touches only [.][ty][lane].
    constexpr int NSA = ZP ? (XP == 2 ? 3 * R + 1 : 2 * R + 1) : 1, NSB = (ZP && XP == 2) ? 2 * R + 1 : 1;
    __shared__ V zra[NSA][ZP ? TY : 1][ZP ? 64 : 1];
    __shared__ V zrb[NSB][(ZP && XP == 2) ? TY : 1][(ZP && XP == 2) ? 64 : 1];
    const unsigned zco = act ? coff : 0u;  // this thread's offset in a plane of the z memory variables (compact)
    // coefficient of plane p by its distance into the border (kernel arguments: scalar loads), 0 off the border
    auto zdist = [&](int p) { return (p >= 0 && p < g.nz) ? max(0, max(a.npml - p, p - (g.nz - 1 - a.npml))) : 0; };
    auto zslab = [&](int p) { return p < a.npml ? max(p, 0) : (p >= g.nz - a.npml ? min(p, g.nz - 1) - (g.nz - 2 * a.npml) : 0); };
    if constexpr (ZP) {
        const V zero = {};
        const bool lo_near = z0 < a.npml + 2 * R;  // (chunk seams are 2r clear of the borders: only chunk 0 starts inside one)
        if (XP == 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) zra[(unsigned)r % NSA][ty][lane] = zero;  // planes z0 - r .. z0 - 1
            // psi' of planes z0 .. z0 + r - 1 (the loop forms psi'(z + r) at plane z)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = z0 + r, d = zdist(p);
                V v = zero;
                if (lo_near && d > 0) {
                    const T av = a.zp_a[d - 1], bv = a.zp_b[d - 1];
                    const V old = ldv<T>(a.zp_psi + (int64_t)zslab(p) * cplane + zco);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T du = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) du = fma(a.xp_dk1[k], zq[r + R + k].v[j] - zq[r + R - k].v[j], du);
                        v.v[j] = fma(bv, old.v[j], av * du);
                    }
                    if (act) stv<T>(a.zp_psi + (int64_t)zslab(p) * cplane + zco, v);
                }
                zra[(unsigned)(r + R) % NSA][ty][lane] = v;
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                zps[i] = ldv<T>(a.zp_psi + (int64_t)zslab(z0 + R + i) * cplane + zco);
                zzt[i] = ldv<T>(a.zp_zeta + (int64_t)zslab(z0 + i) * cplane + zco);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                zra[(unsigned)r % NSA][ty][lane] = zero;
                zrb[(unsigned)r % NSB][ty][lane] = zero;
            }
            // alpha of planes z0 .. z0 + 2r - 1, then beta of planes z0 .. z0 + r - 1
#pragma unroll
            for (int r = 0; r < 2 * R; ++r) {
                const int p = z0 + r, d = zdist(p);
                V v = zero;
                if (lo_near && d > 0) {
                    const T av = a.zp_a[d - 1], bv = a.zp_b[d - 1];
                    V zt = ldv<T>(a.zp_zeta + (int64_t)zslab(p) * cplane + zco);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        zt.v[j] = fma(bv, zt.v[j], zq[r + R].v[j]);
                        v.v[j] = av * zt.v[j];
                    }
                    if (act) stv<T>(a.zp_zeta + (int64_t)zslab(p) * cplane + zco, zt);
                }
                zra[(unsigned)(r + R) % NSA][ty][lane] = v;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = z0 + r, d = zdist(p);
                V v = zero;
                if (lo_near && d > 0) {
                    const T av = a.zp_a[d - 1], bv = a.zp_b[d - 1];
                    V pt = ldv<T>(a.zp_psi + (int64_t)zslab(p) * cplane + zco);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T dw = T(0);  // D (mu + alpha)
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const V ap = zra[(unsigned)(r + R + k) % NSA][ty][lane], am = zra[(unsigned)(r + R - k) % NSA][ty][lane];
                            dw = fma(a.xp_dk1[k], (zq[r + R + k].v[j] + ap.v[j]) - (zq[r + R - k].v[j] + am.v[j]), dw);
                        }
                        pt.v[j] = bv * pt.v[j] - dw;
                        v.v[j] = av * pt.v[j];
                    }
                    if (act) stv<T>(a.zp_psi + (int64_t)zslab(p) * cplane + zco, pt);
                }
                zrb[(unsigned)(r + R) % NSB][ty][lane] = v;
            }
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                zps[i] = ldv<T>(a.zp_psi + (int64_t)zslab(z0 + R + i) * cplane + zco);
                zzt[i] = ldv<T>(a.zp_zeta + (int64_t)zslab(z0 + 2 * R + i) * cplane + zco);
            }
        }
    }

    for (int zb = z0; zb < z1; zb += NQ) {
#pragma unroll
        for (int ph = 0; ph < NQ; ++ph) {
            const int z = zb + ph;
            if (z >= z1) break;
            const int cur = ph % NR, nxt = (ph + PF) % NR;  // NQ % NR == 0: static across blocks
            // slot of plane z - R + k is (ph + k) % NQ
            const V &ctr = zq[(ph + R) % NQ];

            // stage the plane's rows in LDS (double buffered: one barrier per plane)
            V(*L)[LROWV] = lds[z & 1];
            L[R + ty][HV + lane] = ctr;
#pragma unroll
            for (int i = 0; i < NH; ++i) L[hrow[i]][HV + lane] = halo[ph % NRH][i];
            reinterpret_cast<T *>(&L[erow][0])[ecol] = edge[cur];

            // fetch plane z+PF's operands (and plane z+R+PF of the queue) while z is computed
            const int64_t on = (int64_t)(z + PF) * sz;
            zq[(ph + 2 * R + QE + PF) % NQ] =
                ldv<T>(a.u_cur + (int64_t)(ZP ? min(z + R + QE + PF, zlast) : z + R + PF) * sz + poff);
            {
                const int64_t oh = (int64_t)(z + HPF) * sz;
#pragma unroll
                for (int i = 0; i < NH; ++i) halo[(ph + HPF) % NRH][i] = ldv<T>(a.u_cur + oh + hoff[i]);
            }
            edge[nxt] = a.u_cur[on + eoff];
            // (plain loads: non-temporal hints on these read-once streams were measured and
            // rejected -- 256^3 39 -> 51 us/step, they defeat Infinity-Cache residency; 512^3 +-2 %)
            up[nxt] = ldv<T>(pw + on + poff);
            Cc[nxt] = ldv<T>(a.C + on + poff);
            if constexpr (XP != 0) {
                const int64_t ox = (int64_t)min(z + PF, g.nz - 1) * xplane + xld;
                xps[nxt] = ldv<T>(a.xp_psi + ox);
                xzt[nxt] = ldv<T>(a.xp_zeta + ox);
            }
            if constexpr (ZP) {  // (planes off the border re-read slab plane 0: one cache line per wave)
                zps[nxt] = ldv<T>(a.zp_psi + (int64_t)zslab(z + R + PF) * cplane + zco);
                zzt[nxt] = ldv<T>(a.zp_zeta + (int64_t)zslab(z + (XP == 2 ? 2 * R : 0) + PF) * cplane + zco);
            }
            V qi, qi2, gi;
            if (IMAGE) {
                const unsigned co = act ? coff : 0u;
                qi = ld_qf<QB>(a.q_in, (int64_t)z * cplane + co);
                if (IMAGE == 2) qi2 = ld_qf<QB>(a.q_in2, (int64_t)z * cplane + co);
                // g is touched once per (other) step: streaming hints keep it from evicting the three
                // wavefield arrays from the Infinity Cache (adjoint 68 -> 59 us/step at 256^3)
                gi = ldv_stream<T>(a.g + (int64_t)z * cplane + co);
            }
            __syncthreads();

            // X = the HALO elements left of this lane's vector, the vector, the HALO right of it
            T X[2 * HALO + VL];
#pragma unroll
            for (int h = 0; h < HV; ++h) {
                const V xl = L[R + ty][lane + h], xr = L[R + ty][HV + lane + 1 + h];
#pragma unroll
                for (int j = 0; j < VL; ++j) {
                    X[h * VL + j] = xl.v[j];
                    X[HALO + VL + h * VL + j] = xr.v[j];
                }
            }
#pragma unroll
            for (int j = 0; j < VL; ++j) X[HALO + j] = ctr.v[j];
            V lap, ezv;  // (ezv: the z second difference alone, for the z border's zeta)
#pragma unroll
            for (int j = 0; j < VL; ++j) lap.v[j] = ezv.v[j] = T(0);
#pragma unroll
            for (int k = R; k >= 1; --k) {
                const V ym = L[R + ty - k][HV + lane], yp = L[R + ty + k][HV + lane];
                const V &zm = zq[(ph + R - k) % NQ], &zp = zq[(ph + R + k) % NQ];
                const T c = a.ck[k];
#pragma unroll
                for (int j = 0; j < VL; ++j) {
                    T t = (X[HALO + j - k] + X[HALO + j + k]) + (ym.v[j] + yp.v[j]) + (zm.v[j] + zp.v[j]);
                    t = fma(T(-6), X[HALO + j], t);
                    lap.v[j] = fma(c, t, lap.v[j]);
                    if (ZP && XP == 1) ezv.v[j] = fma(c, fma(T(-2), X[HALO + j], zm.v[j] + zp.v[j]), ezv.v[j]);
                }
            }
            // x-border CPML: this lane's cells of the border recursion, neighbours through the wave
            V xterm, xn0, xn1;
            if constexpr (XP != 0) {
                T W[2 * HALO + VL];  // [left lane's cells][own][right lane's] of a quantity, zero beyond the wave
                auto spread = [&](const V &v) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        const T l = __shfl_up(v.v[j], 1, 64), r = __shfl_down(v.v[j], 1, 64);
                        W[j] = lane == 0 ? T(0) : l;
                        W[HALO + j] = v.v[j];
                        W[HALO + VL + j] = lane == 63 ? T(0) : r;
                    }
                };
                const V m0 = xps[cur], m1 = xzt[cur];
                if (XP == 1) {
                    // psi' = b psi + a D u;  zeta' = b zeta + a (E_x u + D psi');  term = D psi' + zeta'
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T du = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) du = fma(a.xp_dk1[k], X[HALO + j + k] - X[HALO + j - k], du);
                        xn0.v[j] = fma(xcb.v[j], m0.v[j], xca.v[j] * du);
                    }
                    spread(xn0);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T dp = T(0), ex = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            dp = fma(a.xp_dk[k], W[HALO + j + k] - W[HALO + j - k], dp);
                            ex = fma(a.ck[k], fma(T(-2), X[HALO + j], X[HALO + j - k] + X[HALO + j + k]), ex);
                        }
                        xn1.v[j] = fma(xcb.v[j], m1.v[j], xca.v[j] * (ex + dp));
                        xterm.v[j] = dp + xn1.v[j];
                    }
                } else {
                    // zt' = b zt + mu;  pt' = b pt - D mu - D (a zt');  term = E_x (a zt') - D (a pt')
                    V al;
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        xn1.v[j] = fma(xcb.v[j], m1.v[j], X[HALO + j]);
                        al.v[j] = xca.v[j] * xn1.v[j];
                    }
                    spread(al);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T d = T(0), e2 = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            d = fma(a.xp_dk1[k], (X[HALO + j + k] - X[HALO + j - k]) + (W[HALO + j + k] - W[HALO + j - k]), d);
                            e2 = fma(a.ck[k], fma(T(-2), W[HALO + j], W[HALO + j + k] + W[HALO + j - k]), e2);
                        }
                        xn0.v[j] = xcb.v[j] * m0.v[j] - d;
                        xterm.v[j] = e2;
                        al.v[j] = xca.v[j] * xn0.v[j];  // (now beta = a pt')
                    }
                    spread(al);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        T db = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) db = fma(a.xp_dk[k], W[HALO + j + k] - W[HALO + j - k], db);
                        xterm.v[j] -= db;
                    }
                }
            }

            // z-border CPML on the march (see the template comment): everything under one wave-uniform test
            V zterm;
#pragma unroll
            for (int j = 0; j < VL; ++j) zterm.v[j] = T(0);
            if constexpr (ZP) {
                const unsigned s0 = (unsigned)(z - z0);  // ring slot of plane z - r (slot of plane p: p - z0 + r)
                const bool near = XP == 1 ? (z < a.npml + R || z + R >= g.nz - a.npml)
                                          : (z < a.npml + R || z + 2 * R >= g.nz - a.npml);
                if (XP == 1) {
                    const int pn = z + R;  // the plane whose psi' is formed now
                    V pnv;
#pragma unroll
                    for (int j = 0; j < VL; ++j) pnv.v[j] = T(0);
                    const int dn = zdist(pn), dz = zdist(z);
                    if (near && dn > 0) {
                        const T an = a.zp_a[dn - 1], bn = a.zp_b[dn - 1];
                        const V old = zps[cur];
#pragma unroll
                        for (int j = 0; j < VL; ++j) {
                            T du = T(0);
#pragma unroll
                            for (int k = 1; k <= R; ++k)
                                du = fma(a.xp_dk1[k], zq[(ph + 2 * R + k) % NQ].v[j] - zq[(ph + 2 * R - k) % NQ].v[j], du);
                            pnv.v[j] = fma(bn, old.v[j], an * du);
                        }
                        if (pn < z1 && act) stv<T>(a.zp_psi + (int64_t)zslab(pn) * cplane + zco, pnv);
                    }
                    zra[(s0 + 2 * R) % NSA][ty][lane] = pnv;
                    if (near) {
                        const T az = dz ? a.zp_a[max(dz, 1) - 1] : T(0), bz = dz ? a.zp_b[max(dz, 1) - 1] : T(0);
                        V dp;
#pragma unroll
                        for (int j = 0; j < VL; ++j) dp.v[j] = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const V pp = zra[(s0 + R + k) % NSA][ty][lane], pm = zra[(s0 + R - k) % NSA][ty][lane];
#pragma unroll
                            for (int j = 0; j < VL; ++j) dp.v[j] = fma(a.xp_dk[k], pp.v[j] - pm.v[j], dp.v[j]);
                        }
                        V zn;
                        const V zo = zzt[cur];
#pragma unroll
                        for (int j = 0; j < VL; ++j) {
                            zn.v[j] = fma(bz, zo.v[j], az * (ezv.v[j] + dp.v[j]));
                            zterm.v[j] = dp.v[j] + zn.v[j];
                        }
                        if (dz > 0 && act) stv<T>(a.zp_zeta + (int64_t)zslab(z) * cplane + zco, zn);
                    }
                } else {
                    const int p2 = z + 2 * R, pn = z + R;
                    V al2, be;
#pragma unroll
                    for (int j = 0; j < VL; ++j) al2.v[j] = be.v[j] = T(0);
                    const int d2 = zdist(p2), dn = zdist(pn);
                    if (near && d2 > 0) {  // zt'(z + 2r) = b zt + mu;  alpha = a zt'
                        const T a2 = a.zp_a[d2 - 1], b2 = a.zp_b[d2 - 1];
                        V zt = zzt[cur];
#pragma unroll
                        for (int j = 0; j < VL; ++j) {
                            zt.v[j] = fma(b2, zt.v[j], zq[(ph + 3 * R) % NQ].v[j]);
                            al2.v[j] = a2 * zt.v[j];
                        }
                        if (p2 < z1 && act) stv<T>(a.zp_zeta + (int64_t)zslab(p2) * cplane + zco, zt);
                    }
                    zra[(s0 + 3 * R) % NSA][ty][lane] = al2;
                    if (near && dn > 0) {  // pt'(z + r) = b pt - D (mu + alpha);  beta = a pt'
                        const T an = a.zp_a[dn - 1], bn = a.zp_b[dn - 1];
                        V pt = zps[cur], dw;
#pragma unroll
                        for (int j = 0; j < VL; ++j) dw.v[j] = T(0);
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const V ap = zra[(s0 + 2 * R + k) % NSA][ty][lane], am = zra[(s0 + 2 * R - k) % NSA][ty][lane];
#pragma unroll
                            for (int j = 0; j < VL; ++j)
                                dw.v[j] = fma(a.xp_dk1[k], (zq[(ph + 2 * R + k) % NQ].v[j] + ap.v[j]) -
                                                               (zq[(ph + 2 * R - k) % NQ].v[j] + am.v[j]), dw.v[j]);
                        }
#pragma unroll
                        for (int j = 0; j < VL; ++j) {
                            pt.v[j] = bn * pt.v[j] - dw.v[j];
                            be.v[j] = an * pt.v[j];
                        }
                        if (pn < z1 && act) stv<T>(a.zp_psi + (int64_t)zslab(pn) * cplane + zco, pt);
                    }
                    zrb[(s0 + 2 * R) % NSB][ty][lane] = be;
                    if (near) {  // term = E (alpha) - D (beta)
                        const V a0 = zra[(s0 + R) % NSA][ty][lane];
#pragma unroll
                        for (int k = 1; k <= R; ++k) {
                            const V ap = zra[(s0 + R + k) % NSA][ty][lane], am = zra[(s0 + R - k) % NSA][ty][lane];
                            const V bp = zrb[(s0 + R + k) % NSB][ty][lane], bm = zrb[(s0 + R - k) % NSB][ty][lane];
#pragma unroll
                            for (int j = 0; j < VL; ++j) {
                                zterm.v[j] = fma(a.ck[k], fma(T(-2), a0.v[j], ap.v[j] + am.v[j]), zterm.v[j]);
                                zterm.v[j] = fma(-a.xp_dk[k], bp.v[j] - bm.v[j], zterm.v[j]);
                            }
                        }
                    }
                }
            }

            V A = Axy, B = Bxy;
            if (DAMP) {
                // d_z(z) from the plane index: a load here would be a VECTOR load (the compiler
                // cannot prove the profile is not aliased by the u_next stores) whose wait
                // drains every prefetch in flight
                const int dist = max(0, max(a.npml - z, z - (g.nz - 1 - a.npml)));
                if (dist != 0) {
                    const T dzv = a.dz_scale * (T)(dist * dist);
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        const T d = dxy.v[j] + dzv;
                        B.v[j] = T(1) - d;
                        A.v[j] = rcp_nr(T(1) + d);
                    }
                }
            }
            V q, un, vn;
#pragma unroll
            for (int j = 0; j < VL; ++j) {
                q.v[j] = XP ? Cc[cur].v[j] * (lap.v[j] + (ZP ? xterm.v[j] + zterm.v[j] : xterm.v[j]))
                            : Cc[cur].v[j] * lap.v[j];
                if (INC) {  // v' = A (B v + q), u' = u + v'
                    vn.v[j] = DAMP ? fma(B.v[j], up[cur].v[j], q.v[j]) * A.v[j] : up[cur].v[j] + q.v[j];
                    un.v[j] = X[HALO + j] + vn.v[j];
                } else if (DAMP)
                    un.v[j] = (fma(T(2), X[HALO + j], -B.v[j] * up[cur].v[j]) + q.v[j]) * A.v[j];
                else
                    un.v[j] = (T(2) * X[HALO + j] - up[cur].v[j]) + q.v[j];
            }
            if constexpr (XP != 0) {
                if (XMASK && xin && xcells != (1u << VL) - 1u) {  // the lane astride the border's inner edge
#pragma unroll
                    for (int j = 0; j < VL; ++j)
                        if (xcells >> j & 1u) {
                            a.xp_psi[(int64_t)z * xplane + xld + j] = xn0.v[j];
                            a.xp_zeta[(int64_t)z * xplane + xld + j] = xn1.v[j];
                        }
                } else if (xin) {
                    stv<T>(a.xp_psi + (int64_t)z * xplane + xld, xn0);
                    stv<T>(a.xp_zeta + (int64_t)z * xplane + xld, xn1);
                }
            }
            if (act) {
                if (INC) stv<T>(a.v + (int64_t)z * sz + poff, vn);
                stv<T>(a.u_prev + (int64_t)z * sz + poff, un);
                if (SAVE_Q) st_qf<QB>(a.q_out, (int64_t)z * cplane + coff, q);
                if (IMAGE) {
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        gi.v[j] = fma(X[HALO + j], qi.v[j], gi.v[j]);
                        if (IMAGE == 2) gi.v[j] = fma(up[cur].v[j], qi2.v[j], gi.v[j]);
                    }
                    stv_stream<T>(a.g + (int64_t)z * cplane + coff, gi);
                }
            }
        }
    }

    // Source / residual injection into the points this workgroup has just written.
    // Wave-uniform and almost always empty; the barrier (which drains this workgroup's
    // stores, vmcnt(0)) orders the float atomics after the plain stores of u_next.
    if (a.inj_start) {
        const int s0 = a.inj_start[bid], s1 = a.inj_start[bid + 1];
        if (s1 > s0) {
            __syncthreads();
            // (bf16 store: the source's own share of the imaging term is added in closed form after the adjoint
            // sweep, source_image_kernel, instead of being rounded into the store)
            inject_runs<T, INC, SAVE_Q && !QB>(a, s0, s1, ty * 64 + lane, 64 * TY);
        }
    }
}

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

// x tiles of `tile_x` elements: 64 lanes x one 16-byte vector (256 floats / 128 doubles)
static inline int stream_nxt(const GridDesc &g, int tile_x) { return (int)(round_up(g.nx, tile_x) / tile_x); }

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

bool stream_zpml_supported(const GridDesc &g, const StreamTuning &t, int npml, bool reverse) {
    // (on top of stream_xpml_supported) chunk seams 2 r planes clear of the borders: a chunk that starts or ends inside
    // a border's reach would need its neighbour's psi'.  The adjoint's rings fit the LDS with 4-row tiles only.
    const int r = g.r, zc = t.zchunk > 0 ? t.zchunk : g.nz;
    (void)reverse;
    if (g.nz <= 2 * (npml + 2 * r)) return false;
    if (t.ty != 4) return false;  // 8-row tiles: the rings' registers spill (forward) / their LDS overflows (adjoint)
    for (int zb = zc; zb < g.nz; zb += zc)
        if (zb < npml + 2 * r || zb > g.nz - npml - 2 * r) return false;
    return true;
}

StreamTuning stream_default_tuning(const GridDesc &g, bool is_f32) {
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
    const bool cache_resident = 3.0 * (double)g.npts * (is_f32 ? 4 : 8) <= 200.0 * 1024 * 1024;
    const double WSAT = cache_resident ? 256.0 : 200.0, TY4_PENALTY = cache_resident ? 1.0 : 1.15;
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

template <typename T, int R, int TY, bool DAMP, bool FULL, int PF>
static hipError_t launch_stream_full(const GridDesc &g, const StepArgs<T> &a, int zchunk, int tw, hipStream_t s) {
    const int nxt = stream_nxt(g, tw);
    const int nyt = (g.ny + TY - 1) / TY;
    const int nzc = (g.nz + zchunk - 1) / zchunk;
    const int nblk = nxt * nyt * nzc;
    const int nrb = (a.rec_out && a.nrec > 0) ? (a.nrec + 64 * TY * 4 - 1) / (64 * TY * 4) : 0;
    dim3 block(64, TY), grid(nblk + nrb);
    static const bool no_remap = getenv("FWI_STREAM_NOREMAP") != nullptr;  // tuning hook
    if (no_remap) zchunk = -zchunk;
    if constexpr (std::is_same<T, float>::value) {
        if (a.v) {  // increment form (fp32 only; the adjoint sweep images one pairing per step)
            if constexpr (R == 4 && !DAMP) {
                if (a.xp_mode == 1 && !a.xp_partial) {  // ... with the x border's recursion in the lanes (the term joins q, hence v' and u')
                    if (a.q_out)
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, true, false, 1>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    else
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, true, false, 1>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    return hipGetLastError();
                }
                if (a.xp_mode == 2 && !a.xp_partial) {
                    if (a.q_in)
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, true, false, 2>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    else
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, true, false, 2>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    return hipGetLastError();
                }
                if (a.xp_mode == 1) {  // (border width not a multiple of 4: masked stores)
                    if (a.q_out)
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, true, false, 3>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    else
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, true, false, 3>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    return hipGetLastError();
                }
                if (a.xp_mode == 2) {
                    if (a.q_in)
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, true, false, 4>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    else
                        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, true, false, 4>), grid, block, 0, s,
                                           a, g, zchunk, nxt, nyt, nblk, tw);
                    return hipGetLastError();
                }
            }
            if (a.q_out)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, true>), grid, block, 0, s, a, g,
                                   zchunk, nxt, nyt, nblk, tw);
            else if (a.q_in)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, true>), grid, block, 0, s, a, g,
                                   zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, true>), grid, block, 0, s, a, g,
                                   zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
    }
    if constexpr (std::is_same<T, float>::value) {
        if (a.q_bf16 && (a.q_out || a.q_in)) {  // forward term stored in bf16
            if (a.q_out)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, false, true>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else if (a.q_in2)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 2, FULL, PF, false, true>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, false, true>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
    }
    if constexpr (std::is_same<T, float>::value && R == 4 && !DAMP) {
        if constexpr (TY == 4) {  // (4-row tiles only: with 8 rows the rings' registers spill / the adjoint's LDS overflows)
            if (a.xp_mode == 1 && a.zp_on) {  // x border in the lanes + z border on the march, forward recursion
                if (a.q_out)
                    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, false, false, 1, true>), grid, block,
                                       0, s, a, g, zchunk, nxt, nyt, nblk, tw);
                else
                    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 1, true>), grid, block,
                                       0, s, a, g, zchunk, nxt, nyt, nblk, tw);
                return hipGetLastError();
            }
        }
        if constexpr (TY == 4) {
            if (a.xp_mode == 2 && a.zp_on) {
                if (a.q_in && a.q_in2)
                    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 2, FULL, PF, false, false, 2, true>), grid, block,
                                       0, s, a, g, zchunk, nxt, nyt, nblk, tw);
                else if (a.q_in)
                    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, false, false, 2, true>), grid, block,
                                       0, s, a, g, zchunk, nxt, nyt, nblk, tw);
                else
                    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 2, true>), grid, block,
                                       0, s, a, g, zchunk, nxt, nyt, nblk, tw);
                return hipGetLastError();
            }
        }
        if (a.xp_mode == 1 && !a.xp_partial) {  // x-border CPML in the lanes, forward recursion
            if (a.q_out)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, false, false, 1>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 1>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
        if (a.xp_mode == 2 && !a.xp_partial) {  // ... its transpose (the adjoint sweep)
            if (a.q_in && a.q_in2)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 2, FULL, PF, false, false, 2>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else if (a.q_in)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, false, false, 2>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 2>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
        if (a.xp_mode == 1) {  // ... borders of 10, 14, ... cells: masked stores in the straddling lane
            if (a.q_out)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF, false, false, 3>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 3>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
        if (a.xp_mode == 2) {  // ... its transpose (the adjoint sweep)
            if (a.q_in && a.q_in2)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 2, FULL, PF, false, false, 4>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else if (a.q_in)
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF, false, false, 4>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            else
                hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF, false, false, 4>), grid, block, 0, s,
                                   a, g, zchunk, nxt, nyt, nblk, tw);
            return hipGetLastError();
        }
    }
    if (a.q_out)
        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, true, 0, FULL, PF>), grid, block, 0, s, a, g,
                           zchunk, nxt, nyt, nblk, tw);
    else if (a.q_in && a.q_in2)
        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 2, FULL, PF>), grid, block, 0, s, a, g,
                           zchunk, nxt, nyt, nblk, tw);
    else if (a.q_in)
        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 1, FULL, PF>), grid, block, 0, s, a, g,
                           zchunk, nxt, nyt, nblk, tw);
    else
        hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, false, 0, FULL, PF>), grid, block, 0, s, a, g,
                           zchunk, nxt, nyt, nblk, tw);
    return hipGetLastError();
}

template <typename T, int R, int TY, bool DAMP>
static hipError_t launch_stream_mode(const GridDesc &g, const StepArgs<T> &a, int zchunk, int tw, hipStream_t s) {
    // FULL: every thread of every tile owns grid points, so the stores need no predicate.
    // Prefetch depth: PF = 1 plane ahead measured best (256^3: 39.5 us/step vs 40.0 / 40.2 for
    // PF = 2 / 3; 512^3 equal), i.e. the kernel is throughput- not latency-bound; deeper rings
    // only cost registers.  The template parameter stays for re-tuning.
    if (g.nx % (64 * VecOf<T>::VL) == 0 && g.ny % TY == 0 && tw == 64 * VecOf<T>::VL)
        return launch_stream_full<T, R, TY, DAMP, true, 1>(g, a, zchunk, tw, s);
    return launch_stream_full<T, R, TY, DAMP, false, 1>(g, a, zchunk, tw, s);
}

template <typename T, int R>
static hipError_t launch_stream_r(const GridDesc &g, const StepArgs<T> &a, const StreamTuning &t, hipStream_t s) {
    const int zc = t.zchunk > 0 ? t.zchunk : g.nz;
    if (a.damp) {
        switch (t.ty) {
            case 4: return launch_stream_mode<T, R, 4, true>(g, a, zc, t.tile_x, s);
            default: return launch_stream_mode<T, R, 8, true>(g, a, zc, t.tile_x, s);
        }
    }
    switch (t.ty) {
        case 4: return launch_stream_mode<T, R, 4, false>(g, a, zc, t.tile_x, s);
        default: return launch_stream_mode<T, R, 8, false>(g, a, zc, t.tile_x, s);
    }
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
