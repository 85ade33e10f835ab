// The 3-D STREAM kernel (step3d_stream) and its launch dispatch.  Included by the translation units that instantiate it
// (fwi_stream3d_f32_o8.hip, fwi_stream3d_f32_lo.hip, fwi_stream3d_f64.hip): one per dtype / order group, so that the
// instantiations compile in parallel and a single group can be rebuilt (and its code object inspected) alone.
#pragma once
#include "fwi_device.h"

namespace fwi {

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
// IMAGE: 0 = off, 1 = g += u_cur * q_in, 2 = additionally g += u_prev * q_in2 (increment form, round 4: u_prev = u_cur - v) (two time levels per
// read-modify-write of g: the adjoint sweep is HBM-bound, this takes it from 28 to 24 B/update).
// XP: the convolutional PML of the x border carried in the lanes (1 = forward recursion, 2 = its transpose): the
// border cells of a row are the first / last npml / 4 lanes of the wave that owns it, psi' of the neighbouring cells
// comes through __shfl_up / __shfl_down (the 2 r-wide dependence of the border recursion never leaves the wave),
// D u and E_x u from the x window the stencil has in registers anyway.  The memory variables are read and written
// once per step by the lanes that own them -- no slab launches for this axis, no second pass over u' and q.
// TP: the z / y border's CPML term is handed over by the line launch that ran before this step (StepArgs::pml_tz /
// pml_ty, compact over the axis' shell) and joins q here -- two more 16-byte streams, fetched one plane ahead like
// u_prev and C, that only shell planes (z: a wave-uniform test of the plane index) and shell rows (y: fixed per
// thread) really read: everything else re-reads one cached line and discards it.
template <typename T, int R, int TY, bool DAMP, bool SAVE_Q, int IMAGE, bool FULL, int PF, bool INC = false,
          bool QB = false, int XPM = 0, bool TP = false>
__global__ __launch_bounds__(64 * TY) void step3d_stream(StepArgs<T> a, GridDesc g, int zchunk,
                                                         int nxt, int nyt, int nblk, int tw) {
    // XPM 1 / 2: the x border in the lanes, forward / adjoint; 3 / 4: the same for a border whose width is not a
    // multiple of the lane vector (npml = 10, 14, ...): the one lane per side that straddles the border's inner edge
    // stores its memory variables cell by cell (the slab row holds the other side's cells right behind)
    constexpr int XP = XPM == 0 ? 0 : ((XPM - 1) & 1) + 1;
    constexpr bool XMASK = XPM > 2;
    // DIET: 8-row tiles are 512 threads = two waves per SIMD = 256 registers per lane, which the x border's recursion
    // (its operands, windows and the handed-over terms on top of the queues) does not fit with rings of PF + 1 slots:
    // round 3 shipped its adjoint variants with 10 - 38 spilled registers, i.e. scratch reloads in the pipelined loop.
    // Each variant now takes as much of the following as it needs to fit (tests/test_code_objects.py: no scratch) and
    // no more -- every item costs a little of the loop's memory-level parallelism (512^3 forward with everything on:
    // 554 -> 607 us/step, measured):
    //   1  the lane's a(x), b(x) live in LDS instead of registers (-8)                       [forward sweeps]
    //   2  + the handed-over terms keep ONE slot: the load of plane z + 1 is issued right after the last use of plane
    //        z's value -- it still has most of an iteration to land (-8)                     [adjoint, no imaging]
    //   3  + the same for u_prev / v, C and the x border's memory variables (-16), and the halo rows are fetched one
    //        plane ahead instead of r (-12)                                                 [adjoint + imaging]
    constexpr int DIET = (TY == 8 && XP != 0) ? (XP == 2 ? (IMAGE ? 3 : 2) : 1) : 0;
    constexpr bool XC_LDS = DIET >= 1, LATE_T = DIET >= 2, LATE = DIET >= 3;
    static_assert(DIET < 2 || PF == 1, "the single-slot rings are written for a prefetch distance of one plane");
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
    __shared__ V xcl[XC_LDS ? 2 : 1][XC_LDS ? TY : 1][XC_LDS ? 64 : 1];  // (DIET: the lane's a(x), b(x), read back per plane)
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
        if constexpr (XC_LDS) {  // (each thread reads its own slots only: no barrier)
            xcl[0][ty][lane] = xca;
            xcl[1][ty][lane] = xcb;
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
    constexpr int NQ = (2 * R + 1 + PF + NR - 1) / NR * NR;
    constexpr int NRP = LATE ? 1 : NR;                                    // slots of the pointwise rings
    constexpr int HPF = (!LATE && NQ % (R + 1) == 0) ? R : PF;            // halo prefetch distance
    constexpr int NRH = HPF + 1;
    V zq[NQ];
#pragma unroll
    for (int k = 0; k < 2 * R + PF; ++k) zq[k] = ldv<T>(a.u_cur + (int64_t)(z0 - R + k) * sz + poff);
    V up[NRP], Cc[NRP], halo[NRH][NH];
    V xps[XP ? NRP : 1], xzt[XP ? NRP : 1];  // psi / zeta (adjoint: pt / zt) of the x border, fetched like up / Cc
    constexpr int NRT = LATE_T ? 1 : NR;
    V tzr[TP ? NRT : 1], tyr[TP ? NRT : 1];  // the handed-over z / y border terms of the planes in flight
    // y shell: fixed per thread.  Rows off the shell (and lanes that own no points) re-read the array's first vectors.
    bool yin = false;
    unsigned tyoff = 0;
    int64_t typl = 0;
    const unsigned tco = act ? coff : 0u;
    if constexpr (TP) {
        const int nsy = pml_shell_rows(g.ny, a.npml, R);
        yin = act && pml_in_shell(y, g.ny, a.npml, R);
        tyoff = yin ? (unsigned)(pml_shell_index(y, g.ny, a.npml, R) * g.cx + x0) : 0u;  // (off the shell: ONE address)
        typl = yin ? (int64_t)nsy * g.cx : 0;
    }
    auto tz_at = [&](int p) {  // plane p of the z term (clamped); planes off the shell: every lane re-reads the array's
                               // first vector (one cache line per wave instead of a plane-sized footprint) and discards it
        const int pc = min(p, g.nz - 1);
        const bool in = pml_in_shell(pc, g.nz, a.npml, R);
        const int sp = in ? pml_shell_index(pc, g.nz, a.npml, R) : 0;
        return ldv<T>(a.pml_tz + (int64_t)sp * cplane + (in ? tco : 0u));
    };
    // (the array holds nz + 1 planes: the prefetch of the plane behind the last one needs no clamp)
    auto ty_at = [&](int p) { return ldv<T>(a.pml_ty + (int64_t)p * typl + tyoff); };
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
        if constexpr (TP) {
            tzr[p] = tz_at(z0 + p);
            tyr[p] = ty_at(z0 + p);
        }
    }
#pragma unroll
    for (int p = 0; p < HPF; ++p) {
        const int64_t o = (int64_t)(z0 + p) * sz;
#pragma unroll
        for (int i = 0; i < NH; ++i) halo[p][i] = ldv<T>(a.u_cur + o + hoff[i]);
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
            zq[(ph + 2 * R + PF) % NQ] = ldv<T>(a.u_cur + (int64_t)(z + R + PF) * sz + poff);
            {
                const int64_t oh = (int64_t)(z + HPF) * sz;
#pragma unroll
                for (int i = 0; i < NH; ++i) halo[(ph + HPF) % NRH][i] = ldv<T>(a.u_cur + oh + hoff[i]);
            }
            edge[nxt] = a.u_cur[on + eoff];
            // the pointwise operands of plane z + PF (LATE: into the one slot, once plane z's values have been used)
            // (plain loads: non-temporal hints on these read-once streams were measured and
            // rejected -- 256^3 39 -> 51 us/step, they defeat Infinity-Cache residency; 512^3 +-2 %)
            const int cp = LATE ? 0 : cur, np = LATE ? 0 : nxt, ct = LATE_T ? 0 : cur, nt_ = LATE_T ? 0 : nxt;
            auto fetch_pointwise = [&]() __attribute__((always_inline)) {
                up[np] = ldv<T>(pw + on + poff);
                Cc[np] = ldv<T>(a.C + on + poff);
                if constexpr (XP != 0) {
                    const int64_t ox = (int64_t)min(z + PF, g.nz - 1) * xplane + xld;
                    xps[np] = ldv<T>(a.xp_psi + ox);
                    xzt[np] = ldv<T>(a.xp_zeta + ox);
                }
            };
            auto fetch_terms = [&]() __attribute__((always_inline)) {
                if constexpr (TP) {
                    tzr[nt_] = tz_at(z + PF);
                    tyr[nt_] = ty_at(z + PF);
                }
            };
            if constexpr (!LATE) fetch_pointwise();
            if constexpr (!LATE_T) fetch_terms();
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
            V lap;
#pragma unroll
            for (int j = 0; j < VL; ++j) lap.v[j] = T(0);
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
                }
            }
            // x-border CPML: this lane's cells of the border recursion, neighbours through the wave
            V xterm, xn0, xn1;
            if constexpr (XP != 0) {
                T W[2 * HALO + VL];  // [left lane's cells][own][right lane's] of a quantity, zero beyond the wave
                // neighbours by DPP (wave_shr:1 / wave_shl:1, GFX9 family: a cross-lane move in the VALU, zero where the
                // wave ends) instead of __shfl_up / __shfl_down, which hipcc lowers to ds_bpermute -- an LDS-crossbar round
                // trip per value, 8 - 24 of them in a dependent chain per plane; at one or two waves per SIMD that chain is
                // what the in-between grid sizes pay for the x border (384^3: step kernel 296 -> see DESIGN.md s.4 CPML)
                auto spread = [&](const V &v) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < VL; ++j) {
                        if constexpr (std::is_same<T, float>::value) {
                            const int b = __builtin_bit_cast(int, v.v[j]);
                            W[j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x138, 0xf, 0xf, true));
                            W[HALO + VL + j] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x130, 0xf, 0xf, true));
                        } else {
                            const T l = __shfl_up(v.v[j], 1, 64), r = __shfl_down(v.v[j], 1, 64);
                            W[j] = lane == 0 ? T(0) : l;
                            W[HALO + VL + j] = lane == 63 ? T(0) : r;
                        }
                        W[HALO + j] = v.v[j];
                    }
                };
                const V m0 = xps[cp], m1 = xzt[cp];
                if constexpr (XC_LDS) {
                    // an index the compiler cannot see through: otherwise it hoists these loop-invariant reads out of the
                    // z loop and the eight registers are back
                    int opq = 0;
                    asm volatile("" : "+v"(opq));
                    xca = xcl[opq][ty][lane];
                    xcb = xcl[opq + 1][ty][lane];
                }
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
            const bool zin = TP && pml_in_shell(z, g.nz, a.npml, R);
#pragma unroll
            for (int j = 0; j < VL; ++j) {
                T br = lap.v[j];  // the bracket of q: L u + the CPML terms of the three axes
                if (XP) br += xterm.v[j];
                if (TP) br += (zin ? tzr[ct].v[j] : T(0)) + (yin ? tyr[ct].v[j] : T(0));
                q.v[j] = Cc[cp].v[j] * br;
                if (INC) {  // v' = A (B v + q), u' = u + v'
                    vn.v[j] = DAMP ? fma(B.v[j], up[cp].v[j], q.v[j]) * A.v[j] : up[cp].v[j] + q.v[j];
                    un.v[j] = X[HALO + j] + vn.v[j];
                } else if (DAMP)
                    un.v[j] = (fma(T(2), X[HALO + j], -B.v[j] * up[cp].v[j]) + q.v[j]) * A.v[j];
                else
                    un.v[j] = (T(2) * X[HALO + j] - up[cp].v[j]) + q.v[j];
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
            if (IMAGE) {
#pragma unroll
                for (int j = 0; j < VL; ++j) {
                    gi.v[j] = fma(X[HALO + j], qi.v[j], gi.v[j]);
                    // (the second pairing takes the field one step further back: u_prev -- in increment form u - v)
                    if (IMAGE == 2) gi.v[j] = fma(INC ? X[HALO + j] - up[cp].v[j] : up[cp].v[j], qi2.v[j], gi.v[j]);
                }
            }
            if constexpr (LATE) fetch_pointwise();  // (every use of plane z's pointwise operands lies above)
            if constexpr (LATE_T) fetch_terms();
            if (act) {
                if (INC) stv<T>(a.v + (int64_t)z * sz + poff, vn);
                stv<T>(a.u_prev + (int64_t)z * sz + poff, un);
                if (SAVE_Q) st_qf<QB>(a.q_out, (int64_t)z * cplane + coff, q);
                if (IMAGE) stv_stream<T>(a.g + (int64_t)z * cplane + coff, gi);
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

// One (dtype, order, tile, border, form) family of the kernel: picks the sweep's variant by its operands -- q_out: a
// forward sweep that stores the forward term; q_in (+ q_in2): an adjoint sweep that images one (two) time level(s)
// per launch.  DIR restricts a family to the variants its sweep direction can need (the in-lane x border is compiled
// per direction): +1 forward (plain / store), -1 adjoint (plain / imaging), 0 both.
struct StreamLaunch {
    dim3 grid, block;
    hipStream_t s;
    int zchunk, nxt, nyt, nblk, tw;
};
template <typename T, int R, int TY, bool DAMP, bool FULL, int PF, bool INC, bool QB, int XPM, bool TP, int DIR>
static hipError_t launch_stream_family(const StreamLaunch &l, const GridDesc &g, const StepArgs<T> &a) {
#define FWI_STREAM_GO(SAVE_Q, IMAGE)                                                                                    \
    hipLaunchKernelGGL((step3d_stream<T, R, TY, DAMP, SAVE_Q, IMAGE, FULL, PF, INC, QB, XPM, TP>), l.grid, l.block, 0, l.s, \
                       a, g, l.zchunk, l.nxt, l.nyt, l.nblk, l.tw)
    if (a.q_out) {
        if constexpr (DIR >= 0) FWI_STREAM_GO(true, 0);
        else return hipErrorInvalidValue;
    } else if (a.q_in && a.q_in2) {
        if constexpr (DIR <= 0) FWI_STREAM_GO(false, 2);
        else return hipErrorInvalidValue;
    } else if (a.q_in) {
        if constexpr (DIR <= 0) FWI_STREAM_GO(false, 1);
        else return hipErrorInvalidValue;
    } else {
        FWI_STREAM_GO(false, 0);
    }
#undef FWI_STREAM_GO
    return hipGetLastError();
}

template <typename T, int R, int TY, bool DAMP, bool FULL, int PF>
static hipError_t launch_stream_full(const GridDesc &g, const StepArgs<T> &a, int zchunk, int tw, hipStream_t s) {
    const int nxt = stream_nxt(g, tw);
    const int nyt = (g.ny + TY - 1) / TY;
    const int nzc = (g.nz + zchunk - 1) / zchunk;
    const int nblk = nxt * nyt * nzc;
    const int nrb = (a.rec_out && a.nrec > 0) ? (a.nrec + 64 * TY * 4 - 1) / (64 * TY * 4) : 0;
    static const bool no_remap = getenv("FWI_STREAM_NOREMAP") != nullptr;  // tuning hook
    if (no_remap) zchunk = -zchunk;
    const StreamLaunch l{dim3(nblk + nrb), dim3(64, TY), s, zchunk, nxt, nyt, nblk, tw};
    constexpr bool F32 = std::is_same<T, float>::value;
    const bool tp = a.pml_tz != nullptr && a.pml_ty != nullptr;  // the z / y border's CPML term comes with the step
    if ((a.pml_tz != nullptr) != (a.pml_ty != nullptr) || (a.xp_mode != 0 && !tp) || (tp && a.q_bf16))
        return hipErrorInvalidValue;  // (the C-ABI never builds these: the in-lane x border exists with the lines only)
    if constexpr (!DAMP) {  // (the convolutional PML switches the sponge off)
        if (tp) {
            if constexpr (F32 && R == 4) {
                // ... with the x border's recursion in the lanes: forward / adjoint, whole-vector / masked border lanes
                if (a.v) {
                    switch (a.xp_mode ? a.xp_mode + (a.xp_partial ? 2 : 0) : 0) {
                        case 1: return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 1, true, 1>(l, g, a);
                        case 2: return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 2, true, -1>(l, g, a);
                        case 3: return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 3, true, 1>(l, g, a);
                        case 4: return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 4, true, -1>(l, g, a);
                        default: break;
                    }
                } else {
                    switch (a.xp_mode ? a.xp_mode + (a.xp_partial ? 2 : 0) : 0) {
                        case 1: return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 1, true, 1>(l, g, a);
                        case 2: return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 2, true, -1>(l, g, a);
                        case 3: return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 3, true, 1>(l, g, a);
                        case 4: return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 4, true, -1>(l, g, a);
                        default: break;
                    }
                }
            } else if (a.xp_mode != 0) {
                return hipErrorInvalidValue;
            }
            if constexpr (F32) {
                if (a.v) return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 0, true, 0>(l, g, a);
            }
            return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 0, true, 0>(l, g, a);
        }
    } else if (tp) {
        return hipErrorInvalidValue;
    }
    if constexpr (F32) {
        if (a.v) return launch_stream_family<T, R, TY, DAMP, FULL, PF, true, false, 0, false, 0>(l, g, a);  // increment form
        if (a.q_bf16 && (a.q_out || a.q_in)) {                                                              // bf16 store
            if constexpr (R == 4) return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, true, 0, false, 0>(l, g, a);
            else return hipErrorInvalidValue;  // (fwi_create admits the bf16 store with the O(8) stencil only)
        }
    }
    return launch_stream_family<T, R, TY, DAMP, FULL, PF, false, false, 0, false, 0>(l, g, a);
}

template <typename T, int R, int TY, bool DAMP>
static hipError_t launch_stream_mode(const GridDesc &g, const StepArgs<T> &a, int zchunk, int tw, hipStream_t s) {
    // FULL: every thread of every tile owns grid points, so the stores need no predicate.
    // Prefetch depth: PF = 1 plane ahead measured best (256^3: 39.5 us/step vs 40.0 / 40.2 for
    // PF = 2 / 3; 512^3 equal), i.e. the kernel is throughput- not latency-bound; deeper rings
    // only cost registers.  The template parameter stays for re-tuning.
    // (O(8) only since round 4: the unpredicated form is a 1-2 % specialisation of the configurations that are
    // benchmarked; for O(2) / O(4) it doubled the instantiations for nothing measurable)
    if constexpr (R == 4) {
        if (g.nx % (64 * VecOf<T>::VL) == 0 && g.ny % TY == 0 && tw == 64 * VecOf<T>::VL)
            return launch_stream_full<T, R, TY, DAMP, true, 1>(g, a, zchunk, tw, s);
    }
    return launch_stream_full<T, R, TY, DAMP, false, 1>(g, a, zchunk, tw, s);
}

template <typename T, int R>
hipError_t launch_stream_r(const GridDesc &g, const StepArgs<T> &a, const StreamTuning &t, hipStream_t s) {
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

}  // namespace fwi
