// Device helpers shared by the gfx950 step kernels (fwi_kernels.hip, fwi_stream3d.h): 16-byte lane vectors, streaming
// (non-temporal) loads / stores, the bf16 forward-term store, the reciprocal of the damping factor, fused injection.
// Internal; not part of the public boundary (include/fwi.h).
#pragma once
#include "fwi_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace fwi {

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

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

// x tiles of `tile_x` elements: 64 lanes x one 16-byte vector (256 floats / 128 doubles)
static inline int stream_nxt(const GridDesc &g, int tile_x) { return (int)(round_up(g.nx, tile_x) / tile_x); }

}  // namespace fwi
