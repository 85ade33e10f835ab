// The reference's REAL hot loop on gfx950: batched forward_model + similarity metric + likelihood
// for N source samples (full_waveform_inversion.py:713-774).
//
//   forward_model (:253-264)            synth[k,t] = sum_j G[k,j,t] * M[j]
//   compare_synth_to_real_waveforms     (:584-684) optional per-trace max-abs normalisation, then
//     variance_reduction (:512), cross_corr_comparison (:534), pearson_correlation_comparison (:568),
//     cross_corr_comparison_shift_allowed (:548), gaussian_comparison (:578), per trace or flattened
//   likelihood exp(-(1-s)/2) (:774)
//
// fp64 like the reference.  One 256-thread workgroup scores SPB samples at once: each thread streams
// its time samples of G (L2-resident, read once per workgroup, reused for SPB samples in registers),
// forms the synthetic on the fly and accumulates raw moments per (sample, trace); moments are reduced
// with wave64 shuffles + LDS, and one thread per sample evaluates the metric in closed form from the
// moments (the synthetic is never written).  CC-shift uses the exact sums of the 4x linearly
// interpolated sequences (np.interp clamps past the last sample), expressed through lag-1 products.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

namespace fwi {

constexpr int SPB = 4;        // samples per workgroup (G is re-read once per workgroup)
constexpr int MC_THREADS = 256;
// moments per (sample, trace)
enum { M_S1 = 0, M_S2, M_DS, M_MAX, M_SS1, M_DS1, M_D1S, M_FIRST, M_LAST, NMOM };
// data-only moments per trace (host computed): sum d, sum d^2, max|d|, sum d_i d_{i+1}, d_first, d_last
enum { D_1 = 0, D_2, D_MAX, D_DD1, D_FIRST, D_LAST, NDMOM };

enum { MC_VR = 0, MC_CC = 1, MC_PCC = 2, MC_CCSHIFT = 3, MC_GAU = 4 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

struct SeqMom {  // moments of a pair of (possibly upsampled) sequences of common length len
    double sx, sy, sxx, syy, sxy, len;
};

__device__ __forceinline__ double corr_from(const SeqMom &m) {  // Pearson r == the reference's CC
    const double mx = m.sx / m.len, my = m.sy / m.len;
    const double cov = m.sxy / m.len - mx * my;
    const double ex2 = m.sxx / m.len, ey2 = m.syy / m.len;
    const double vx = ex2 - mx * mx, vy = ey2 - my * my;
    // A constant sequence (one-sample traces, flat data): the reference's centred sums give exactly 0 and
    // 0/0 = NaN; from raw moments the variance is round-off of either sign instead.  Anything below
    // 1e-13 of the mean square carries no information in double precision: report it like the reference.
    if (!(vx > 1e-13 * ex2) || !(vy > 1e-13 * ey2)) return NAN;
    const double r = cov / (sqrt(vx) * sqrt(vy));
    return r < 0.0 ? 0.0 : r;
}

// sums of the 4x interpolated sequence built from a plain sequence described by its moments
__device__ __forceinline__ double up_sum(double s1, double first, double last) {
    return 2.5 * (s1 - last) + 1.5 * (s1 - first) + 4.0 * last;
}
__device__ __forceinline__ double up_dot(double p0, double p1, double p1r, double x0y0, double xLyL) {
    return 1.875 * (p0 - xLyL) + 0.875 * (p0 - x0y0) + 0.625 * (p1 + p1r) + 4.0 * xLyL;
}

// NC = number of source components when it is one of the reference's 3 / 6 / 9 (the component loop is
// then unrolled with all G loads of a time sample in flight at once), 0 = any n (rolled loop).
template <bool LAG, int NC>
__global__ __launch_bounds__(MC_THREADS) void mc_score_kernel(
    const double *__restrict__ G, const double *__restrict__ d, const double *__restrict__ Ms,
    const double *__restrict__ dmom, int k, int n, int t, int64_t nsamp, int metric, int normalise,
    int all_at_once, double gau_sigma, double *__restrict__ sim_out, double *__restrict__ like_out) {
    extern __shared__ double smem[];
    double *Mloc = smem;                          // [n][SPB]
    double *mom = Mloc + (size_t)n * SPB;         // [k][SPB][NMOM]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * SPB;

    for (int i = tid; i < n * SPB; i += MC_THREADS) {
        const int j = i / SPB, s = i % SPB;
        Mloc[i] = (s0 + s < nsamp) ? Ms[(int64_t)j * nsamp + s0 + s] : 0.0;  // reference layout MTs[:, i]
    }
    __syncthreads();

    // Each wave owns whole traces (kk = wave, wave + 4, ...): lanes stream consecutive time samples
    // (coalesced), keep per-lane partial moments, and one shuffle reduction per trace yields the
    // moments -- no workgroup barrier inside the trace loop.
    for (int kk = wave; kk < k; kk += MC_THREADS / 64) {
        double a1[SPB], a2[SPB], ads[SPB], amx[SPB], ass1[SPB], ads1[SPB], ad1s[SPB];
#pragma unroll
        for (int s = 0; s < SPB; ++s) a1[s] = a2[s] = ads[s] = amx[s] = ass1[s] = ads1[s] = ad1s[s] = 0.0;
        const double *Gk = G + (int64_t)kk * n * t;
        const double *dk = d + (int64_t)kk * t;
        // software pipeline (NC > 0, no lag terms): the G column and data sample of time index
        // e + 64 are fetched (address clamped, branch-free) while index e is being accumulated
        double gp[NC > 0 ? NC : 1], dp = 0.0;
        if (NC > 0 && !LAG) {
            const int e0 = min(lane, t - 1);
#pragma unroll
            for (int j = 0; j < NC; ++j) gp[j] = Gk[(int64_t)j * t + e0];
            dp = dk[e0];
        }
        for (int e = lane; e < t; e += 64) {
            double sv[SPB], sn[SPB];
#pragma unroll
            for (int s = 0; s < SPB; ++s) sv[s] = sn[s] = 0.0;
            const bool has_next = LAG && (e + 1 < t);
            double dv_pf = 0.0;
            if (NC > 0 && !LAG) {
                double g[NC > 0 ? NC : 1];
#pragma unroll
                for (int j = 0; j < NC; ++j) g[j] = gp[j];
                dv_pf = dp;
                const int en = min(e + 64, t - 1);
#pragma unroll
                for (int j = 0; j < NC; ++j) gp[j] = Gk[(int64_t)j * t + en];
                dp = dk[en];
#pragma unroll
                for (int j = 0; j < NC; ++j) {  // same j order as the reference's accumulation
#pragma unroll
                    for (int s = 0; s < SPB; ++s) sv[s] += g[j] * Mloc[j * SPB + s];
                }
            } else if (NC > 0) {
                double g[NC > 0 ? NC : 1], gn[NC > 0 ? NC : 1];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    g[j] = Gk[(int64_t)j * t + e];
                    gn[j] = has_next ? Gk[(int64_t)j * t + e + 1] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < NC; ++j) {
#pragma unroll
                    for (int s = 0; s < SPB; ++s) {
                        sv[s] += g[j] * Mloc[j * SPB + s];
                        sn[s] += gn[j] * Mloc[j * SPB + s];
                    }
                }
            } else {
                for (int j = 0; j < n; ++j) {
                    const double g = Gk[(int64_t)j * t + e];
                    const double gn = has_next ? Gk[(int64_t)j * t + e + 1] : 0.0;
#pragma unroll
                    for (int s = 0; s < SPB; ++s) {
                        sv[s] += g * Mloc[j * SPB + s];
                        if (LAG) sn[s] += gn * Mloc[j * SPB + s];
                    }
                }
            }
            const double dv = (NC > 0 && !LAG) ? dv_pf : dk[e];
            const double dn = has_next ? dk[e + 1] : 0.0;
#pragma unroll
            for (int s = 0; s < SPB; ++s) {
                a1[s] += sv[s];
                a2[s] += sv[s] * sv[s];
                ads[s] += dv * sv[s];
                amx[s] = fmax(amx[s], fabs(sv[s]));
                if (LAG && has_next) {
                    ass1[s] += sv[s] * sn[s];
                    ads1[s] += dv * sn[s];
                    ad1s[s] += dn * sv[s];
                }
                if (e == 0) mom[((size_t)kk * SPB + s) * NMOM + M_FIRST] = sv[s];
                if (e == t - 1) mom[((size_t)kk * SPB + s) * NMOM + M_LAST] = sv[s];
            }
        }
#pragma unroll
        for (int s = 0; s < SPB; ++s) {
            const double r1 = wave_sum(a1[s]), r2 = wave_sum(a2[s]), r3 = wave_sum(ads[s]), r4 = wave_max(amx[s]);
            double r5 = 0.0, r6 = 0.0, r7 = 0.0;
            if (LAG) {
                r5 = wave_sum(ass1[s]);
                r6 = wave_sum(ads1[s]);
                r7 = wave_sum(ad1s[s]);
            }
            if (lane == 0) {
                double *p = mom + ((size_t)kk * SPB + s) * NMOM;
                p[M_S1] = r1; p[M_S2] = r2; p[M_DS] = r3; p[M_MAX] = r4; p[M_SS1] = r5; p[M_DS1] = r6; p[M_D1S] = r7;
            }
        }
    }
    __syncthreads();

    // one thread per sample turns the moments into the similarity value
    if (tid < SPB && s0 + tid < nsamp) {
        const int s = tid;
        const double L = (double)t;
        double acc = 0.0;                    // per-trace mode: sum of per-trace similarities
        double num = 0.0, den = 0.0;         // all-at-once VR / gau sums
        SeqMom tot = {0, 0, 0, 0, 0, 0};     // all-at-once correlation moments
        double prev_dl = 0.0, prev_sl = 0.0; // last normalised values of the previous trace (flattened lag)
        for (int kk = 0; kk < k; ++kk) {
            const double *m = mom + ((size_t)kk * SPB + s) * NMOM;
            const double *dm = dmom + (size_t)kk * NDMOM;
            const double a = normalise ? 1.0 / dm[D_MAX] : 1.0;
            const double b = normalise ? 1.0 / m[M_MAX] : 1.0;
            const double dd = a * a * dm[D_2], ss = b * b * m[M_S2], ds = a * b * m[M_DS];
            const double d1 = a * dm[D_1], s1 = b * m[M_S1];
            const double res = dd - 2.0 * ds + ss;  // sum (d' - s')^2
            if (metric == MC_VR) {
                if (all_at_once) { num += res; den += dd; }
                else { const double vr = 1.0 - res / dd; acc += vr < 0.0 ? 0.0 : vr; }
            } else if (metric == MC_GAU) {
                num += res;  // per-trace 'gau' returns 0 in the reference (quirk), handled below
            } else if (metric == MC_CC || metric == MC_PCC) {
                if (all_at_once) { tot.sx += d1; tot.sy += s1; tot.sxx += dd; tot.syy += ss; tot.sxy += ds; tot.len += L; }
                else { SeqMom q = {d1, s1, dd, ss, ds, L}; acc += corr_from(q); }
            } else {  // MC_CCSHIFT: moments of the 4x interpolated sequences
                const double df = a * dm[D_FIRST], dl = a * dm[D_LAST], sf = b * m[M_FIRST], sl = b * m[M_LAST];
                const double dd1 = a * a * dm[D_DD1], ss1 = b * b * m[M_SS1];
                const double ds1 = a * b * m[M_DS1], d1s = a * b * m[M_D1S];
                if (!all_at_once) {
                    SeqMom q;
                    q.len = 4.0 * L;
                    q.sx = up_sum(d1, df, dl);
                    q.sy = up_sum(s1, sf, sl);
                    q.sxx = up_dot(dd, dd1, dd1, df * df, dl * dl);
                    q.syy = up_dot(ss, ss1, ss1, sf * sf, sl * sl);
                    q.sxy = up_dot(ds, ds1, d1s, df * sf, dl * sl);
                    acc += corr_from(q);
                } else {
                    // flattened sequence: plain sums now, lag-1 products get the trace-boundary terms;
                    // the first/last corrections are applied once, after the loop
                    tot.sx += d1; tot.sy += s1; tot.sxx += dd; tot.syy += ss; tot.sxy += ds; tot.len += L;
                    num += dd1 + (kk ? prev_dl * df : 0.0);        // sum x_i x_{i+1}   (data)
                    den += ss1 + (kk ? prev_sl * sf : 0.0);        // sum y_i y_{i+1}   (synth)
                    acc += ds1 + d1s + (kk ? prev_dl * sf + prev_sl * df : 0.0);  // x_i y_{i+1} + x_{i+1} y_i
                    prev_dl = dl; prev_sl = sl;
                }
            }
        }
        double sim;
        if (metric == MC_VR) {
            if (all_at_once) { sim = 1.0 - num / den; sim = sim < 0.0 ? 0.0 : sim; }
            else sim = acc / k;
        } else if (metric == MC_GAU) {
            sim = all_at_once ? exp(-num / (2.0 * gau_sigma * gau_sigma)) : 0.0;
        } else if (metric == MC_CC || metric == MC_PCC) {
            sim = all_at_once ? corr_from(tot) : acc / k;
        } else if (!all_at_once) {
            sim = acc / k;
        } else {
            const double *m0 = mom + ((size_t)0 * SPB + s) * NMOM, *mL = mom + ((size_t)(k - 1) * SPB + s) * NMOM;
            const double a0 = normalise ? 1.0 / dmom[D_MAX] : 1.0, b0 = normalise ? 1.0 / m0[M_MAX] : 1.0;
            const double aL = normalise ? 1.0 / dmom[(size_t)(k - 1) * NDMOM + D_MAX] : 1.0;
            const double bL = normalise ? 1.0 / mL[M_MAX] : 1.0;
            const double df = a0 * dmom[D_FIRST], sf = b0 * m0[M_FIRST];
            const double dl = aL * dmom[(size_t)(k - 1) * NDMOM + D_LAST], sl = bL * mL[M_LAST];
            SeqMom q;
            q.len = 4.0 * tot.len;
            q.sx = up_sum(tot.sx, df, dl);
            q.sy = up_sum(tot.sy, sf, sl);
            q.sxx = up_dot(tot.sxx, num, num, df * df, dl * dl);
            q.syy = up_dot(tot.syy, den, den, sf * sf, sl * sl);
            q.sxy = 1.875 * (tot.sxy - dl * sl) + 0.875 * (tot.sxy - df * sf) + 0.625 * acc + 4.0 * dl * sl;
            sim = corr_from(q);
        }
        sim_out[s0 + s] = sim;
        if (like_out) like_out[s0 + s] = exp(-(1.0 - sim) / 2.0);  // :774
    }
}

// ---------------------------------------------------------------------------------------------
// Lane-per-sample kernel (n = 3 / 6 / 9): every lane scores ONE sample, marching through all
// (trace, time) pairs itself.  The Green's function column of a time sample is the same for all
// lanes, so it comes through the scalar path (G is repacked [k][t][n] on the host: one contiguous
// scalar load per time sample) and enters v_fma_f64 as an SGPR operand; the per-trace moments
// live in the lane's registers and are folded into the metric at the end of each trace.  No LDS,
// no shuffles, no barriers: the kernel is pure fp64 FMA work (13 VALU per sample-time-sample).
// ---------------------------------------------------------------------------------------------
// Two consecutive rows of the packed stream Gd[k][t][NC + 1] (row = g_0 .. g_{NC-1}, d) held in
// SGPRs.  The loads are issued by hand: hipcc sinks compiler-visible scalar loads down to their
// first use, which exposes the scalar-cache miss latency of every group; here the next group's
// s_load is in flight while the current group's FMAs issue, and fence() is the matching
// s_waitcnt (SMEM returns out of order, so lgkmcnt(0) is the only usable count).  `pin` is a VGPR
// every FMA chain starts from (M[0]): routing it through the asm keeps the scheduler from hoisting
// the previous group's chains below the load or this group's above it.
typedef double sd8 __attribute__((ext_vector_type(8)));
typedef double sd4 __attribute__((ext_vector_type(4)));
typedef double sd2 __attribute__((ext_vector_type(2)));
constexpr int MC_LANES_UNROLL = 2;  // rows per group; the stream carries >= 3 groups of zero padding
// Generic stage: 2 (NC + 1) doubles cut into pieces of 8 / 8 / 4 / 2 (NC = 3, 6, 9 -- the counts the
// reference produces -- have hand-written single-asm forms below).
template <int NC>
struct McStage {
    static constexpr int D = 2 * (NC + 1), N8 = D / 8, H4 = (D % 8) / 4, H2 = (D % 4) / 2;
    sd8 a, b;
    sd4 c;
    sd2 e;
    __device__ __forceinline__ void load(const double *p, double &pin) {
        if constexpr (N8 >= 1) asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=&s"(a), "+v"(pin) : "s"(p) : "memory");
        if constexpr (N8 >= 2) asm volatile("s_load_dwordx16 %0, %2, 0x40" : "=&s"(b), "+v"(pin) : "s"(p) : "memory");
        if constexpr (H4 == 1)
            asm volatile("s_load_dwordx8 %0, %2, %3" : "=&s"(c), "+v"(pin) : "s"(p), "n"(64 * N8) : "memory");
        if constexpr (H2 == 1)
            asm volatile("s_load_dwordx4 %0, %2, %3" : "=&s"(e), "+v"(pin) : "s"(p), "n"(64 * N8 + 32 * H4) : "memory");
    }
    __device__ __forceinline__ void fence() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (N8 >= 1) asm volatile("" : "+s"(a));  // every later use of a piece depends on these,
        if constexpr (N8 >= 2) asm volatile("" : "+s"(b));  // and volatile asms keep their order
        if constexpr (H4 == 1) asm volatile("" : "+s"(c));
        if constexpr (H2 == 1) asm volatile("" : "+s"(e));
    }
    __device__ __forceinline__ double at(int i) const {
        if (i < 8 * N8) return i < 8 ? a[i & 7] : b[i & 7];
        i -= 8 * N8;
        if (H4 == 1 && i < 4) return c[i & 3];
        return e[(i - 4 * H4) & 1];
    }
};
template <>
struct McStage<9> {  // 20 doubles = 8 + 8 + 4
    sd8 a, b;
    sd4 c;
    __device__ __forceinline__ void load(const double *p, double &pin) {
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx8 %2, %4, 0x80"
                     : "=&s"(a), "=&s"(b), "=&s"(c), "+v"(pin) : "s"(p) : "memory");
    }
    __device__ __forceinline__ void fence() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c)); }
    __device__ __forceinline__ double at(int i) const { return i < 8 ? a[i] : i < 16 ? b[i - 8] : c[i - 16]; }
};
template <>
struct McStage<6> {  // 14 doubles = 8 + 4 + 2
    sd8 a;
    sd4 b;
    sd2 c;
    __device__ __forceinline__ void load(const double *p, double &pin) {
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x40\n\ts_load_dwordx4 %2, %4, 0x60"
                     : "=&s"(a), "=&s"(b), "=&s"(c), "+v"(pin) : "s"(p) : "memory");
    }
    __device__ __forceinline__ void fence() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c)); }
    __device__ __forceinline__ double at(int i) const { return i < 8 ? a[i] : i < 12 ? b[i - 8] : c[i - 12]; }
};
template <>
struct McStage<3> {  // 8 doubles
    sd8 a;
    __device__ __forceinline__ void load(const double *p, double &pin) {
        asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=&s"(a), "+v"(pin) : "s"(p) : "memory");
    }
    __device__ __forceinline__ void fence() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a)); }
    __device__ __forceinline__ double at(int i) const { return a[i]; }
};

template <bool LAG, int NC>
__global__ __launch_bounds__(256) void mc_score_lanes(const double *__restrict__ Gd, const double *__restrict__ Ms,
                                                      const double *__restrict__ dmom, int k, int t, int64_t nsamp,
                                                      int metric, int normalise, int all_at_once,
                                                      double gau_sigma, double *__restrict__ sim_out,
                                                      double *__restrict__ like_out) {
    constexpr int U = MC_LANES_UNROLL, ROW = NC + 1;
    const int64_t smp = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t sc = smp < nsamp ? smp : nsamp - 1;
    double M[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) M[j] = Ms[(int64_t)j * nsamp + sc];  // reference layout MTs[:, i]

    const double L = (double)t;
    double acc = 0.0, num = 0.0, den = 0.0;
    SeqMom tot = {0, 0, 0, 0, 0, 0};
    double prev_dl = 0.0, prev_sl = 0.0;
    double first_sf = 0.0, first_b = 1.0, last_sl = 0.0;  // normalised first / last synthetic samples
    for (int kk = 0; kk < k; ++kk) {
        const double *g = Gd + (int64_t)kk * t * ROW;
        double a1 = 0.0, a2 = 0.0, ads = 0.0, amx = 0.0, ass1 = 0.0, ads1 = 0.0, ad1s = 0.0;
        double sprev = 0.0, dprev = 0.0, sv = 0.0;
        // one time sample: synthetic value (same j order as the reference) and its moments
        auto sample = [&](auto gv, double dv, bool first) __attribute__((always_inline)) {
            sv = 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) sv += gv(j) * M[j];
            a1 += sv;
            a2 += sv * sv;
            ads += dv * sv;
            amx = fmax(amx, fabs(sv));
            if (LAG) {
                if (!first) {
                    ass1 += sprev * sv;
                    ads1 += dprev * sv;
                    ad1s += dv * sprev;
                }
                sprev = sv;
                dprev = dv;
            }
        };
        auto group = [&](const McStage<NC> &st) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; ++u) sample([&](int j) { return st.at(u * ROW + j); }, st.at(u * ROW + NC), false);
        };
        auto plain = [&](int e, bool first) __attribute__((always_inline)) {
            const double *r = g + (int64_t)e * ROW;
            sample([&](int j) { return r[j]; }, r[NC], first);
        };
        // sample 0 apart (it has no lag partner and is the "first" boundary value) ...
        plain(0, true);
        const double sfirst = sv;
        // ... then groups of U rows through two SGPR stages in ping-pong
        int e = 1;
        const int ngroups = (t - 1) / U;
        McStage<NC> A, B;
        A.load(g + (int64_t)e * ROW, M[0]);
        A.fence();
        int gi = 0;
        for (; gi + 2 <= ngroups; gi += 2) {
            B.load(g + (int64_t)(e + U) * ROW, M[0]);
            group(A);
            B.fence();
            A.load(g + (int64_t)(e + 2 * U) * ROW, M[0]);  // may run into the next trace / the padding
            group(B);
            A.fence();
            e += 2 * U;
        }
        if (gi < ngroups) {
            group(A);
            e += U;
        }
        for (; e < t; ++e) plain(e, false);
        const double slast = sv;
        // ---- fold this trace into the metric (same closed forms as the moment kernel above) ----
        const double *dm = dmom + (size_t)kk * NDMOM;
        const double a = normalise ? 1.0 / dm[D_MAX] : 1.0;
        const double b = normalise ? 1.0 / amx : 1.0;
        const double dd = a * a * dm[D_2], ss = b * b * a2, ds = a * b * ads;
        const double d1 = a * dm[D_1], s1 = b * a1;
        const double res = dd - 2.0 * ds + ss;
        if (metric == MC_VR) {
            if (all_at_once) { num += res; den += dd; }
            else { const double vr = 1.0 - res / dd; acc += vr < 0.0 ? 0.0 : vr; }
        } else if (metric == MC_GAU) {
            num += res;
        } else if (metric == MC_CC || metric == MC_PCC) {
            if (all_at_once) { tot.sx += d1; tot.sy += s1; tot.sxx += dd; tot.syy += ss; tot.sxy += ds; tot.len += L; }
            else { SeqMom q = {d1, s1, dd, ss, ds, L}; acc += corr_from(q); }
        } else {
            const double df = a * dm[D_FIRST], dl = a * dm[D_LAST], sf = b * sfirst, sl = b * slast;
            const double dd1 = a * a * dm[D_DD1], ss1 = b * b * ass1;
            const double ds1 = a * b * ads1, d1s = a * b * ad1s;
            if (!all_at_once) {
                SeqMom q;
                q.len = 4.0 * L;
                q.sx = up_sum(d1, df, dl);
                q.sy = up_sum(s1, sf, sl);
                q.sxx = up_dot(dd, dd1, dd1, df * df, dl * dl);
                q.syy = up_dot(ss, ss1, ss1, sf * sf, sl * sl);
                q.sxy = up_dot(ds, ds1, d1s, df * sf, dl * sl);
                acc += corr_from(q);
            } else {
                tot.sx += d1; tot.sy += s1; tot.sxx += dd; tot.syy += ss; tot.sxy += ds; tot.len += L;
                num += dd1 + (kk ? prev_dl * df : 0.0);
                den += ss1 + (kk ? prev_sl * sf : 0.0);
                acc += ds1 + d1s + (kk ? prev_dl * sf + prev_sl * df : 0.0);
                prev_dl = dl; prev_sl = sl;
                if (kk == 0) { first_sf = sf; first_b = b; }
                last_sl = sl;
            }
        }
    }
    (void)first_b;
    double sim;
    if (metric == MC_VR) {
        if (all_at_once) { sim = 1.0 - num / den; sim = sim < 0.0 ? 0.0 : sim; }
        else sim = acc / k;
    } else if (metric == MC_GAU) {
        sim = all_at_once ? exp(-num / (2.0 * gau_sigma * gau_sigma)) : 0.0;
    } else if (metric == MC_CC || metric == MC_PCC) {
        sim = all_at_once ? corr_from(tot) : acc / k;
    } else if (!all_at_once) {
        sim = acc / k;
    } else {
        const double a0 = normalise ? 1.0 / dmom[D_MAX] : 1.0;
        const double aL = normalise ? 1.0 / dmom[(size_t)(k - 1) * NDMOM + D_MAX] : 1.0;
        const double df = a0 * dmom[D_FIRST], sf = first_sf;
        const double dl = aL * dmom[(size_t)(k - 1) * NDMOM + D_LAST], sl = last_sl;
        SeqMom q;
        q.len = 4.0 * tot.len;
        q.sx = up_sum(tot.sx, df, dl);
        q.sy = up_sum(tot.sy, sf, sl);
        q.sxx = up_dot(tot.sxx, num, num, df * df, dl * dl);
        q.syy = up_dot(tot.syy, den, den, sf * sf, sl * sl);
        q.sxy = 1.875 * (tot.sxy - dl * sl) + 0.875 * (tot.sxy - df * sf) + 0.625 * acc + 4.0 * dl * sl;
        sim = corr_from(q);
    }
    if (smp < nsamp) {
        sim_out[smp] = sim;
        if (like_out) like_out[smp] = exp(-(1.0 - sim) / 2.0);  // :774
    }
}

// posterior normalisation (:847-848): p_data = sum_i p_model L_i, post_i = L_i p_model / p_data.
// The sum is taken in a fixed order (per-block partials, then one block over the partials), so the
// posterior is bitwise reproducible from run to run.
constexpr int MC_SUM_BLOCKS = 1024;

__device__ __forceinline__ double mc_block_sum(double s, double *part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void mc_sum_like_kernel(const double *__restrict__ like, int64_t n, double p_model,
                                                          double *__restrict__ partial) {
    __shared__ double part[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += p_model * like[i];
    s = mc_block_sum(s, part);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void mc_sum_partials_kernel(double *__restrict__ partial, int nblocks) {
    __shared__ double part[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    s = mc_block_sum(s, part);
    if (threadIdx.x == 0) partial[MC_SUM_BLOCKS] = s;
}

__global__ __launch_bounds__(256) void mc_posterior_kernel(const double *__restrict__ like, int64_t n, double p_model,
                                                           const double *__restrict__ partial,
                                                           double *__restrict__ post) {
    const double p_data = partial[MC_SUM_BLOCKS];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        post[i] = like[i] * p_model / p_data;
}

int mc_posterior_scratch_doubles() { return MC_SUM_BLOCKS + 1; }

// post (device, n) from like (device, n); scratch = mc_posterior_scratch_doubles() device doubles
hipError_t launch_mc_posterior(const double *like, int64_t n, double *scratch, double *post, hipStream_t s) {
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, MC_SUM_BLOCKS);
    const double p_model = 1.0 / (double)n;
    hipLaunchKernelGGL(mc_sum_like_kernel, dim3(blocks), dim3(256), 0, s, like, n, p_model, scratch);
    hipLaunchKernelGGL(mc_sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, blocks);
    hipLaunchKernelGGL(mc_posterior_kernel, dim3(blocks), dim3(256), 0, s, like, n, p_model, scratch, post);
    return hipGetLastError();
}

// forward_model for a batch: synth[i, k, t] (full_waveform_inversion.py:253-264), same j order
__global__ void mc_forward_kernel(const double *__restrict__ G, const double *__restrict__ Ms, int k, int n,
                                  int t, int64_t nsamp, double *__restrict__ synth) {
    const int64_t kt = (int64_t)k * t;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nsamp * kt;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / kt, r = i - s * kt;
        const int kk = (int)(r / t), e = (int)(r % t);
        double v = 0.0;
        for (int j = 0; j < n; ++j) v += G[((int64_t)kk * n + j) * t + e] * Ms[(int64_t)j * nsamp + s];
        synth[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Device-side source samplers: the reference's seven generate_random_* functions
// (full_waveform_inversion.py:282-510), one thread per sample.  The deterministic maps are the
// reference's (same steps as full_waveform_inversion_amd/samplers.py, which is pinned to reference
// golden vectors); the deviates come from a counter-based generator so that sample i is a pure
// function of (seed, i) whatever the launch shape or the number of GPUs:
//   Philox4x32-10, counter (i_lo, i_hi, block, 0), key (seed_lo, seed_hi); one block = 4 x u32 =
//   two doubles in [0, 1) (53 bits each, (a >> 5) * 2^26 + (b >> 6), as numpy forms them);
//   blocks 0..2  -> normal pairs (z0, z1), (z2, z3), (z4, z5) by Box-Muller
//                   r = sqrt(-2 ln(1 - u_a)), z = r cos / sin(2 pi u_b);
//   blocks 8, 9  -> uniforms (U0, U1), (U2, U3): u_theta = 2 U0 - 1, r_phi = U1,
//                   r_quadrant = U2, frac = U3 (the coupled non-crack types take frac = U0).
// oracle/mc_oracle.py restates the generator (known-answer vectors of Philox4x32-10 included).
// ---------------------------------------------------------------------------------------------
enum { MC_T_FULL = 0, MC_T_DC = 1, MC_T_SF = 2, MC_T_DC_SF_C = 3, MC_T_DC_SF_U = 4, MC_T_DC_CRACK = 5,
       MC_T_SF_CRACK = 6 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct McRng {
    uint32_t i_lo, i_hi, k0, k1;
    __device__ void uniforms(uint32_t block, double &ua, double &ub) const {
        uint32_t r[4];
        philox4x32_10(i_lo, i_hi, block, 0u, k0, k1, r);
        ua = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
        ub = ((double)(r[2] >> 5) * 67108864.0 + (double)(r[3] >> 6)) * (1.0 / 9007199254740992.0);
    }
    __device__ void normals(uint32_t block, double &z0, double &z1) const {
        double ua, ub;
        uniforms(block, ua, ub);
        const double r = sqrt(-2.0 * log(1.0 - ua)), ang = 6.283185307179586476925 * ub;
        double sn, cs;
        sincos(ang, &sn, &cs);
        z0 = r * cs;
        z1 = r * sn;
    }
};

// the reference's two-step normalisation (:288-290): a / (sum a^2)^-0.5, then / norm of that
template <int N>
__device__ __forceinline__ void mc_unit(double (&a)[N]) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) s += a[j] * a[j];
    const double f = pow(s, -0.5);
    double s1 = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        a[j] = a[j] / f;
        s1 += a[j] * a[j];
    }
    const double nrm = sqrt(s1);
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = a[j] / nrm;
}

// rot_mt_by_theta_phi (:226-232): R_phi (R_theta M R_theta^T) R_phi^T, then the six-vector (:206-208)
__device__ void mc_rot_six(const double (&Min)[3][3], double theta, double phi, double (&six)[6]) {
    double st, ct, sp, cp;
    sincos(theta, &st, &ct);
    sincos(phi, &sp, &cp);
    const double Rt[3][3] = {{ct, 0.0, st}, {0.0, 1.0, 0.0}, {-st, 0.0, ct}};
    const double Rp[3][3] = {{cp, -sp, 0.0}, {sp, cp, 0.0}, {0.0, 0.0, 1.0}};
    double A[3][3], B[3][3];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const double(&R)[3][3] = pass ? Rp : Rt;
        const double(&S)[3][3] = pass ? B : Min;
#pragma unroll
        for (int i = 0; i < 3; ++i)  // A = S R^T
#pragma unroll
            for (int j = 0; j < 3; ++j) A[i][j] = S[i][0] * R[j][0] + S[i][1] * R[j][1] + S[i][2] * R[j][2];
        double C[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)  // C = R A
#pragma unroll
            for (int j = 0; j < 3; ++j) C[i][j] = R[i][0] * A[0][j] + R[i][1] * A[1][j] + R[i][2] * A[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) B[i][j] = C[i][j];
    }
    const double r2 = 1.4142135623730951;
    six[0] = B[0][0]; six[1] = B[1][1]; six[2] = B[2][2];
    six[3] = r2 * B[0][1]; six[4] = r2 * B[0][2]; six[5] = r2 * B[1][2];
}

__device__ __forceinline__ void mc_norm6(double (&six)[6]) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += six[j] * six[j];
    const double nrm = sqrt(s);
#pragma unroll
    for (int j = 0; j < 6; ++j) six[j] = six[j] / nrm;
}

// crack tensor on the lune boundary (:393-424): diagonal (c - sqrt2 s, same, c + 2 sqrt2 s) * scale
__device__ void mc_crack(double u_theta, double r_phi, double r_quadrant, double (&M)[3][3]) {
    const double pi = 3.141592653589793;
    const double theta_l = u_theta * pi / 2.0, phi_l = r_phi <= 0.5 ? 0.0 : pi / 3;
    double ang = atan(sin(phi_l) / sin(theta_l));
    if (r_quadrant > 0.25 && r_quadrant <= 0.5) ang += pi;
    if (r_quadrant > 0.5 && r_quadrant <= 0.75) ang += pi / 2;
    if (r_quadrant > 0.75 && r_quadrant <= 1.0) ang += 3 * pi / 2;
    double s, c;
    sincos(ang, &s, &c);
    const double scale = pow(4.0 * (s * s) + c * c, -0.5) / sqrt(3.0);
    const double r2 = 1.4142135623730951;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = 0.0;
    M[0][0] = M[1][1] = scale * (c - r2 * s);
    M[2][2] = scale * (c + 2.0 * r2 * s);
}

__global__ __launch_bounds__(256) void mc_sample_kernel(int type, uint64_t seed, int64_t first, int64_t nsamp,
                                                        double amplitude, double *__restrict__ Ms, int64_t ld,
                                                        double *__restrict__ frac_out) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nsamp) return;
    const uint64_t idx = (uint64_t)(first + s);
    const McRng rng = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)seed, (uint32_t)(seed >> 32)};
    double z[6];
    rng.normals(0, z[0], z[1]);
    rng.normals(1, z[2], z[3]);
    double out[9];
    double frac = 0.0;
    int n = 6;
    if (type == MC_T_FULL) {
        rng.normals(2, z[4], z[5]);
        mc_unit<6>(z);
#pragma unroll
        for (int j = 0; j < 6; ++j) out[j] = z[j];
    } else if (type == MC_T_SF) {
        double a[3] = {z[0], z[1], z[2]};
        mc_unit<3>(a);
        n = 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = a[j];
    } else {
        rng.normals(2, z[4], z[5]);
        double U0, U1, U2, U3;
        rng.uniforms(8, U0, U1);
        rng.uniforms(9, U2, U3);
        const bool crack = type == MC_T_DC_CRACK || type == MC_T_SF_CRACK;
        frac = crack ? U3 : U0;
        // orientation vector: z[0..2], except the single-force-crack type, whose first triple is the force
        double a[3];
        const int o = type == MC_T_SF_CRACK ? 3 : 0;
        a[0] = z[o]; a[1] = z[o + 1]; a[2] = z[o + 2];
        mc_unit<3>(a);
        double theta, phi;
        if (type == MC_T_SF_CRACK) {  // :497-498
            theta = acos(a[2]);
            phi = acos(a[0] / sin(theta));
        } else {  // :308-309
            theta = atan2(sqrt(a[0] * a[0] + a[1] * a[1]), a[2]);
            phi = atan2(a[1], a[0]);
        }
        double Mt[3][3] = {{0.0, 0.0, 1.0}, {0.0, 0.0, 0.0}, {1.0, 0.0, 0.0}};  // the double couple (:299)
        if (crack) {
            double Cr[3][3];
            mc_crack(2.0 * U0 - 1.0, U1, U2, Cr);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Mt[i][j] = type == MC_T_DC_CRACK ? frac * Mt[i][j] + (1.0 - frac) * Cr[i][j] : Cr[i][j];
        }
        double six[6];
        mc_rot_six(Mt, theta, phi, six);
        if (type != MC_T_SF_CRACK) mc_norm6(six);  // the single-force-crack tensor is not re-normalised (:500-503)
        if (type == MC_T_DC || type == MC_T_DC_CRACK) {
#pragma unroll
            for (int j = 0; j < 6; ++j) out[j] = six[j];
        } else {
            n = 9;
            double f[3];
            if (type == MC_T_DC_SF_C) {  // force along the slip vector: rotate (1, 0, 0), NED -> END (:356-358)
                double st, ct, sp, cp;
                sincos(theta, &st, &ct);
                sincos(phi, &sp, &cp);
                const double v0 = ct, v2 = -st;              // R_theta (1, 0, 0)
                const double w0 = cp * v0, w1 = sp * v0;     // R_phi v
                f[0] = w1; f[1] = w0; f[2] = v2;
            } else {
                const int fo = type == MC_T_SF_CRACK ? 0 : 3;
                f[0] = z[fo]; f[1] = z[fo + 1]; f[2] = z[fo + 2];
                mc_unit<3>(f);
            }
            // amplitude split: DC part * frac, force * (1 - frac) (:362-363, :378-379); the
            // single-force-crack type gives frac to the FORCE (:507-509)
            const double wt = type == MC_T_SF_CRACK ? 1.0 - frac : frac, wf = 1.0 - wt;
#pragma unroll
            for (int j = 0; j < 6; ++j) out[j] = six[j] * wt;
#pragma unroll
            for (int j = 0; j < 3; ++j) out[6 + j] = f[j] * wf;
        }
    }
    for (int j = 0; j < n; ++j) Ms[(int64_t)j * ld + s] = out[j] * amplitude;
    if (frac_out) frac_out[s] = frac;
}

int mc_sampler_components(int type) {
    switch (type) {
        case MC_T_FULL: case MC_T_DC: case MC_T_DC_CRACK: return 6;
        case MC_T_SF: return 3;
        case MC_T_DC_SF_C: case MC_T_DC_SF_U: case MC_T_SF_CRACK: return 9;
        default: return 0;
    }
}

hipError_t launch_mc_sample(int type, uint64_t seed, int64_t first, int64_t nsamp, double amplitude, double *Ms,
                            int64_t ld, double *frac, hipStream_t s) {
    const unsigned blocks = (unsigned)((nsamp + 255) / 256);
    hipLaunchKernelGGL(mc_sample_kernel, dim3(blocks), dim3(256), 0, s, type, seed, first, nsamp, amplitude, Ms, ld,
                       frac);
    return hipGetLastError();
}

// Gt = G and d packed [k][t][n + 1] (row = g_0..g_{n-1}, d) followed by >= 1 KiB of zeros; given for
// n <= 9, which take the lane-per-sample kernel
hipError_t launch_mc_score(const double *G, const double *Gt, const double *d, const double *Ms, const double *dmom,
                           int k, int n, int t, int64_t nsamp, int metric, int normalise, int all_at_once,
                           double gau_sigma, double *sim, double *like, hipStream_t s) {
    if (Gt && n >= 1 && n <= 9) {
        const unsigned blocks = (unsigned)((nsamp + 255) / 256);
        const bool lg = metric == MC_CCSHIFT;
#define MC_LANES(LAG, NC)                                                                                \
    hipLaunchKernelGGL((mc_score_lanes<LAG, NC>), dim3(blocks), dim3(256), 0, s, Gt, Ms, dmom, k, t, nsamp, \
                       metric, normalise, all_at_once, gau_sigma, sim, like)
#define MC_LANES_N(NC) case NC: if (lg) MC_LANES(true, NC); else MC_LANES(false, NC); break
        switch (n) {
            MC_LANES_N(1); MC_LANES_N(2); MC_LANES_N(3); MC_LANES_N(4); MC_LANES_N(5);
            MC_LANES_N(6); MC_LANES_N(7); MC_LANES_N(8); MC_LANES_N(9);
        }
#undef MC_LANES_N
#undef MC_LANES
        return hipGetLastError();
    }
    const size_t shm = ((size_t)n * SPB + (size_t)k * SPB * NMOM) * sizeof(double);
    const unsigned grid = (unsigned)((nsamp + SPB - 1) / SPB);
#define MC_LAUNCH(LAG, NC)                                                                                  \
    hipLaunchKernelGGL((mc_score_kernel<LAG, NC>), dim3(grid), dim3(MC_THREADS), shm, s, G, d, Ms, dmom, k, n, \
                       t, nsamp, metric, normalise, all_at_once, gau_sigma, sim, like)
    const bool lag = metric == MC_CCSHIFT;
    switch (n) {
        case 3: if (lag) MC_LAUNCH(true, 3); else MC_LAUNCH(false, 3); break;
        case 6: if (lag) MC_LAUNCH(true, 6); else MC_LAUNCH(false, 6); break;
        case 9: if (lag) MC_LAUNCH(true, 9); else MC_LAUNCH(false, 9); break;
        default: if (lag) MC_LAUNCH(true, 0); else MC_LAUNCH(false, 0); break;
    }
#undef MC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_mc_forward(const double *G, const double *Ms, int k, int n, int t, int64_t nsamp, double *synth,
                             hipStream_t s) {
    const int64_t tot = nsamp * k * t;
    const unsigned grid = (unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    hipLaunchKernelGGL(mc_forward_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, G, Ms, k, n, t, nsamp, synth);
    return hipGetLastError();
}

size_t mc_score_lds_bytes(int k, int n) {
    return ((size_t)n * SPB + (size_t)k * SPB * NMOM) * sizeof(double);
}

}  // namespace fwi
