// C-ABI of the reference's real hot loop: batched sampling and scoring of source
// samples -- fwi_mc_score, fwi_mc_invert, fwi_mc_sample, fwi_mc_forward (include/fwi.h).  Stateless:
// every call uploads, computes and frees; errors are returned as codes, the text through
// fwi_last_error(NULL).
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/fwi.h"
#include "mc_kernels.h"

namespace {

// `gerr = text` sets the text fwi_last_error(NULL) returns (owned by fwi_api.hip)
const struct {
    void operator=(const std::string &m) const { fwi::set_global_error(m); }
} gerr;

struct DevBuf {  // frees on scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct Event {  // destroyed on scope exit
    hipEvent_t e = nullptr;
    ~Event() { if (e) (void)hipEventDestroy(e); }
};

int mc_fail(int code, const char *what, hipError_t e) {
    gerr = std::string(what) + ": " + hipGetErrorString(e);
    return code;
}

#define MCCHK(call)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return mc_fail(e_ == hipErrorOutOfMemory ? FWI_ENOMEM : FWI_EHIP, #call, e_); \
    } while (0)

int mc_check_args(const char *fn, int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const void *a,
                  const void *b, const void *c) {
    if (k < 1 || n < 1 || t < 1 || nsamp < 1 || !a || !b || !c) {
        gerr = std::string(fn) + ": bad argument";
        return FWI_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        gerr = std::string(fn) + ": no HIP device available (this library has no CPU fallback)";
        return FWI_EHIP;
    }
    if (device < 0 || device >= ndev) {
        gerr = std::string(fn) + ": device ordinal out of range";
        return FWI_EINVAL;
    }
    return FWI_OK;
}

// Device sampler request of fwi_mc_invert (type < 0: none, the samples come from the host).
struct McSampler {
    int type = -1;
    uint64_t seed = 0;
    int64_t first = 0;
    double amplitude = 1.0;
    double *samples_out = nullptr, *frac_out = nullptr;
};

// Shared body of fwi_mc_score (host samples) and fwi_mc_invert (samples drawn on the device).
int mc_score_impl(const char *fn, int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp,
                  const double *green, const double *data, const double *samples, const McSampler &smp,
                  int32_t metric, int32_t normalise, int32_t all_at_once, double *similarity_out,
                  double *likelihood_out, double *posterior_out, double *kernel_ms_out) {
    int rc = mc_check_args(fn, device, k, n, t, nsamp, green, data, smp.type >= 0 ? (const void *)green : samples);
    if (rc) return rc;
    if (!similarity_out || metric < FWI_MC_VR || metric > FWI_MC_GAU) {
        gerr = "fwi_mc_score: bad metric or null output";
        return FWI_EINVAL;
    }
    const bool lane_kernel = n <= 9;  // keeps everything in registers: no limit on k
    if (!lane_kernel && fwi::mc_score_lds_bytes(k, n) > 64 * 1024) {
        gerr = std::string(fn) + ": k * n too large for the moment kernel's per-workgroup table";
        return FWI_EINVAL;
    }
    MCCHK(hipSetDevice(device));
    // data-only moments per trace (sum d, sum d^2, max|d|, sum d_i d_{i+1}, first, last) and the
    // noise level of gaussian_comparison (:580): mean |flattened (normalised) data[-60:-10]|
    std::vector<double> dmom((size_t)k * 6);
    for (int kk = 0; kk < k; ++kk) {
        const double *dk = data + (size_t)kk * t;
        double s1 = 0, s2 = 0, mx = 0, dd1 = 0;
        for (int e = 0; e < t; ++e) {
            s1 += dk[e];
            s2 += dk[e] * dk[e];
            mx = std::max(mx, std::fabs(dk[e]));
            if (e + 1 < t) dd1 += dk[e] * dk[e + 1];
        }
        double *m = &dmom[(size_t)kk * 6];
        m[0] = s1; m[1] = s2; m[2] = mx; m[3] = dd1; m[4] = dk[0]; m[5] = dk[t - 1];
    }
    double sigma = 0.0;
    {
        const int64_t len = (int64_t)k * t;  // numpy slice [-60:-10] of the flattened array
        const int64_t lo = std::max<int64_t>(0, len - 60), hi = std::max<int64_t>(0, len - 10);
        double acc = 0.0;
        for (int64_t i = lo; i < hi; ++i) {
            const double v = data[i];
            acc += std::fabs(normalise ? v / dmom[(size_t)(i / t) * 6 + 2] : v);
        }
        sigma = (hi > lo) ? acc / (double)(hi - lo) : NAN;
    }
    hipStream_t s = nullptr;  // default stream: this call is synchronous
    DevBuf G, Gt, d, M, dm, sim, like;
    const size_t mc_pad = 1024;  // bytes of zero padding behind Gt and d (>= 2 pipelined groups of 9 doubles)
    const size_t gb = (size_t)k * n * t * 8, db = (size_t)k * t * 8, mb = (size_t)n * nsamp * 8, sb = (size_t)nsamp * 8;
    MCCHK(hipMalloc(&G.p, gb));
    if (lane_kernel) {  // packed [k][t][n + 1] stream for the lane-per-sample kernel's scalar loads
        const size_t row = (size_t)n + 1, pb = (size_t)k * t * row * 8;
        std::vector<double> gt((size_t)k * t * row);
        for (int kk = 0; kk < k; ++kk)
            for (int e = 0; e < t; ++e) {
                double *r = &gt[((size_t)kk * t + e) * row];
                for (int j = 0; j < n; ++j) r[j] = green[((size_t)kk * n + j) * t + e];
                r[n] = data[(size_t)kk * t + e];
            }
        MCCHK(hipMalloc(&Gt.p, pb + mc_pad));  // the kernel's look-ahead reads a few rows past the end
        MCCHK(hipMemcpy(Gt.p, gt.data(), pb, hipMemcpyHostToDevice));
        MCCHK(hipMemset((char *)Gt.p + pb, 0, mc_pad));
    }
    MCCHK(hipMalloc(&d.p, db));
    MCCHK(hipMalloc(&M.p, mb));
    MCCHK(hipMalloc(&dm.p, dmom.size() * 8));
    MCCHK(hipMalloc(&sim.p, sb));
    MCCHK(hipMalloc(&like.p, sb));
    MCCHK(hipMemcpy(G.p, green, gb, hipMemcpyHostToDevice));
    MCCHK(hipMemcpy(d.p, data, db, hipMemcpyHostToDevice));
    DevBuf fr;
    if (smp.type >= 0) {
        MCCHK(hipMalloc(&fr.p, sb));
        hipError_t se = fwi::launch_mc_sample(smp.type, smp.seed, smp.first, nsamp, smp.amplitude, (double *)M.p,
                                              nsamp, (double *)fr.p, nullptr);
        if (se != hipSuccess) return mc_fail(FWI_EHIP, "mc_sample_kernel launch", se);
    } else {
        MCCHK(hipMemcpy(M.p, samples, mb, hipMemcpyHostToDevice));
    }
    MCCHK(hipMemcpy(dm.p, dmom.data(), dmom.size() * 8, hipMemcpyHostToDevice));
    Event e0, e1;
    MCCHK(hipEventCreate(&e0.e));
    MCCHK(hipEventCreate(&e1.e));
    MCCHK(hipEventRecord(e0.e, s));
    hipError_t le = fwi::launch_mc_score((const double *)G.p, (const double *)Gt.p, (const double *)d.p, (const double *)M.p,
                                         (const double *)dm.p, k, n, t, nsamp, metric, normalise != 0,
                                         all_at_once != 0, sigma, (double *)sim.p, (double *)like.p, s);
    if (le != hipSuccess) return mc_fail(FWI_EHIP, "mc_score_kernel launch", le);
    MCCHK(hipEventRecord(e1.e, s));
    MCCHK(hipMemcpy(similarity_out, sim.p, sb, hipMemcpyDeviceToHost));
    float ms = 0.f;
    MCCHK(hipEventElapsedTime(&ms, e0.e, e1.e));
    if (kernel_ms_out) *kernel_ms_out = ms;
    if (smp.samples_out) MCCHK(hipMemcpy(smp.samples_out, M.p, mb, hipMemcpyDeviceToHost));
    if (smp.frac_out) MCCHK(hipMemcpy(smp.frac_out, fr.p, sb, hipMemcpyDeviceToHost));
    if (likelihood_out) MCCHK(hipMemcpy(likelihood_out, like.p, sb, hipMemcpyDeviceToHost));
    if (posterior_out) {  // :847-848, p_model = 1/N; normalised on the device, `sim` reused for the result
        DevBuf acc;
        MCCHK(hipMalloc(&acc.p, (size_t)fwi::mc_posterior_scratch_doubles() * 8));
        hipError_t pe = fwi::launch_mc_posterior((const double *)like.p, nsamp, (double *)acc.p, (double *)sim.p,
                                                 nullptr);
        if (pe != hipSuccess) return mc_fail(FWI_EHIP, "mc_posterior_kernel launch", pe);
        MCCHK(hipMemcpy(posterior_out, sim.p, sb, hipMemcpyDeviceToHost));
    }
    return FWI_OK;
}

}  // namespace

extern "C" {

int fwi_mc_score(int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const double *green,
                 const double *data, const double *samples, int32_t metric, int32_t normalise,
                 int32_t all_at_once, double *similarity_out, double *likelihood_out, double *posterior_out,
                 double *kernel_ms_out) {
    return mc_score_impl("fwi_mc_score", device, k, n, t, nsamp, green, data, samples, McSampler(), metric,
                         normalise, all_at_once, similarity_out, likelihood_out, posterior_out, kernel_ms_out);
}

int fwi_mc_invert(int32_t device, int32_t inversion_type, uint64_t seed, int64_t first_sample, int64_t nsamp,
                  double amplitude, int32_t k, int32_t n, int32_t t, const double *green, const double *data,
                  int32_t metric, int32_t normalise, int32_t all_at_once, double *samples_out, double *frac_out,
                  double *similarity_out, double *likelihood_out, double *posterior_out, double *kernel_ms_out) {
    const int nc = fwi::mc_sampler_components(inversion_type);
    if (nc == 0 || nc != n || first_sample < 0) {
        gerr = nc == 0 ? "fwi_mc_invert: unknown inversion_type"
                                 : nc != n ? "fwi_mc_invert: the Green's functions must have the inversion type's "
                                             "number of components (6, 3 or 9)"
                                           : "fwi_mc_invert: negative first_sample";
        return FWI_EINVAL;
    }
    McSampler smp;
    smp.type = inversion_type;
    smp.seed = seed;
    smp.first = first_sample;
    smp.amplitude = amplitude;
    smp.samples_out = samples_out;
    smp.frac_out = frac_out;
    return mc_score_impl("fwi_mc_invert", device, k, n, t, nsamp, green, data, nullptr, smp, metric, normalise,
                         all_at_once, similarity_out, likelihood_out, posterior_out, kernel_ms_out);
}

int fwi_mc_sample(int32_t device, int32_t inversion_type, uint64_t seed, int64_t first_sample, int64_t nsamp,
                  double amplitude, double *samples_out, double *frac_out) {
    const int n = fwi::mc_sampler_components(inversion_type);
    if (n == 0 || nsamp < 1 || first_sample < 0 || !samples_out) {
        gerr = "fwi_mc_sample: unknown inversion_type, nsamp < 1, negative first_sample or null output";
        return FWI_EINVAL;
    }
    int rc = mc_check_args("fwi_mc_sample", device, 1, n, 1, nsamp, samples_out, samples_out, samples_out);
    if (rc) return rc;
    MCCHK(hipSetDevice(device));
    DevBuf M, fr;
    MCCHK(hipMalloc(&M.p, (size_t)n * nsamp * 8));
    MCCHK(hipMalloc(&fr.p, (size_t)nsamp * 8));
    hipError_t se = fwi::launch_mc_sample(inversion_type, seed, first_sample, nsamp, amplitude, (double *)M.p, nsamp,
                                          (double *)fr.p, nullptr);
    if (se != hipSuccess) return mc_fail(FWI_EHIP, "mc_sample_kernel launch", se);
    MCCHK(hipMemcpy(samples_out, M.p, (size_t)n * nsamp * 8, hipMemcpyDeviceToHost));
    if (frac_out) MCCHK(hipMemcpy(frac_out, fr.p, (size_t)nsamp * 8, hipMemcpyDeviceToHost));
    return FWI_OK;
}

int fwi_mc_forward(int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const double *green,
                   const double *samples, double *synth_out) {
    int rc = mc_check_args("fwi_mc_forward", device, k, n, t, nsamp, green, samples, synth_out);
    if (rc) return rc;
    MCCHK(hipSetDevice(device));
    DevBuf G, M, S;
    const size_t gb = (size_t)k * n * t * 8, mb = (size_t)n * nsamp * 8, sb = (size_t)nsamp * k * t * 8;
    MCCHK(hipMalloc(&G.p, gb));
    MCCHK(hipMalloc(&M.p, mb));
    MCCHK(hipMalloc(&S.p, sb));
    MCCHK(hipMemcpy(G.p, green, gb, hipMemcpyHostToDevice));
    MCCHK(hipMemcpy(M.p, samples, mb, hipMemcpyHostToDevice));
    hipError_t le = fwi::launch_mc_forward((const double *)G.p, (const double *)M.p, k, n, t, nsamp,
                                           (double *)S.p, nullptr);
    if (le != hipSuccess) return mc_fail(FWI_EHIP, "mc_forward_kernel launch", le);
    MCCHK(hipMemcpy(synth_out, S.p, sb, hipMemcpyDeviceToHost));
    return FWI_OK;
}

}  // extern "C"
