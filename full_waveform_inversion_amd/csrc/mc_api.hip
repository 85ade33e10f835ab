// C-ABI of the reference's real hot loop: batched sampling and scoring of source
// samples -- fwi_mc_score, fwi_mc_invert, fwi_mc_sample, fwi_mc_forward (include/fwi.h).  Stateless:
// every call uploads, computes and frees; errors are returned as codes, the text through
// fwi_last_error(NULL).
#include <algorithm>
#include <cmath>
#include <new>
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

}  // namespace

// Device state of one (Green's functions, data) problem: everything that does not depend on the
// samples.  fwi_mc_score / fwi_mc_invert build one on the stack for a single call; fwi_mc_plan_*
// keep it alive so that a long run in many blocks uploads and allocates once.
struct fwi_mc_plan {
    int device = 0, k = 0, n = 0, t = 0;
    int64_t cap = 0;             // samples per call the buffers hold
    bool lane_kernel = false;
    std::vector<double> data;    // host copy (k, t): the gaussian noise level depends on `normalise`
    std::vector<double> dmom;    // per-trace data moments
    DevBuf G, Gt, d, dm, M, fr, sim, like, scratch;
};

namespace {

int plan_init(fwi_mc_plan &P, const char *fn, int32_t device, int32_t k, int32_t n, int32_t t, const double *green,
              const double *data, int64_t cap) {
    int rc = mc_check_args(fn, device, k, n, t, cap, green, data, green);
    if (rc) return rc;
    P.device = device; P.k = k; P.n = n; P.t = t; P.cap = cap;
    P.lane_kernel = n <= 9;  // keeps everything in registers: no limit on k
    if (!P.lane_kernel && fwi::mc_score_lds_bytes(k, n) > 64 * 1024) {
        gerr = std::string(fn) + ": k * n too large for the moment kernel's per-workgroup table";
        return FWI_EINVAL;
    }
    MCCHK(hipSetDevice(device));
    P.data.assign(data, data + (size_t)k * t);
    // data-only moments per trace: sum d, sum d^2, max|d|, sum d_i d_{i+1}, first, last
    P.dmom.assign((size_t)k * 6, 0.0);
    for (int kk = 0; kk < k; ++kk) {
        const double *dk = data + (size_t)kk * t;
        double s1 = 0, s2 = 0, mx = 0, dd1 = 0;
        for (int e = 0; e < t; ++e) {
            s1 += dk[e];
            s2 += dk[e] * dk[e];
            mx = std::max(mx, std::fabs(dk[e]));
            if (e + 1 < t) dd1 += dk[e] * dk[e + 1];
        }
        double *m = &P.dmom[(size_t)kk * 6];
        m[0] = s1; m[1] = s2; m[2] = mx; m[3] = dd1; m[4] = dk[0]; m[5] = dk[t - 1];
    }
    const size_t mc_pad = 1024;  // bytes of zero padding behind Gt (>= 3 pipelined groups of 10 doubles)
    const size_t gb = (size_t)k * n * t * 8, db = (size_t)k * t * 8, sb = (size_t)cap * 8;
    MCCHK(hipMalloc(&P.G.p, gb));
    if (P.lane_kernel) {  // packed [k][t][n + 1] stream for the lane-per-sample kernel's scalar loads
        const size_t row = (size_t)n + 1, pb = (size_t)k * t * row * 8;
        std::vector<double> gt((size_t)k * t * row);
        for (int kk = 0; kk < k; ++kk)
            for (int e = 0; e < t; ++e) {
                double *r = &gt[((size_t)kk * t + e) * row];
                for (int j = 0; j < n; ++j) r[j] = green[((size_t)kk * n + j) * t + e];
                r[n] = data[(size_t)kk * t + e];
            }
        MCCHK(hipMalloc(&P.Gt.p, pb + mc_pad));  // the kernel's look-ahead reads a few rows past the end
        MCCHK(hipMemcpy(P.Gt.p, gt.data(), pb, hipMemcpyHostToDevice));
        MCCHK(hipMemset((char *)P.Gt.p + pb, 0, mc_pad));
    }
    MCCHK(hipMalloc(&P.d.p, db));
    MCCHK(hipMalloc(&P.dm.p, P.dmom.size() * 8));
    MCCHK(hipMalloc(&P.M.p, (size_t)n * cap * 8));
    MCCHK(hipMalloc(&P.fr.p, sb));
    MCCHK(hipMalloc(&P.sim.p, sb));
    MCCHK(hipMalloc(&P.like.p, sb));
    MCCHK(hipMalloc(&P.scratch.p, (size_t)fwi::mc_posterior_scratch_doubles() * 8));
    MCCHK(hipMemcpy(P.G.p, green, gb, hipMemcpyHostToDevice));
    MCCHK(hipMemcpy(P.d.p, data, db, hipMemcpyHostToDevice));
    MCCHK(hipMemcpy(P.dm.p, P.dmom.data(), P.dmom.size() * 8, hipMemcpyHostToDevice));
    return FWI_OK;
}

// One block of samples through the plan: host samples (smp.type < 0) or device-drawn ones.
int plan_run(fwi_mc_plan &P, const char *fn, int64_t nsamp, const double *samples, const McSampler &smp,
             int32_t metric, int32_t normalise, int32_t all_at_once, double *similarity_out, double *likelihood_out,
             double *posterior_out, double *like_sum_out, double *kernel_ms_out) {
    if (nsamp < 1 || nsamp > P.cap || (smp.type < 0 && !samples)) {
        gerr = std::string(fn) + ": nsamp outside [1, capacity] or null samples";
        return FWI_EINVAL;
    }
    if (!similarity_out || metric < FWI_MC_VR || metric > FWI_MC_GAU) {
        gerr = std::string(fn) + ": bad metric or null output";
        return FWI_EINVAL;
    }
    MCCHK(hipSetDevice(P.device));
    const int k = P.k, n = P.n, t = P.t;
    // noise level of gaussian_comparison (:580): mean |flattened (normalised) data[-60:-10]|
    double sigma = 0.0;
    {
        const int64_t len = (int64_t)k * t;  // numpy slice [-60:-10] of the flattened array
        const int64_t lo = std::max<int64_t>(0, len - 60), hi = std::max<int64_t>(0, len - 10);
        double acc = 0.0;
        for (int64_t i = lo; i < hi; ++i) {
            const double v = P.data[i];
            acc += std::fabs(normalise ? v / P.dmom[(size_t)(i / t) * 6 + 2] : v);
        }
        sigma = (hi > lo) ? acc / (double)(hi - lo) : NAN;
    }
    hipStream_t s = nullptr;  // default stream: the call is synchronous
    const size_t mb = (size_t)n * nsamp * 8, sb = (size_t)nsamp * 8;
    if (smp.type >= 0) {
        hipError_t se = fwi::launch_mc_sample(smp.type, smp.seed, smp.first, nsamp, smp.amplitude, (double *)P.M.p,
                                              nsamp, (double *)P.fr.p, s);
        if (se != hipSuccess) return mc_fail(FWI_EHIP, "mc_sample_kernel launch", se);
    } else {
        MCCHK(hipMemcpy(P.M.p, samples, mb, hipMemcpyHostToDevice));
    }
    Event e0, e1;
    MCCHK(hipEventCreate(&e0.e));
    MCCHK(hipEventCreate(&e1.e));
    MCCHK(hipEventRecord(e0.e, s));
    hipError_t le = fwi::launch_mc_score((const double *)P.G.p, (const double *)P.Gt.p, (const double *)P.d.p,
                                         (const double *)P.M.p, (const double *)P.dm.p, k, n, t, nsamp, metric,
                                         normalise != 0, all_at_once != 0, sigma, (double *)P.sim.p,
                                         (double *)P.like.p, s);
    if (le != hipSuccess) return mc_fail(FWI_EHIP, "mc_score_kernel launch", le);
    MCCHK(hipEventRecord(e1.e, s));
    MCCHK(hipMemcpy(similarity_out, P.sim.p, sb, hipMemcpyDeviceToHost));
    float ms = 0.f;
    MCCHK(hipEventElapsedTime(&ms, e0.e, e1.e));
    if (kernel_ms_out) *kernel_ms_out = ms;
    if (smp.samples_out) MCCHK(hipMemcpy(smp.samples_out, P.M.p, mb, hipMemcpyDeviceToHost));
    if (smp.frac_out) MCCHK(hipMemcpy(smp.frac_out, P.fr.p, sb, hipMemcpyDeviceToHost));
    if (likelihood_out) MCCHK(hipMemcpy(likelihood_out, P.like.p, sb, hipMemcpyDeviceToHost));
    if (posterior_out || like_sum_out) {
        // :847-848, p_model = 1/N of THIS block; normalised on the device, `sim` reused for the result.
        // A run in several blocks renormalises with the sum of the blocks' like_sum on the host.
        hipError_t pe = fwi::launch_mc_posterior((const double *)P.like.p, nsamp, (double *)P.scratch.p,
                                                 (double *)P.sim.p, s);
        if (pe != hipSuccess) return mc_fail(FWI_EHIP, "mc_posterior_kernel launch", pe);
        if (posterior_out) MCCHK(hipMemcpy(posterior_out, P.sim.p, sb, hipMemcpyDeviceToHost));
        if (like_sum_out) {
            double p_data = 0.0;  // = sum_i L_i / nsamp
            MCCHK(hipMemcpy(&p_data, (double *)P.scratch.p + (fwi::mc_posterior_scratch_doubles() - 1), 8,
                            hipMemcpyDeviceToHost));
            *like_sum_out = p_data * (double)nsamp;
        }
    }
    return FWI_OK;
}

int sampler_ok(const char *fn, int32_t inversion_type, int32_t n, int64_t first_sample) {
    const int nc = fwi::mc_sampler_components(inversion_type);
    if (nc == 0 || nc != n || first_sample < 0) {
        gerr = std::string(fn) + (nc == 0 ? ": unknown inversion_type"
                                  : nc != n ? ": the Green's functions must have the inversion type's number of "
                                              "components (6, 3 or 9)"
                                            : ": negative first_sample");
        return FWI_EINVAL;
    }
    return FWI_OK;
}

}  // namespace

extern "C" {

int fwi_mc_score(int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const double *green,
                 const double *data, const double *samples, int32_t metric, int32_t normalise,
                 int32_t all_at_once, double *similarity_out, double *likelihood_out, double *posterior_out,
                 double *kernel_ms_out) {
    if (!samples) {
        gerr = "fwi_mc_score: bad argument";
        return FWI_EINVAL;
    }
    fwi_mc_plan P;
    int rc = plan_init(P, "fwi_mc_score", device, k, n, t, green, data, nsamp);
    if (rc) return rc;
    return plan_run(P, "fwi_mc_score", nsamp, samples, McSampler(), metric, normalise, all_at_once, similarity_out,
                    likelihood_out, posterior_out, nullptr, kernel_ms_out);
}

int fwi_mc_invert(int32_t device, int32_t inversion_type, uint64_t seed, int64_t first_sample, int64_t nsamp,
                  double amplitude, int32_t k, int32_t n, int32_t t, const double *green, const double *data,
                  int32_t metric, int32_t normalise, int32_t all_at_once, double *samples_out, double *frac_out,
                  double *similarity_out, double *likelihood_out, double *posterior_out, double *kernel_ms_out) {
    int rc = sampler_ok("fwi_mc_invert", inversion_type, n, first_sample);
    if (rc) return rc;
    McSampler smp;
    smp.type = inversion_type;
    smp.seed = seed;
    smp.first = first_sample;
    smp.amplitude = amplitude;
    smp.samples_out = samples_out;
    smp.frac_out = frac_out;
    fwi_mc_plan P;
    rc = plan_init(P, "fwi_mc_invert", device, k, n, t, green, data, nsamp);
    if (rc) return rc;
    return plan_run(P, "fwi_mc_invert", nsamp, nullptr, smp, metric, normalise, all_at_once, similarity_out,
                    likelihood_out, posterior_out, nullptr, kernel_ms_out);
}

int fwi_mc_plan_create(int32_t device, int32_t k, int32_t n, int32_t t, const double *green, const double *data,
                       int64_t max_samples, fwi_mc_plan **out) {
    if (!out) {
        gerr = "fwi_mc_plan_create: null output";
        return FWI_EINVAL;
    }
    *out = nullptr;
    fwi_mc_plan *P = new (std::nothrow) fwi_mc_plan;
    if (!P) return FWI_ENOMEM;
    int rc = plan_init(*P, "fwi_mc_plan_create", device, k, n, t, green, data, max_samples);
    if (rc) {
        delete P;
        return rc;
    }
    *out = P;
    return FWI_OK;
}

void fwi_mc_plan_destroy(fwi_mc_plan *plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    delete plan;
}

int fwi_mc_plan_invert(fwi_mc_plan *plan, int32_t inversion_type, uint64_t seed, int64_t first_sample,
                       int64_t nsamp, double amplitude, int32_t metric, int32_t normalise, int32_t all_at_once,
                       double *samples_out, double *frac_out, double *similarity_out, double *likelihood_out,
                       double *like_sum_out, double *kernel_ms_out) {
    if (!plan) {
        gerr = "fwi_mc_plan_invert: null plan";
        return FWI_EINVAL;
    }
    int rc = sampler_ok("fwi_mc_plan_invert", inversion_type, plan->n, first_sample);
    if (rc) return rc;
    McSampler smp;
    smp.type = inversion_type;
    smp.seed = seed;
    smp.first = first_sample;
    smp.amplitude = amplitude;
    smp.samples_out = samples_out;
    smp.frac_out = frac_out;
    return plan_run(*plan, "fwi_mc_plan_invert", nsamp, nullptr, smp, metric, normalise, all_at_once, similarity_out,
                    likelihood_out, nullptr, like_sum_out, kernel_ms_out);
}

int fwi_mc_plan_score(fwi_mc_plan *plan, int64_t nsamp, const double *samples, int32_t metric, int32_t normalise,
                      int32_t all_at_once, double *similarity_out, double *likelihood_out, double *like_sum_out,
                      double *kernel_ms_out) {
    if (!plan) {
        gerr = "fwi_mc_plan_score: null plan";
        return FWI_EINVAL;
    }
    return plan_run(*plan, "fwi_mc_plan_score", nsamp, samples, McSampler(), metric, normalise, all_at_once,
                    similarity_out, likelihood_out, nullptr, like_sum_out, kernel_ms_out);
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
