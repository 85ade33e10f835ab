// Internal launch interface between the Monte Carlo C-ABI (mc_api.hip) and its gfx950 kernels
// (mc_kernels.hip), and the error slot shared with fwi_api.hip.  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace fwi {
hipError_t launch_mc_sample(int type, uint64_t seed, int64_t first, int64_t nsamp, double amplitude, double *Ms,
                            int64_t ld, double *frac, hipStream_t s);
int mc_sampler_components(int type);
hipError_t launch_mc_posterior(const double *like, int64_t n, double *scratch, double *post, hipStream_t s);
int mc_posterior_scratch_doubles();
hipError_t launch_mc_score(const double *G, const double *Gt, const double *d, const double *Ms, const double *dmom,
                           int k, int n, int t, int64_t nsamp, int metric, int normalise, int all_at_once,
                           double gau_sigma, double *sim, double *like, hipStream_t s);
hipError_t launch_mc_forward(const double *G, const double *Ms, int k, int n, int t, int64_t nsamp, double *synth,
                             hipStream_t s);
size_t mc_score_lds_bytes(int k, int n);

// Text returned by fwi_last_error(NULL): calls that have no context report through it (thread-local).
void set_global_error(const std::string &msg);
}  // namespace fwi
