/* fwi.h -- C ABI of the MI355X-native acoustic forward / adjoint / gradient engine.
 *
 * Drop-in boundary.  The reference (Kevin2599/full_waveform_inversion) has NO
 * plugin / operator / FFI interface and no wave-propagation code
 * (SURVEY.md s.0, s.8b): its only seams are plain Python functions called
 * positionally -- forward_model(green_func_array, M)
 * (full_waveform_inversion.py:253), compare_synth_to_real_waveforms(...)
 * (:584) and perform_monte_carlo_sampled_waveform_inversion(...) (:786).
 * The entry points below are therefore the ones BASELINE.json's north_star
 * names -- forward(model, src, rec), adjoint(residual), gradient() -- bound
 * from Python through ctypes (full_waveform_inversion_amd/_lib.py); each
 * declaration says which north_star item / SURVEY.md s.8(a-1) row it serves
 * and, where one exists, the nearest reference line.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch / numpy types.
 *  - Every function returns 0 on success or an FWI_E* code; the message is
 *    available from fwi_last_error().  The library never calls exit()
 *    (the reference print+sys.exit()s, full_waveform_inversion.py:124-125).
 *  - Host buffers are caller-owned, C-contiguous, x fastest: a model is
 *    (nz, nx) or (nz, ny, nx); time series are (nt, n).  Their element type is
 *    the context's dtype (float for FWI_F32, double for FWI_F64).
 *  - The library owns all device memory inside the context.  One context per
 *    GPU, used from one host thread at a time.  It is stateful: fwi_adjoint()
 *    consumes what the last fwi_forward(save=1) stored, fwi_gradient()
 *    returns what the adjoint calls since the last fwi_gradient_reset()
 *    accumulated (the sum over this rank's shots).
 *  - Grid indices are int32 triples/pairs (z[, y], x).
 */
#ifndef FWI_H
#define FWI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWI_ABI_VERSION 13

enum { FWI_F32 = 0, FWI_F64 = 1 };

/* Stencil kernel selection (FWI_KERNEL_AUTO picks the fastest valid one). */
enum {
    FWI_KERNEL_AUTO = 0,
    FWI_KERNEL_POINT = 1,  /* one thread per grid point, neighbours through L1/L2 */
    FWI_KERNEL_STREAM = 2  /* 16-byte-per-lane kernels with LDS-staged halo tiles, any grid size: 3-D
                              z-marching register queue (step3d_stream, fp32 / fp64), 2-D fp32 row
                              tiles (step2d_tile) and 4-steps-per-launch tiles (step2d_fused) */
};

enum { FWI_WRT_VELOCITY = 0, FWI_WRT_SLOWNESS2 = 1 };
enum { FWI_UPDATE_STANDARD = 0, FWI_UPDATE_INCREMENT = 1 };
enum { FWI_ABC_SPONGE = 0, FWI_ABC_CPML = 1 };
enum { FWI_STORE_NATIVE = 0, FWI_STORE_BF16 = 1 };
enum { FWI_LAUNCH_AUTO = 0, FWI_LAUNCH_STREAM = 1, FWI_LAUNCH_GRAPH = 2 };

enum {
    FWI_OK = 0,
    FWI_EINVAL = 1,   /* bad argument / configuration */
    FWI_EHIP = 2,     /* a HIP runtime call failed */
    FWI_ESTATE = 3,   /* call order violated (e.g. adjoint before forward) */
    FWI_ENOMEM = 4,   /* device or host allocation failed */
    FWI_ECOMM = 5     /* RCCL failure */
};

typedef struct fwi_ctx fwi_ctx;

typedef struct fwi_config {
    int32_t struct_size; /* = sizeof(fwi_config); ABI guard */
    int32_t ndim;        /* 2 or 3 */
    int32_t nz, ny, nx;  /* ny is ignored (treated as 1) when ndim == 2 */
    int32_t order;       /* spatial order 2, 4 or 8 */
    int32_t nt_max;      /* largest number of time steps a shot will use */
    int32_t npml;        /* absorbing border width in cells; 0 = off */
    int32_t device;      /* HIP device ordinal */
    int32_t dtype;       /* FWI_F32 or FWI_F64 */
    int32_t kernel;      /* FWI_KERNEL_* */
    int32_t zchunk;      /* 3-D STREAM kernel: planes marched per workgroup; 0 = auto */
    int32_t ckpt_interval; /* 0: keep the imaging term of every step (nt_max x npts elements);
                              K > 0: snapshot the wavefields every K steps and recompute each
                              K-step forward segment during fwi_adjoint (one extra forward sweep,
                              ~(2 nt_max / K + K) x npts elements)  [SURVEY s.8f-3] */
    int32_t image_stride; /* 0 / 1: the imaging condition uses every time step (the exact discrete
                              gradient).  S > 1: the forward term is stored and correlated every S-th
                              step only, weighted by S (a Riemann sum over the oversampled time axis):
                              the store shrinks to ceil(nt_max / S) x npts elements and the adjoint
                              sweep skips the imaging traffic on the other steps.  Not combinable with
                              ckpt_interval. */
    int32_t update_form;  /* FWI_UPDATE_STANDARD: u' = A (2u - B u_prev + q).  FWI_UPDATE_INCREMENT: the same
                              recursion carried as (u, v = u - u_prev): v' = A (B v + q), u' = u + v'.  In fp32 the
                              round-off of the standard form is amplified by ~1 / (omega dt) (the cancellation in
                              2u - u_prev); the increment form rounds u' relative to u: ~4x smaller errors on
                              seismograms and gradient.  Kernels: 3-D stream (fp32; 20 instead of 16 B/update),
                              2-D fused (fp32; same traffic, the tile holds u twice, v and C) and the point kernel. */
    int32_t abc;          /* absorbing boundary of the npml border: FWI_ABC_SPONGE (damping factors A, B) or
                              FWI_ABC_CPML (convolutional PML: memory variables in the border only) */
    int32_t store_dtype;  /* forward-term store: FWI_STORE_NATIVE (the field type) or FWI_STORE_BF16 (fp32 contexts:
                              half the store and half its traffic; imaging error ~1e-3) */
    int32_t launch_mode;  /* How the launches of a sweep's time loop reach the GPU.  FWI_LAUNCH_STREAM: one by one on the
                              context's HIP stream, as they are formed.  FWI_LAUNCH_GRAPH: the whole time loop of a sweep
                              (every line / step / record launch of fwi_forward or fwi_adjoint) is captured into a hipGraph
                              and launched once -- same kernels, same arguments, same order, bit-identical results
                              (SURVEY s.7.2 "or capture the step loop in a hipGraph").  FWI_LAUNCH_AUTO (0, the default)
                              resolves to STREAM: measured on one context with the mode switched between sweeps
                              (profiles/r04_graph_probe.jsonl), the graph changes the loop time by -0.6 ... +1.4 % on every
                              configuration tried (2-D 256^2 ... 1024^2 with and without the CPML, 3-D 256^3 with one and
                              with two launches per step) and costs the host the same ~2.3 us per launch to build: a
                              recorded negative result, DESIGN.md s.4.  (was reserved0 up to ABI 11) */
    double h;            /* grid spacing (m) */
    double dt;           /* time step (s) */
    double sigma_max;    /* peak damping rate (1/s) of the absorbing border, >= 0 (used when npml > 0) */
    double pml_alpha_max; /* CPML only: peak of the frequency-shift profile alpha (1/s), ~ pi f0; 0 = plain PML */
} fwi_config;

/* Context life cycle.  [north_star: "thin ctypes C-ABI shim"; SURVEY s.8b] */
int fwi_create(const fwi_config *cfg, fwi_ctx **out);
void fwi_destroy(fwi_ctx *ctx);
/* Last error text of ctx (or of the failed fwi_create when ctx == NULL). */
const char *fwi_last_error(const fwi_ctx *ctx);
int fwi_abi_version(void);

/* Upload the velocity model c (m/s), model-shaped.  The `model` argument of
 * north_star's forward(model, src, rec).  [SURVEY s.8(a-1) row forward] */
int fwi_set_model(fwi_ctx *ctx, const void *c_host);

/* forward(model, src, rec): nt leapfrog steps of the 2-D/3-D O(2|4|8) stencil
 * with sponge update, injection of wavelet (nt, nsrc) at src_idx (nsrc, ndim)
 * and sampling at rec_idx (nrec, ndim) into seis_out (nt, nrec).  save != 0
 * keeps the per-step forward term the imaging condition needs.
 * [SURVEY s.8(a-1) rows forward / stencil / PML / source / receiver; the
 * structurally analogous reference slot is forward_model(),
 * full_waveform_inversion.py:253-264: unknowns -> synthetic seismograms] */
int fwi_forward(fwi_ctx *ctx, int32_t nt, int32_t nsrc, const int32_t *src_idx,
                const void *wavelet, int32_t nrec, const int32_t *rec_idx, int32_t save,
                void *seis_out);

/* forward() with OFF-GRID sources / receivers (multilinear interpolation onto the surrounding nodes).  The node
 * lists (src_idx / rec_idx, as for fwi_forward) hold every point's nodes contiguously: point p owns entries
 * pt_start[p] .. pt_start[p + 1] (at most 8 = 2^3; pt_start has npts + 1 entries), each with its interpolation
 * weight (context dtype).  Time series cross the boundary PER POINT -- wavelet (nt, nsrc_pts), seis_out
 * (nt, nrec_pts) -- and are scattered onto / gathered from the nodes on the device, the gather by a wave-level
 * __shfl_down reduction over each point's 8 lanes.  The following fwi_adjoint / fwi_misfit_l2 take and return
 * per-point series too.  npts = 0 with nodes = 0: no points of that kind.
 * [north_star "__shfl-based wavefront reductions for the receiver gather"; SURVEY s.8(a-1) row receiver sampling] */
int fwi_forward_spread(fwi_ctx *ctx, int32_t nt, int32_t nsrc_pts, int32_t nsrc_nodes, const int32_t *src_idx,
                       const int32_t *src_pt_start, const void *src_weight, const void *wavelet, int32_t nrec_pts,
                       int32_t nrec_nodes, const int32_t *rec_idx, const int32_t *rec_pt_start, const void *rec_weight,
                       int32_t save, void *seis_out);

/* adjoint(residual): reverse-time propagation of residual (nt, nrec) injected
 * at the receivers of the last forward.  image != 0 accumulates the zero-lag
 * forward x adjoint correlation into the gradient accumulator.  residual == NULL
 * back-propagates the residual fwi_misfit_l2() left on the device.  adj_src_out,
 * if not NULL, receives F^T residual as (nt, nsrc).
 * [SURVEY s.8(a-1) rows adjoint / imaging condition] */
int fwi_adjoint(fwi_ctx *ctx, const void *residual, int32_t image, void *adj_src_out);

/* Least-squares misfit on the device: with d_syn the seismograms of the last fwi_forward (still resident),
 * forms the residual r = d_syn - d_obs there, returns J = 1/2 sum r^2 (wave64 __shfl_down reduction, fp64
 * accumulate) and keeps r as the residual of the next fwi_adjoint(ctx, NULL, ...): neither the residual nor a
 * second copy of the data crosses PCIe.  [SURVEY s.8(a-1) row "gradient dot-products / J"] */
int fwi_misfit_l2(fwi_ctx *ctx, const void *d_obs /* (nt, nrec) */, double *J_out);

/* gradient(): copy out the accumulated gradient, model-shaped, as dJ/dc
 * (FWI_WRT_VELOCITY) or dJ/d(1/c^2) (FWI_WRT_SLOWNESS2).
 * [SURVEY s.8(a-1) row gradient] */
int fwi_gradient(fwi_ctx *ctx, int32_t wrt, void *g_out);
int fwi_gradient_reset(fwi_ctx *ctx);
/* dst accumulator += src accumulator (two contexts of the same shape on the same GPU): several
 * contexts can work through a rank's shots concurrently -- a 2-D shot cannot fill an MI355X on
 * its own -- and are summed on the device before the exchange. */
int fwi_gradient_add(fwi_ctx *dst, fwi_ctx *src);

/* Device-side reductions for the misfit and the optimiser's dot products:
 * sum_i a[i]*b[i] over n host elements of the context dtype, wave-shuffle
 * reduced on the GPU, fp64 accumulate.  [SURVEY s.8(a-1) row dot-products] */
int fwi_dot(fwi_ctx *ctx, const void *a_host, const void *b_host, int64_t n, double *out);

/* Device-resident, model-shaped vectors for the optimiser (L-BFGS history, search direction,
 * trial model): `count` slots owned by the context; all algebra runs on the GPU so an iteration
 * moves no model-sized array over PCIe.  [SURVEY s.8(a-1) rows L-BFGS / dot-products] */
int fwi_vec_create(fwi_ctx *ctx, int32_t count);
int fwi_vec_upload(fwi_ctx *ctx, int32_t slot, const void *host);
int fwi_vec_download(fwi_ctx *ctx, int32_t slot, void *host);
int fwi_vec_copy(fwi_ctx *ctx, int32_t dst, int32_t src);
int fwi_vec_axpby(fwi_ctx *ctx, int32_t y, double a, int32_t x, double b); /* y = a x + b y */
int fwi_vec_dot(fwi_ctx *ctx, int32_t x, int32_t y, double *out);
int fwi_vec_absmax(fwi_ctx *ctx, int32_t x, double *out);
int fwi_vec_clip(fwi_ctx *ctx, int32_t x, double lo, double hi);
/* model := slot (velocity), without leaving the device */
int fwi_set_model_vec(fwi_ctx *ctx, int32_t slot);
/* slot := accumulated gradient (after fwi_allreduce_gradient, if any), as fwi_gradient() */
int fwi_gradient_vec(fwi_ctx *ctx, int32_t wrt, int32_t slot);

/* Shot-parallel exchange: one RCCL communicator per context, sum of the
 * gradient accumulators over ranks (in place, on device).  The reference's
 * only counterpart is the gather-by-concatenation of Monte Carlo samples,
 * full_waveform_inversion.py:816-848.  [SURVEY s.8e] */
#define FWI_UNIQUE_ID_BYTES 128
int fwi_comm_unique_id(void *id_out /* FWI_UNIQUE_ID_BYTES */);
int fwi_comm_init(fwi_ctx *ctx, int32_t rank, int32_t nranks, const void *id);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): the evidence that
 * the N ranks of a job really share one communicator, echoed by bench.py as config.rccl_ranks. */
int fwi_comm_info(fwi_ctx *ctx, int32_t *nranks_out, int32_t *rank_out);
/* Tear the communicator down without waiting for peers (ncclCommAbort): the clean-up of a rank whose
 * job is being abandoned because another rank failed to join. */
int fwi_comm_abort(fwi_ctx *ctx);
int fwi_allreduce_gradient(fwi_ctx *ctx);
/* In-place sum / max over ranks of n <= 8 doubles (misfit values; the slowest rank's elapsed time). */
int fwi_allreduce_f64(fwi_ctx *ctx, double *vals, int32_t n);
int fwi_allreduce_f64_max(fwi_ctx *ctx, double *vals, int32_t n);

/* Measurement hooks (bench.py): device time of the last time loop, from HIP
 * events on the context's stream, and the synchronising fence. */
int fwi_last_loop_ms(fwi_ctx *ctx, double *ms_out);
/* Host side of the last time loop: wall time the calling thread spent forming and submitting its launches
 * (FWI_LAUNCH_GRAPH: capture + instantiation + the one graph launch), and of that the graph's capture + instantiation
 * alone (0 in stream mode).  Compared with fwi_last_loop_ms it tells whether a loop is bound by the host's launch rate. */
int fwi_last_host_ms(fwi_ctx *ctx, double *submit_ms_out, double *graph_build_ms_out);
/* Placement search (3-D fp32 stream contexts with the convolutional PML or in increment form, past the cache-resident
 * sizes): fwi_create times a few steps with some of the context's arrays at different offsets inside padded
 * allocations and keeps the fastest; results never depend on it.  Time per step before and after the search in
 * microseconds (0 = no search ran for this context) and, into shift_bytes_out[8], the chosen offsets in bytes of the
 * movable arrays in search order (CPML: ty, zeta_x, tz, psi_x, then v in increment form; increment form without CPML:
 * v, C; unused entries 0).  Any out pointer may be NULL.  No reference counterpart. */
int fwi_placement_info(fwi_ctx *ctx, double *us_before_out, double *us_after_out, int64_t *shift_bytes_out);
/* Change fwi_config.launch_mode of a live context (takes effect with the next sweep): the A/B of stream launches against
 * hipGraph launches on ONE context, the same buffers and the same cache state (tools/graph_probe.py). */
int fwi_set_launch_mode(fwi_ctx *ctx, int32_t mode);
int fwi_synchronize(fwi_ctx *ctx);
/* Layout invariant check (tests): the number of cells of the context's padded fields (wavefields, dt^2 c^2, the increment
 * field, the spare / recomputation pairs) that lie OUTSIDE the grid's interior -- halo planes and rows, the x halo a
 * row shares with the next, the look-ahead planes, the tail -- and are not exactly zero.  Every kernel relies on those
 * cells being zero and never writes them; anything but 0 here is a bug. */
int fwi_check_padding(fwi_ctx *ctx, int64_t *dirty_out);
/* Name of the stencil kernel the context dispatches to (static string). */
const char *fwi_kernel_name(const fwi_ctx *ctx);
/* Device count / name without creating a context (returns FWI_EHIP if none). */
int fwi_device_count(int32_t *n_out);

/* ---------------------------------------------------------------------------
 * the reference's REAL hot loop, batched over N source samples.
 * Steps 4-7 of PARALLEL_worker_mc_inv (full_waveform_inversion.py:713-774) for
 * given samples: forward_model (:253-264), compare_synth_to_real_waveforms
 * (:584-684) with one of the five similarity metrics (:512-582), the likelihood
 * map exp(-(1-s)/2) (:774) and the posterior normalisation (:847-848).  fp64 like
 * the reference; the samples are the caller's (fwi_mc_invert draws them on the device).
 *   green   (k, n, t)   Green's functions        data (k, t)   observed traces
 *   samples (n, nsamp)  source vectors, the reference's MTs[:, i] layout
 * Stateless: buffers are uploaded, scored and freed inside the call.
 * ------------------------------------------------------------------------- */
enum { FWI_MC_VR = 0, FWI_MC_CC = 1, FWI_MC_PCC = 2, FWI_MC_CCSHIFT = 3, FWI_MC_GAU = 4 };

int fwi_mc_score(int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const double *green,
                 const double *data, const double *samples, int32_t metric, int32_t normalise,
                 int32_t all_at_once, double *similarity_out /* nsamp */,
                 double *likelihood_out /* nsamp or NULL */, double *posterior_out /* nsamp or NULL */,
                 double *kernel_ms_out /* or NULL */);

/* The whole loop body of PARALLEL_worker_mc_inv (:713-774) on the device, samplers included:
 * sample i = first_sample .. first_sample + nsamp - 1 of `inversion_type` is drawn by the
 * reference's generate_random_* map (:282-510) from counter-based deviates (Philox4x32-10 keyed by
 * `seed`, counter = sample index; a pure function of (seed, i), so ranks / calls can split a
 * run by index ranges), scaled by `amplitude` (M_amplitude, :741-760) and scored like
 * fwi_mc_score.  n must be the type's component count (6, 3 or 9).  samples_out (n, nsamp) and
 * frac_out (nsamp; the sampler's amplitude fraction, 0 for the uncoupled types) may be NULL when
 * only the scores are wanted -- then no sample ever crosses PCIe. */
enum {
    FWI_MC_FULL_MT = 0,                        /* generate_random_MT                              :282 */
    FWI_MC_DC = 1,                             /* generate_random_DC_MT                           :295 */
    FWI_MC_SINGLE_FORCE = 2,                   /* generate_random_single_force_vector             :320 */
    FWI_MC_DC_SINGLE_FORCE_COUPLE = 3,         /* generate_random_DC_single_force_coupled_tensor  :333 */
    FWI_MC_DC_SINGLE_FORCE_NO_COUPLING = 4,    /* generate_random_DC_single_force_uncoupled_tensor :369 */
    FWI_MC_DC_CRACK_COUPLE = 5,                /* generate_random_DC_crack_coupled_tensor         :384 */
    FWI_MC_SINGLE_FORCE_CRACK_NO_COUPLING = 6  /* generate_random_single_force_crack_uncoupled_tensor :448 */
};

int fwi_mc_invert(int32_t device, int32_t inversion_type, uint64_t seed, int64_t first_sample, int64_t nsamp,
                  double amplitude, int32_t k, int32_t n, int32_t t, const double *green, const double *data,
                  int32_t metric, int32_t normalise, int32_t all_at_once,
                  double *samples_out /* (n, nsamp) or NULL */, double *frac_out /* nsamp or NULL */,
                  double *similarity_out /* nsamp */, double *likelihood_out /* nsamp or NULL */,
                  double *posterior_out /* nsamp or NULL */, double *kernel_ms_out /* or NULL */);

/* A run in many blocks (N beyond one call's memory, or streamed): the plan keeps the Green's
 * functions, the data and every work buffer on the device, so each block costs its kernels and its
 * downloads only.  Blocks hold at most max_samples samples.  *like_sum_out receives the block's
 * sum of likelihoods; the posterior over the whole run is likelihood / (sum of the blocks'
 * like_sum) (:847-848), formed by the caller.  One plan per GPU, one host thread at a time. */
typedef struct fwi_mc_plan fwi_mc_plan;
int fwi_mc_plan_create(int32_t device, int32_t k, int32_t n, int32_t t, const double *green, const double *data,
                       int64_t max_samples, fwi_mc_plan **out);
void fwi_mc_plan_destroy(fwi_mc_plan *plan);
int fwi_mc_plan_invert(fwi_mc_plan *plan, int32_t inversion_type, uint64_t seed, int64_t first_sample,
                       int64_t nsamp, double amplitude, int32_t metric, int32_t normalise, int32_t all_at_once,
                       double *samples_out /* (n, nsamp) or NULL */, double *frac_out /* nsamp or NULL */,
                       double *similarity_out /* nsamp */, double *likelihood_out /* nsamp or NULL */,
                       double *like_sum_out /* or NULL */, double *kernel_ms_out /* or NULL */);
int fwi_mc_plan_score(fwi_mc_plan *plan, int64_t nsamp, const double *samples /* (n, nsamp) */, int32_t metric,
                      int32_t normalise, int32_t all_at_once, double *similarity_out, double *likelihood_out,
                      double *like_sum_out, double *kernel_ms_out);

/* The device sampler alone: samples_out (n, nsamp), frac_out (nsamp) or NULL. */
int fwi_mc_sample(int32_t device, int32_t inversion_type, uint64_t seed, int64_t first_sample, int64_t nsamp,
                  double amplitude, double *samples_out, double *frac_out);

/* forward_model for a batch: synth_out (nsamp, k, t).  [full_waveform_inversion.py:253-264] */
int fwi_mc_forward(int32_t device, int32_t k, int32_t n, int32_t t, int64_t nsamp, const double *green,
                   const double *samples, double *synth_out);

#ifdef __cplusplus
}
#endif
#endif /* FWI_H */
