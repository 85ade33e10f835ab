// C-ABI of the engine (include/fwi.h): context, device memory, time loops,
// RCCL exchange.  Host-side runtime only; the arithmetic is in fwi_kernels.hip.
//
// No reference counterpart (the reference has neither a native layer nor this
// path); the entry points are the ones BASELINE.json's
// north_star names.  The time loop mirrors oracle/fwi_oracle.py::_propagate
// one-to-one so the parity tests can compare every intermediate.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/fwi.h"
#include "fwi_kernels.h"

using namespace fwi;

namespace {

thread_local std::string g_create_error;

const double COEF2[] = {-2.0, 1.0};
const double COEF4[] = {-5.0 / 2.0, 4.0 / 3.0, -1.0 / 12.0};
const double COEF8[] = {-205.0 / 72.0, 8.0 / 5.0, -1.0 / 5.0, 8.0 / 315.0, -1.0 / 560.0};
// centred first-difference weights d_1 .. d_r (CPML)
const double DCOEF2[] = {1.0 / 2.0};
const double DCOEF4[] = {2.0 / 3.0, -1.0 / 12.0};
const double DCOEF8[] = {4.0 / 5.0, -1.0 / 5.0, 4.0 / 105.0, -1.0 / 280.0};

std::string vformat(const char *fmt, va_list ap) {
    char buf[512];
    vsnprintf(buf, sizeof buf, fmt, ap);
    return buf;
}

}  // namespace

struct fwi_ctx {
    fwi_config cfg;
    GridDesc gd;
    int kernel = K_POINT;
    StreamTuning tune{8, 0, 1, 256};
    size_t esize = 4;  // bytes per element
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_loop_time = false;
    // FWI_LAUNCH_GRAPH: the time loop of a sweep is captured and launched as one hipGraph.  The executable graph of
    // the last sweep is kept until the next sweep (or destroy) replaces it -- it may still be running when the call
    // that launched it returns its samples.
    bool graph_mode = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    double host_submit_ms = 0.0, host_graph_ms = 0.0;

    // device fields
    void *u[2] = {nullptr, nullptr};  // padded wavefields (ping-pong)
    bool inc = false;                 // increment form: state (u, v = u - u_prev); u[] ping-pongs u, v lives in vf
    void *vf = nullptr, *fwv = nullptr;  // v of the running sweep / of the checkpointed forward recomputation
    // 3-D temporal blocking (fwi_pair3d.hip): forward sweeps without imaging advance two steps per pass
    bool qbf16 = false;  // forward-term store in bf16 (fp32 3-D stream contexts)
    size_t qes = 4;      // bytes per stored forward-term element
    bool pair3d = false;
    int pair_zc = 0, pair_tw = 256;
    int fused_skipd = -1;  // Fused2dArgs::skipd (FWI_FUSED2D_SKIPD, read at create)
    int *fused_order = nullptr;  // Fused2dArgs::tile_order (device; grids of more than one round of tiles with a border)
    int fused_ft = FUSED2D_TILE;  // interior tile edge of the fused 2-D kernel (fused2d_pick_tile; 64 with CPML / increment)
    // convolutional PML: memory variables per axis (z, y, x), compact over that axis' border, and 1-D coefficients
    bool cpml = false;
    bool xpml = false;  // 3-D fp32 stream contexts: the x border's recursion runs inside the step kernel
    int pml_lines = 0;  // axes (z = 1, y = 2) whose border runs as one line launch per step (pml_line_axes)
    // ... whose term T_d the line launch hands to the step kernel: arrays compact over the axis' shell (StepArgs::pml_tz /
    // pml_ty; scratch within one time step, so one pair serves every sweep of the context)
    void *pml_tz = nullptr, *pml_ty = nullptr;
    void *pml_psi[3] = {nullptr, nullptr, nullptr}, *pml_zeta[3] = {nullptr, nullptr, nullptr};
    // Placement of the four small arrays the 3-D CPML step kernel streams beside the fields (the x border's psi and zeta,
    // the handed-over terms tz and ty): each sits `place_shift` bytes into an allocation `place_pad` bytes longer than
    // the array, at the position Impl::tune_placement measured as the fastest (0 = untuned; see there)
    // (place_kind 2, increment-form contexts without CPML: the fields v and C instead -- worth 1.5 - 3 % there, the
    // per-process spread of that form is not theirs to fix; the u buffers change roles with others and stay put)
    int place_kind = 0;
    size_t place_pad = 0;
    int nmov = 0;                 // the movable arrays in search order: the member that points at them, their size,
    void **mov_slot[8] = {};      // how far into their allocation they currently sit
    size_t mov_bytes[8] = {}, mov_shift[8] = {};
    void movable(void **slot, size_t bytes) {
        if (place_pad && nmov < 8) {
            mov_slot[nmov] = slot;
            mov_bytes[nmov] = bytes;
            mov_shift[nmov++] = 0;
        }
    }
    float place_us[2] = {0.f, 0.f};  // step time before / after the placement search (fwi_placement_info)
    // with checkpointing: the memory variables of the forward recomputation (the running adjoint sweep keeps its own
    // in the set above) and, per snapshot, a copy of the forward set (psi then zeta, axis by axis)
    void *pml_psi_fw[3] = {nullptr, nullptr, nullptr}, *pml_zeta_fw[3] = {nullptr, nullptr, nullptr};
    // 2-D contexts that carry the CPML inside the fused launch: the set of arrays the launch writes (swapped with the
    // set it read after every launch; zeroed once -- the pad columns of the z border's rows are never written)
    void *pml_spare_psi[3] = {nullptr, nullptr, nullptr}, *pml_spare_zeta[3] = {nullptr, nullptr, nullptr};
    void *pml_snap = nullptr;
    size_t pml_snap_stride = 0;  // bytes of one snapshot of the memory variables
    void *pml_a[3] = {nullptr, nullptr, nullptr}, *pml_b[3] = {nullptr, nullptr, nullptr};
    size_t pml_bytes[3] = {0, 0, 0};
    void *C = nullptr;                // padded dt^2 c^2
    void *c_dev = nullptr;            // compact velocity
    void *dz = nullptr, *dy = nullptr, *dx = nullptr;
    void *q_store = nullptr;  // nt_max x npts forward terms
    int istride = 1;          // imaging stride: the forward term is stored / correlated every istride-th step
    void *logical = nullptr;  // nz x ny x nx staging array for host <-> compact copies when cx != nx
    void *g_acc = nullptr;    // compact gradient accumulator
    void *g_out = nullptr;    // compact scratch for fwi_gradient
    double *red = nullptr;    // reduction scalars

    // host copies
    std::vector<double> pz, py, px;
    bool have_model = false;

    // per-shot point sets (device) and their sizes
    int nt = 0, nsrc = 0, nrec = 0;
    bool have_forward = false, have_q = false;
    bool have_syn = false;  // ctx->series still holds the last forward's synthetics (an adjoint sweep records its
                            // source-side series into the same buffer)
    bool have_dev_residual = false;  // ctx->amp holds the residual fwi_misfit_l2 formed on the device
    // A point set on the device: original order (sampling, POINT-kernel injection) and the
    // copy sorted by stream-kernel tile with its CSR offsets (fused injection).
    struct PointSet {
        int n = 0;
        void *pidx = nullptr, *cidx = nullptr, *cu = nullptr, *cq = nullptr;
        void *s_start = nullptr, *s_pidx = nullptr, *s_cidx = nullptr, *s_cu = nullptr, *s_cq = nullptr,
             *s_col = nullptr, *s_run = nullptr;
        size_t cap = 0, cap_start = 0;
        // tables for the 2-D fused kernel: injection entries per tile (extended region) and
        // sampling entries per tile (interior)
        void *fi_start = nullptr, *fi_lz = nullptr, *fi_lx = nullptr, *fi_col = nullptr, *fi_int = nullptr,
             *fi_cidx = nullptr, *fi_cu = nullptr, *fi_cq = nullptr, *fi_run = nullptr;
        void *fr_start = nullptr, *fr_lz = nullptr, *fr_lx = nullptr, *fr_col = nullptr;
        size_t fcap = 0, fcap_start = 0;
        // entries of the 3-D two-step kernel, CSR over its workgroups
        void *pr_start = nullptr, *pr_ent = nullptr;
        size_t pcap = 0, pcap_start = 0;
    } src, rec;
    // Off-grid points of the last forward (fwi_forward_spread): per point set the CSR over points, the owner of
    // every node entry and its interpolation weight; time series cross the boundary per POINT and are scattered /
    // gathered on the device.  npts = 0: node-based call.
    struct SpreadSet {
        int npts = 0;
        void *pt_start = nullptr, *owner = nullptr, *weight = nullptr;
        size_t cap = 0;
    } src_sp, rec_sp;
    void *pts_a = nullptr;   // (nt, npts) per-point series being uploaded / downloaded
    void *pts_d = nullptr;   // (nt, nrec points) gathered synthetics of the last forward (fwi_misfit_l2)
    size_t cap_pts_a = 0, cap_pts_d = 0;
    void *wav = nullptr;     // (nt, nsrc) source wavelets of the last forward (kept for recomputation)
    void *amp = nullptr;     // (nt, nrec) residual being back-propagated
    void *series = nullptr;  // (nt, n) sampled series of the running sweep
    size_t cap_wav = 0, cap_amp = 0, cap_series = 0;
    std::vector<void *> vecs;  // optimiser vectors (compact, model-sized)
    void *pin = nullptr;     // pinned host staging for the time series
    size_t cap_pin = 0;
    // checkpointing (SURVEY s.8f-3): snapshot of (u^n, u^{n-1}) every `ckpt` steps instead of the
    // imaging term of every step; q_store then holds ckpt + 1 slots and fwd[] the recomputed fields
    int ckpt = 0;
    bool ckpt_ready = false;  // every checkpoint buffer below is allocated
    void *snap = nullptr, *fwd[2] = {nullptr, nullptr};
    // 2-D temporal blocking: second buffer pair the fused kernel writes into, and whether it is used
    bool fused2d = false;
    void *fx[2] = {nullptr, nullptr};   // second pair for the time sweep in flight
    void *fwx[2] = {nullptr, nullptr};  // second pair for the checkpointed forward recomputation

    ncclComm_t comm = nullptr;
    int nranks = 1;

    std::string err;

    int fail(int code, const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        err = vformat(fmt, ap);
        va_end(ap);
        return code;
    }
};

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? FWI_ENOMEM : FWI_EHIP, "%s: %s",   \
                               #call, hipGetErrorString(e_));                                 \
    } while (0)

#define NCCLCHK(ctx, call)                                                                    \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) return (ctx)->fail(FWI_ECOMM, "%s: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

namespace {

void profile(std::vector<double> &p, int n, int npml, double sigma_max, double dt) {
    p.assign(n, 0.0);
    if (npml <= 0) return;
    for (int i = 0; i < n; ++i) {
        double dist = std::max(0.0, std::max((double)npml - i, (double)i - (n - 1 - npml)));
        p[i] = 0.5 * dt * sigma_max * (dist / npml) * (dist / npml);
    }
}

template <typename T>
int upload_vec(fwi_ctx *ctx, void *dst, const std::vector<double> &v) {
    std::vector<T> t(v.begin(), v.end());
    HIPCHK(ctx, hipMemcpyAsync(dst, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

// Time series cross the PCIe boundary through a persistent pinned staging buffer: the caller's
// NumPy arrays are pageable and usually freshly mapped, and the driver's pageable path cost up to
// tens of ms per 2 MB copy (measured in the 2-D shot loop).
int stage_reserve(fwi_ctx *ctx, size_t bytes) {
    if (ctx->cap_pin >= bytes && ctx->pin) return FWI_OK;
    if (ctx->pin) {
        // an H2D copy out of the buffer may still be queued on the stream (forward: the wavelet upload)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipHostFree(ctx->pin));
    }
    ctx->pin = nullptr;
    HIPCHK(ctx, hipHostMalloc(&ctx->pin, bytes ? bytes : 16, hipHostMallocDefault));
    ctx->cap_pin = bytes;
    return FWI_OK;
}

int upload_series(fwi_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return FWI_OK;
    int rc = stage_reserve(ctx, bytes);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer may still be in flight
    memcpy(ctx->pin, src, bytes);
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    return FWI_OK;
}

// device -> caller; synchronises the stream
int download_series(fwi_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes) {
        int rc = stage_reserve(ctx, bytes);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes) memcpy(dst, ctx->pin, bytes);
    return FWI_OK;
}

int ensure(fwi_ctx *ctx, void **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes && *p) return FWI_OK;
    if (*p) HIPCHK(ctx, hipFree(*p));
    *p = nullptr;
    HIPCHK(ctx, hipMalloc(p, bytes ? bytes : 16));
    *cap = bytes;
    return FWI_OK;
}

// The time loop of one sweep, either submitted launch by launch or captured into a hipGraph and launched once
// (fwi_config.launch_mode).  begin() stands where the loop's first launch is about to be formed, end() behind its
// last one; between them only kernel launches and device-to-device copies on ctx->stream may happen (true of
// run_steps / run_fused / run_pairs, the checkpoint copies and the record / imaging tails).  A TimeLoop that goes out
// of scope inside a capture (an error return) ends the capture and drops the partial graph.
struct TimeLoop {
    fwi_ctx *ctx;
    bool capturing = false;
    std::chrono::steady_clock::time_point t0;
    explicit TimeLoop(fwi_ctx *c) : ctx(c) {}
    int begin() {
        t0 = std::chrono::steady_clock::now();
        ctx->host_graph_ms = 0.0;
        if (ctx->graph_exec) {  // the previous sweep's graph: finished by now or not, the stream orders us behind it
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipGraphExecDestroy(ctx->graph_exec);
            ctx->graph_exec = nullptr;
        }
        if (ctx->graph) {
            (void)hipGraphDestroy(ctx->graph);
            ctx->graph = nullptr;
        }
        if (ctx->graph_mode) {
            HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            capturing = true;
        } else {
            HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        }
        return FWI_OK;
    }
    int end() {
        if (capturing) {
            capturing = false;
            HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &ctx->graph));
            HIPCHK(ctx, hipGraphInstantiate(&ctx->graph_exec, ctx->graph, nullptr, nullptr, 0));
            ctx->host_graph_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            HIPCHK(ctx, hipGraphLaunch(ctx->graph_exec, ctx->stream));
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        ctx->host_submit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->have_loop_time = true;
        return FWI_OK;
    }
    ~TimeLoop() {
        if (capturing) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(ctx->stream, &g);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
        }
    }
};

// Translate (n, ndim) int32 grid indices into padded / compact flat indices.
int flatten(fwi_ctx *ctx, const int32_t *idx, int n, std::vector<int64_t> &pidx,
            std::vector<int64_t> &cidx) {
    const GridDesc &g = ctx->gd;
    pidx.resize(n);
    cidx.resize(n);
    for (int i = 0; i < n; ++i) {
        const int32_t *t = idx + (size_t)i * g.ndim;
        const int z = t[0], y = (g.ndim == 3) ? t[1] : 0, x = t[g.ndim - 1];
        if (z < 0 || z >= g.nz || y < 0 || y >= g.ny || x < 0 || x >= g.nx)
            return ctx->fail(FWI_EINVAL, "grid index %d outside the grid", i);
        pidx[i] = g.off0 + (int64_t)z * g.sz + (int64_t)y * g.sy + x;
        cidx[i] = ((int64_t)z * g.ny + y) * g.cx + x;
    }
    return FWI_OK;
}

template <typename T>
struct Impl {
    // State of one field pair being stepped: which buffer holds the newest field, and the step
    // whose sampling is still owed (it rides on the next launch).
    struct Sweep {
        void *f[2];
        int cur = 0;
        int prev_n = -1;
        void *v = nullptr;  // increment form: the v field of this sweep
        bool pml_fw = false;  // CPML: this sweep is the checkpointed forward recomputation (its own memory variables)
    };

    static StepArgs<T> base_args(fwi_ctx *ctx, int cur) {
        StepArgs<T> a;
        a.u_cur = (const T *)ctx->u[cur];
        a.u_prev = (T *)ctx->u[cur ^ 1];
        a.v = nullptr;
        a.C = (const T *)ctx->C;
        a.dz = (const T *)ctx->dz;
        a.dy = (const T *)ctx->dy;
        a.dx = (const T *)ctx->dx;
        a.q_out = nullptr;
        a.q_in = nullptr;
        a.q_in2 = nullptr;
        a.q_bf16 = ctx->qbf16 ? 1 : 0;
        a.g = (T *)ctx->g_acc;
        // Star weights NORMALISED by a_1 (the factor a_1 / h^2 lives in the padded C, see finish_model): in fp32 the
        // rounded a_k / h^2 are a systematic perturbation of the operator -- a phase error growing with omega T that
        // was the largest term of the fp32 seismogram error (2-D layered 256^2 x 1000 steps in emulation: 3.4e-6 ->
        // 1.8e-6 standard form, 2.9e-6 -> 7.7e-7 increment form).  Normalised, a_1 -> 1 and a_2 -> -1/8 (O(8)) or
        // -1/16 (O(4)) are exact and the inexact a_3, a_4 weigh 0.016 and 0.0011: their rounding no longer matters.
        const double *co = ctx->cfg.order == 2 ? COEF2 : ctx->cfg.order == 4 ? COEF4 : COEF8;
        for (int k = 0; k < 5; ++k) a.ck[k] = T(0);
        for (int k = 1; k <= ctx->gd.r; ++k) a.ck[k] = (T)(co[k] / co[1]);
        a.damp = ctx->cfg.npml > 0 && !ctx->cpml;
        a.npml = ctx->cfg.npml;
        a.dz_scale = a.damp ? (T)(0.5 * ctx->cfg.dt * ctx->cfg.sigma_max /
                                  ((double)ctx->cfg.npml * ctx->cfg.npml)) : T(0);
        a.pml_tz = a.pml_ty = nullptr;
        a.xp_mode = 0;
        a.xp_partial = 0;
        a.xp_psi = a.xp_zeta = nullptr;
        a.xp_a = a.xp_b = nullptr;
        for (int k = 0; k < 5; ++k) a.xp_dk[k] = a.xp_dk1[k] = T(0);
        a.inj_start = nullptr;
        a.inj_pidx = a.inj_cidx = nullptr;
        a.inj_cu = a.inj_cq = a.inj_amp = nullptr;
        a.inj_col = nullptr;
        a.inj_run = nullptr;
        a.rec_pidx = nullptr;
        a.rec_out = nullptr;
        a.rec_scale = T(0);
        a.nrec = 0;
        return a;
    }

    // caller's per-point (or per-node) series -> the per-node device array `dst` (nt x nnodes)
    static int upload_amplitudes(fwi_ctx *ctx, void *dst, const void *host, int nt, int nnodes,
                                 const fwi_ctx::SpreadSet &sp) {
        if (sp.npts == 0) return upload_series(ctx, dst, host, (size_t)nt * nnodes * sizeof(T));
        int rc = ensure(ctx, &ctx->pts_a, &ctx->cap_pts_a, (size_t)nt * sp.npts * sizeof(T));
        if (rc) return rc;
        if ((rc = upload_series(ctx, ctx->pts_a, host, (size_t)nt * sp.npts * sizeof(T)))) return rc;
        HIPCHK(ctx, launch_scatter_series<T>((const T *)ctx->pts_a, (T *)dst, (const int *)sp.owner, (const T *)sp.weight,
                                             nt, sp.npts, nnodes, ctx->stream));
        return FWI_OK;
    }

    // the per-node device series `src` (nt x nnodes) -> the caller, per node or gathered per point; `keep` (or
    // nullptr) receives the device copy of what was handed out
    static int download_samples(fwi_ctx *ctx, void *host, const void *src, int nt, int nnodes,
                                const fwi_ctx::SpreadSet &sp, void **keep, size_t *keep_cap) {
        if (sp.npts == 0) return download_series(ctx, host, src, host ? (size_t)nt * nnodes * sizeof(T) : 0);
        void **buf = keep ? keep : &ctx->pts_a;
        size_t *cap = keep ? keep_cap : &ctx->cap_pts_a;
        int rc = ensure(ctx, buf, cap, (size_t)nt * sp.npts * sizeof(T));
        if (rc) return rc;
        HIPCHK(ctx, launch_gather_series<T>((const T *)src, (T *)*buf, (const int *)sp.pt_start, (const T *)sp.weight, nt,
                                            sp.npts, nnodes, ctx->stream));
        return download_series(ctx, host, *buf, host ? (size_t)nt * sp.npts * sizeof(T) : 0);
    }

    static PmlArgs<T> pml_args(fwi_ctx *ctx, const Sweep &sw, T *q_out) {
        PmlArgs<T> p;
        p.u_cur = (const T *)sw.f[sw.cur];
        p.u_next = (T *)sw.f[sw.cur ^ 1];
        p.v = (T *)sw.v;
        p.C = (const T *)ctx->C;
        p.q_out = q_out;
        for (int d = 0; d < 3; ++d) {
            p.psi[d] = (T *)(sw.pml_fw ? ctx->pml_psi_fw[d] : ctx->pml_psi[d]);
            p.zeta[d] = (T *)(sw.pml_fw ? ctx->pml_zeta_fw[d] : ctx->pml_zeta[d]);
            p.a[d] = (const T *)ctx->pml_a[d];
            p.b[d] = (const T *)ctx->pml_b[d];
        }
        const double *co = ctx->cfg.order == 2 ? COEF2 : ctx->cfg.order == 4 ? COEF4 : COEF8;
        const double *dc = ctx->cfg.order == 2 ? DCOEF2 : ctx->cfg.order == 4 ? DCOEF4 : DCOEF8;
        // The padded C carries a_1 / h^2 (base_args), so the border term is formed times s = h^2 / a_1: the second
        // differences take the normalised weights, the first differences that act on the FIELD (inside the psi / pt
        // recursions) take dk1 = s d_k / h, those that act on a memory variable the plain d_k / h.  The memory
        // variables are thereby carried times s, which nothing outside these kernels sees.
        const double h = ctx->cfg.h;
        for (int k = 0; k < 5; ++k) p.ck[k] = p.dk[k] = p.dk1[k] = T(0);
        for (int k = 0; k <= ctx->gd.r; ++k) p.ck[k] = (T)(co[k] / co[1]);
        for (int k = 1; k <= ctx->gd.r; ++k) {
            p.dk[k] = (T)(dc[k - 1] / h);
            p.dk1[k] = (T)(dc[k - 1] * h / co[1]);
        }
        p.npml = ctx->cfg.npml;
        p.tz = (T *)ctx->pml_tz;
        p.ty = (T *)ctx->pml_ty;
        return p;
    }

    // the memory variables <-> snapshot `seg` (forward set when `fw`, else the main set: the plain forward sweep)
    static int pml_snapshot(fwi_ctx *ctx, int seg, bool fw, bool restore) {
        char *base = (char *)ctx->pml_snap + (size_t)seg * ctx->pml_snap_stride;
        size_t off = 0;
        for (int d = 0; d < 3; ++d) {
            if (!ctx->pml_psi[d]) continue;
            for (void *arr : {fw ? ctx->pml_psi_fw[d] : ctx->pml_psi[d], fw ? ctx->pml_zeta_fw[d] : ctx->pml_zeta[d]}) {
                if (restore)
                    HIPCHK(ctx, hipMemcpyAsync(arr, base + off, ctx->pml_bytes[d], hipMemcpyDeviceToDevice, ctx->stream));
                else
                    HIPCHK(ctx, hipMemcpyAsync(base + off, arr, ctx->pml_bytes[d], hipMemcpyDeviceToDevice, ctx->stream));
                off += ctx->pml_bytes[d];
            }
        }
        return FWI_OK;
    }

    static int pml_zero(fwi_ctx *ctx) {
        for (int d = 0; d < 3; ++d)
            if (ctx->pml_psi[d]) {
                HIPCHK(ctx, hipMemsetAsync(ctx->pml_psi[d], 0, ctx->pml_bytes[d], ctx->stream));
                HIPCHK(ctx, hipMemsetAsync(ctx->pml_zeta[d], 0, ctx->pml_bytes[d], ctx->stream));
            }
        return FWI_OK;
    }

    // Host arrays are model-shaped (row stride nx); compact device arrays have row stride cx (nx
    // rounded up to 4, pad columns zero).  Equal for nx % 4 == 0: one plain copy.  Otherwise the
    // contiguous array goes through a device staging buffer and a repack kernel.
    static int upload_compact(fwi_ctx *ctx, void *dst, const void *host) {
        const GridDesc &g = ctx->gd;
        if (g.cx == g.nx) {
            HIPCHK(ctx, hipMemcpyAsync(dst, host, (size_t)g.npts * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            return FWI_OK;
        }
        const size_t nlog = (size_t)g.nz * g.ny * g.nx;
        if (!ctx->logical) HIPCHK(ctx, hipMalloc(&ctx->logical, nlog * sizeof(T)));
        HIPCHK(ctx, hipMemcpyAsync(ctx->logical, host, nlog * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, launch_repack<T>(g, (T *)dst, (const T *)ctx->logical, 1, ctx->stream));
        return FWI_OK;
    }

    static int download_compact(fwi_ctx *ctx, void *host, const void *src) {
        const GridDesc &g = ctx->gd;
        if (g.cx == g.nx) {
            HIPCHK(ctx, hipMemcpyAsync(host, src, (size_t)g.npts * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            return FWI_OK;
        }
        const size_t nlog = (size_t)g.nz * g.ny * g.nx;
        if (!ctx->logical) HIPCHK(ctx, hipMalloc(&ctx->logical, nlog * sizeof(T)));
        HIPCHK(ctx, launch_repack<T>(g, (T *)ctx->logical, (const T *)src, 0, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(host, ctx->logical, nlog * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        return FWI_OK;
    }

    static int set_model(fwi_ctx *ctx, const T *c) {
        // compact H2D copy, then validation + padded C = dt^2 c^2 on the device (the halo of C
        // was zeroed at creation and is never written)
        int rc = upload_compact(ctx, ctx->c_dev, c);
        if (rc) return rc;
        return finish_model(ctx);
    }

    static int set_model_vec(fwi_ctx *ctx, const void *dev) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->c_dev, dev, (size_t)ctx->gd.npts * sizeof(T), hipMemcpyDeviceToDevice,
                                   ctx->stream));
        return finish_model(ctx);
    }

    static int finish_model(fwi_ctx *ctx) {
        const GridDesc &g = ctx->gd;
        hipStream_t s = ctx->stream;
        int *bad = (int *)(ctx->red + 4);
        HIPCHK(ctx, hipMemsetAsync(bad, 0, sizeof(int), s));
        // padded C = dt^2 c^2 a_1 / h^2: the kernels' star weights are normalised by a_1 (base_args)
        const double *co = ctx->cfg.order == 2 ? COEF2 : ctx->cfg.order == 4 ? COEF4 : COEF8;
        HIPCHK(ctx, launch_build_model<T>(g, (const T *)ctx->c_dev, (T *)ctx->C,
                                          ctx->cfg.dt * ctx->cfg.dt * co[1] / (ctx->cfg.h * ctx->cfg.h), bad, s));
        int nbad = 0;
        HIPCHK(ctx, hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        ctx->have_forward = false;
        if (nbad) {
            ctx->have_model = false;
            return ctx->fail(FWI_EINVAL, "velocity must be finite and > 0 (%d bad entries)", nbad);
        }
        ctx->have_model = true;
        return FWI_OK;
    }

    // coefficient of an injected amplitude in u (A*C*scale) and in q (C*scale)
    static void point_coefs(fwi_ctx *ctx, const int32_t *idx, int n, double scale, const std::vector<T> &cpt,
                            std::vector<T> &cu, std::vector<T> &cq) {
        const GridDesc &g = ctx->gd;
        const double dt2 = ctx->cfg.dt * ctx->cfg.dt;
        cu.resize(n);
        cq.resize(n);
        for (int i = 0; i < n; ++i) {
            const int32_t *t = idx + (size_t)i * g.ndim;
            const int z = t[0], y = (g.ndim == 3) ? t[1] : 0, x = t[g.ndim - 1];
            const double cv = (double)cpt[i];
            const double d = ctx->pz[z] + ((g.ndim == 3) ? ctx->py[y] : 0.0) + ctx->px[x];
            const double Cv = dt2 * cv * cv;  // dt^2 c^2 itself (the padded C of the kernels carries a_1 / h^2 as well)
            cq[i] = (T)(Cv * scale);
            cu[i] = (T)(Cv * scale / (1.0 + d));
        }
    }

    static int upload_set(fwi_ctx *ctx, fwi_ctx::PointSet &ps, int32_t n, const int32_t *idx,
                          double scale) {
        const GridDesc &g = ctx->gd;
        std::vector<int64_t> p, c;
        int rcode;
        if ((rcode = flatten(ctx, idx, n, p, c))) return rcode;
        const size_t need = (size_t)n * 8 + 16;
        if (ps.cap < need) {
            for (void **q : {&ps.pidx, &ps.cidx, &ps.cu, &ps.cq, &ps.s_pidx, &ps.s_cidx, &ps.s_cu, &ps.s_cq,
                             &ps.s_col, &ps.s_run}) {
                if (*q) HIPCHK(ctx, hipFree(*q));
                *q = nullptr;
                HIPCHK(ctx, hipMalloc(q, need));
            }
            ps.cap = need;
        }
        auto up = [&](void *d, const void *h, size_t b) {
            return b ? hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
        };
        HIPCHK(ctx, up(ps.pidx, p.data(), (size_t)n * 8));
        HIPCHK(ctx, up(ps.cidx, c.data(), (size_t)n * 8));
        // velocities at the points: gathered from the device copy of the model
        std::vector<T> cpt(n), cu, cq;
        if (n) {
            HIPCHK(ctx, launch_record<T>((const T *)ctx->c_dev, (const int64_t *)ps.cidx, (T *)ps.cu, T(1), n,
                                         ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(cpt.data(), ps.cu, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        point_coefs(ctx, idx, n, scale, cpt, cu, cq);
        HIPCHK(ctx, up(ps.cu, cu.data(), (size_t)n * sizeof(T)));
        HIPCHK(ctx, up(ps.cq, cq.data(), (size_t)n * sizeof(T)));
        std::vector<int> start, col;
        std::vector<int64_t> sp, sc;
        std::vector<T> scu, scq;
        {  // counting sort of the entries by the step kernel's workgroup tile
            const bool st = ctx->kernel == K_STREAM;
            const int ntile = st ? stream_num_tiles(g, ctx->tune) : point_num_tiles(g);
            start.assign(ntile + 1, 0);
            std::vector<int> tile(n);
            for (int i = 0; i < n; ++i) {
                const int32_t *t = idx + (size_t)i * g.ndim;
                const int z = t[0], y = (g.ndim == 3) ? t[1] : 0, x = t[g.ndim - 1];
                tile[i] = st ? stream_tile_of(g, ctx->tune, z, y, x) : point_tile_of(g, z, y, x);
                ++start[tile[i] + 1];
            }
            for (int k = 0; k < ntile; ++k) start[k + 1] += start[k];
            // within a tile by node, then by entry: the entries of one node are consecutive (the kernels add such a
            // run from one thread in this order, inject_runs)
            std::vector<int> ord(n);
            for (int i = 0; i < n; ++i) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                return tile[x] != tile[y] ? tile[x] < tile[y] : p[x] < p[y];
            });
            col.resize(n); sp.resize(n); sc.resize(n); scu.resize(n); scq.resize(n);
            std::vector<int> run(n, 0);
            for (int k = 0; k < n; ++k) {
                const int i = ord[k];
                col[k] = i; sp[k] = p[i]; sc[k] = c[i]; scu[k] = cu[i]; scq[k] = cq[i];
            }
            for (int k = n - 1, len = 0; k >= 0; --k) {  // run length at the first entry of each node's run
                len = (k + 1 < n && sp[k + 1] == sp[k] && tile[ord[k + 1]] == tile[ord[k]]) ? len + 1 : 1;
                run[k] = (k == 0 || sp[k - 1] != sp[k] || tile[ord[k - 1]] != tile[ord[k]]) ? len : 0;
            }
            const size_t sb = start.size() * sizeof(int);
            if (ps.cap_start < sb) {
                if (ps.s_start) HIPCHK(ctx, hipFree(ps.s_start));
                ps.s_start = nullptr;
                HIPCHK(ctx, hipMalloc(&ps.s_start, sb));
                ps.cap_start = sb;
            }
            HIPCHK(ctx, up(ps.s_start, start.data(), sb));
            HIPCHK(ctx, up(ps.s_pidx, sp.data(), (size_t)n * 8));
            HIPCHK(ctx, up(ps.s_cidx, sc.data(), (size_t)n * 8));
            HIPCHK(ctx, up(ps.s_cu, scu.data(), (size_t)n * sizeof(T)));
            HIPCHK(ctx, up(ps.s_cq, scq.data(), (size_t)n * sizeof(T)));
            HIPCHK(ctx, up(ps.s_col, col.data(), (size_t)n * sizeof(int)));
            HIPCHK(ctx, up(ps.s_run, run.data(), (size_t)n * sizeof(int)));
        }
        std::vector<int> fst, flz, flx, fcol, rst, rlz, rlx, rcol;
        std::vector<unsigned char> fint;
        std::vector<int64_t> fcidx;
        std::vector<T> fcu, fcq;
        if (ctx->fused2d) {
            // entries of the fused 2-D kernel: a point is injected by every tile whose EXTENDED region
            // holds it (each keeps a private copy of the halo) and sampled by the one tile owning it
            const int FT = ctx->fused_ft;
            const int HL = (FUSED2D_STEPS * g.r + 3) / 4 * 4;  // as in step2d_fused
            const int ntx = (g.nx + FT - 1) / FT, ntz = (g.nz + FT - 1) / FT, ntile = ntx * ntz;
            struct Ent { int tile, lz, lx, col; unsigned char interior; };
            std::vector<Ent> ents;
            // (with the CPML inside the launch the tiles are whole, with one overlap seam per axis: fused2d_origin /
            // fused2d_own give every tile's first cell and the first cell it owns)
            const int seam = ctx->cpml ? 1 : 0;
            struct Ax { int t, l; bool own; };
            auto tiles_of = [&](int c, int nn, int nt, Ax *out) {  // tiles whose extended region holds coordinate c
                int m = 0;
                // HL <= FT and the seam shifts a tile by less than FT: the owning tile's index is within 2 of c / FT
                for (int t = std::max(0, c / FT - 1); t <= std::min(nt - 1, c / FT + 2); ++t) {
                    const int o = fused2d_origin(t, nn, FT, seam);
                    if (c < o - HL || c >= o + FT + HL) continue;
                    const int lo = fused2d_own(t, nn, FT, seam), hi = t + 1 < nt ? fused2d_own(t + 1, nn, FT, seam) : nn;
                    out[m++] = Ax{t, c - (o - HL), c >= lo && c < hi};
                }
                return m;
            };
            for (int i = 0; i < n; ++i) {
                const int z = idx[(size_t)i * g.ndim], x = idx[(size_t)i * g.ndim + 1];
                Ax az[4], ax[4];
                const int mz = tiles_of(z, g.nz, ntz, az), mx = tiles_of(x, g.nx, ntx, ax);
                for (int p = 0; p < mz; ++p)
                    for (int q = 0; q < mx; ++q)
                        ents.push_back({az[p].t * ntx + ax[q].t, az[p].l, ax[q].l, i, (unsigned char)(az[p].own && ax[q].own)});
            }
            std::stable_sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) {
                if (x.tile != y.tile) return x.tile < y.tile;
                return x.lz != y.lz ? x.lz < y.lz : x.lx < y.lx;  // a node's entries consecutive, in entry order
            });
            auto build = [&](bool interior_only, std::vector<int> &st, std::vector<int> &lz, std::vector<int> &lx,
                             std::vector<int> &col, std::vector<int> *order) {
                st.assign(ntile + 1, 0);
                for (const Ent &e : ents)
                    if (!interior_only || e.interior) ++st[e.tile + 1];
                for (int k = 0; k < ntile; ++k) st[k + 1] += st[k];
                std::vector<int> fill(st.begin(), st.end() - 1);
                const int tot = st[ntile];
                lz.resize(tot); lx.resize(tot); col.resize(tot);
                if (order) order->resize(tot);
                for (size_t q = 0; q < ents.size(); ++q) {
                    const Ent &e = ents[q];
                    if (interior_only && !e.interior) continue;
                    const int k = fill[e.tile]++;
                    lz[k] = e.lz; lx[k] = e.lx; col[k] = e.col;
                    if (order) (*order)[k] = (int)q;
                }
            };
            std::vector<int> order;
            build(false, fst, flz, flx, fcol, &order);
            build(true, rst, rlz, rlx, rcol, nullptr);
            const size_t ne = order.size();
            fint.resize(ne); fcidx.resize(ne); fcu.resize(ne); fcq.resize(ne);
            for (size_t k = 0; k < ne; ++k) {
                const Ent &e = ents[order[k]];
                fint[k] = e.interior; fcidx[k] = c[e.col]; fcu[k] = cu[e.col]; fcq[k] = cq[e.col];
            }
            const size_t fneed = ne * 8 + 16, sneed = (size_t)(ntile + 1) * sizeof(int);
            if (ps.fcap < fneed) {
                for (void **q : {&ps.fi_lz, &ps.fi_lx, &ps.fi_col, &ps.fi_int, &ps.fi_cidx, &ps.fi_cu, &ps.fi_cq,
                                 &ps.fi_run, &ps.fr_lz, &ps.fr_lx, &ps.fr_col}) {
                    if (*q) HIPCHK(ctx, hipFree(*q));
                    *q = nullptr;
                    HIPCHK(ctx, hipMalloc(q, fneed));
                }
                ps.fcap = fneed;
            }
            if (ps.fcap_start < sneed) {
                for (void **q : {&ps.fi_start, &ps.fr_start}) {
                    if (*q) HIPCHK(ctx, hipFree(*q));
                    *q = nullptr;
                    HIPCHK(ctx, hipMalloc(q, sneed));
                }
                ps.fcap_start = sneed;
            }
            HIPCHK(ctx, up(ps.fi_start, fst.data(), sneed));
            HIPCHK(ctx, up(ps.fr_start, rst.data(), sneed));
            HIPCHK(ctx, up(ps.fi_lz, flz.data(), ne * sizeof(int)));
            HIPCHK(ctx, up(ps.fi_lx, flx.data(), ne * sizeof(int)));
            HIPCHK(ctx, up(ps.fi_col, fcol.data(), ne * sizeof(int)));
            HIPCHK(ctx, up(ps.fi_int, fint.data(), ne));
            HIPCHK(ctx, up(ps.fi_cidx, fcidx.data(), ne * 8));
            HIPCHK(ctx, up(ps.fi_cu, fcu.data(), ne * sizeof(T)));
            HIPCHK(ctx, up(ps.fi_cq, fcq.data(), ne * sizeof(T)));
            {
                std::vector<int> frun(ne, 0);  // run length at the first entry of each (tile, node) run
                for (int k = (int)ne - 1, len = 0; k >= 0; --k) {
                    auto same = [&](int x, int y) {
                        const Ent &ex = ents[order[x]], &ey = ents[order[y]];
                        return ex.tile == ey.tile && ex.lz == ey.lz && ex.lx == ey.lx;
                    };
                    len = (k + 1 < (int)ne && same(k, k + 1)) ? len + 1 : 1;
                    frun[k] = (k == 0 || !same(k - 1, k)) ? len : 0;
                }
                HIPCHK(ctx, up(ps.fi_run, frun.data(), ne * sizeof(int)));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // frun goes out of scope
            }
            HIPCHK(ctx, up(ps.fr_lz, rlz.data(), rlz.size() * sizeof(int)));
            HIPCHK(ctx, up(ps.fr_lx, rlx.data(), rlx.size() * sizeof(int)));
            HIPCHK(ctx, up(ps.fr_col, rcol.data(), rcol.size() * sizeof(int)));
        }
        std::vector<int> pst;
        std::vector<Pair3dInj> pent;
        if (ctx->pair3d) {
            if constexpr (std::is_same<T, float>::value) {
                // a point is injected by every workgroup whose step-1 region (tile widened by r) holds it
                const int zc = ctx->pair_zc, tw = ctx->pair_tw, r = g.r, TYI = PAIR3D_TY;
                const int nxt = g.nx <= 256 ? 1 : (g.nx + tw - 1) / tw;
                const int nyt = (g.ny + TYI - 1) / TYI, nzc = (g.nz + zc - 1) / zc;
                const int ntile = nxt * nyt * nzc;
                std::vector<std::vector<Pair3dInj>> per(ntile);
                for (int i = 0; i < n; ++i) {
                    const int z = idx[(size_t)i * 3], y = idx[(size_t)i * 3 + 1], x = idx[(size_t)i * 3 + 2];
                    for (int bz = std::max(0, (z - r) / zc - 1); bz < nzc; ++bz) {
                        const int z0 = bz * zc, z1 = std::min(g.nz, z0 + zc);
                        if (z < z0 - r || z >= z1 + r) continue;
                        for (int by = std::max(0, (y - r) / TYI - 1); by < nyt; ++by) {
                            const int y0 = by * TYI;
                            if (y < y0 - r || y >= y0 + TYI + r) continue;
                            for (int bx = 0; bx < nxt; ++bx) {
                                const int xi0 = bx * tw, xbase = nxt == 1 ? 0 : xi0 - 2 * HALO;
                                if (nxt > 1 && (x < xi0 - r || x >= xi0 + tw + r)) continue;
                                const bool own = z >= z0 && z < z1 && y >= y0 && y < y0 + TYI &&
                                                 (nxt == 1 || (x >= xi0 && x < xi0 + tw));
                                per[(bz * nyt + by) * nxt + bx].push_back(
                                    Pair3dInj{z, y - y0, x - xbase, i, (float)cu[i], own ? 1 : 0});
                            }
                        }
                    }
                }
                pst.assign(ntile + 1, 0);
                for (int k = 0; k < ntile; ++k) {
                    pst[k + 1] = pst[k] + (int)per[k].size();
                    pent.insert(pent.end(), per[k].begin(), per[k].end());
                }
                const size_t sb = pst.size() * sizeof(int), eb = pent.size() * sizeof(Pair3dInj) + 16;
                if (ps.pcap_start < sb) {
                    if (ps.pr_start) HIPCHK(ctx, hipFree(ps.pr_start));
                    ps.pr_start = nullptr;
                    HIPCHK(ctx, hipMalloc(&ps.pr_start, sb));
                    ps.pcap_start = sb;
                }
                if (ps.pcap < eb) {
                    if (ps.pr_ent) HIPCHK(ctx, hipFree(ps.pr_ent));
                    ps.pr_ent = nullptr;
                    HIPCHK(ctx, hipMalloc(&ps.pr_ent, eb));
                    ps.pcap = eb;
                }
                HIPCHK(ctx, up(ps.pr_start, pst.data(), sb));
                HIPCHK(ctx, up(ps.pr_ent, pent.data(), pent.size() * sizeof(Pair3dInj)));
            }
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
        ps.n = n;
        return FWI_OK;
    }

    // 3-D temporal blocking: `count` (even) forward steps from n0, two per pass, ping-ponging between the buffer
    // pair in `sw` and `spare`.  Both new fields are sampled by the next pass; the last pair by the caller's
    // flush_record (newest field) and one explicit record here (the older one).
    static int run_pairs(fwi_ctx *ctx, Sweep &sw, void **spare, int n0, int count, const fwi_ctx::PointSet &inj,
                         const T *amp, const fwi_ctx::PointSet *out, T *series, T out_scale) {
        if constexpr (std::is_same<T, float>::value) {
            const GridDesc &g = ctx->gd;
            const StepArgs<T> b = base_args(ctx, 0);
            for (int done = 0; done < count; done += 2) {
                const int n = n0 + done;
                Pair3dArgs a{};
                a.u_cur = (const float *)sw.f[sw.cur];
                a.u_prev = (const float *)sw.f[sw.cur ^ 1];
                a.C = (const float *)ctx->C;
                a.out2 = (float *)spare[0];
                a.out1 = (float *)spare[1];
                a.dy = (const float *)ctx->dy;
                a.dx = (const float *)ctx->dx;
                for (int k = 0; k < 5; ++k) a.ck[k] = b.ck[k];
                a.damp = b.damp;
                a.npml = b.npml;
                a.dz_scale = b.dz_scale;
                if (inj.n > 0) {
                    a.inj_start = (const int *)inj.pr_start;
                    a.inj = (const Pair3dInj *)inj.pr_ent;
                    a.inj_amp0 = amp + (size_t)n * inj.n;
                    a.inj_amp1 = amp + (size_t)(n + 1) * inj.n;
                }
                if (out && out->n > 0 && done > 0) {  // the previous pass' fields: steps n - 2 and n - 1
                    a.rec_pidx = (const int64_t *)out->pidx;
                    a.rec_out0 = series + (size_t)(n - 2) * out->n;
                    a.rec_out1 = series + (size_t)(n - 1) * out->n;
                    a.rec_scale = out_scale;
                    a.nrec = out->n;
                }
                HIPCHK(ctx, launch_pair3d(g, a, ctx->pair_zc, ctx->pair_tw, ctx->stream));
                void *oc = sw.f[sw.cur], *op = sw.f[sw.cur ^ 1];
                sw.f[0] = spare[0];  // u^{n+2}
                sw.f[1] = spare[1];  // u^{n+1}
                sw.cur = 0;
                spare[0] = oc;
                spare[1] = op;
            }
            if (out && out->n > 0 && count > 0)
                HIPCHK(ctx, launch_record<T>((const T *)sw.f[1], (const int64_t *)out->pidx,
                                             series + (size_t)(n0 + count - 2) * out->n, out_scale, out->n, ctx->stream));
            sw.prev_n = count > 0 ? n0 + count - 1 : sw.prev_n;
            return FWI_OK;
        } else {
            (void)sw; (void)spare; (void)n0; (void)count; (void)inj; (void)amp; (void)out; (void)series; (void)out_scale;
            return ctx->fail(FWI_ESTATE, "3-D two-step path is fp32 only");
        }
    }

    // 2-D temporal blocking: `count` steps (a multiple of FUSED2D_STEPS) starting at n0 in direction
    // dn, FUSED2D_STEPS per launch, ping-ponging between the buffer pair in `sw` and ctx->fx.
    // Sampling is not lagged here (it happens inside the sub-steps): sw.prev_n stays -1.
    static int run_fused(fwi_ctx *ctx, Sweep &sw, void **spare, int n0, int dn, int count,
                         const fwi_ctx::PointSet &inj, const T *amp, const fwi_ctx::PointSet *out, T *series,
                         T out_scale, int mode, T *q_base) {
        if constexpr (std::is_same<T, float>::value) {
            const GridDesc &g = ctx->gd;
            const StepArgs<T> b = base_args(ctx, 0);
            for (int done = 0; done < count; done += FUSED2D_STEPS) {
                Fused2dArgs a{};
                a.u_cur = (const float *)sw.f[sw.cur];
                a.u_prev = (const float *)(ctx->inc ? sw.v : sw.f[sw.cur ^ 1]);  // increment form: v in / v out
                a.inc = ctx->inc ? 1 : 0;
                a.skipd = ctx->fused_skipd;
                a.ft = ctx->fused_ft;
                a.tile_order = ctx->fused_order;
                a.C = (const float *)ctx->C;
                a.out_cur = (float *)spare[0];
                a.out_prev = (float *)spare[1];
                a.dz = (const float *)ctx->dz;
                a.dx = (const float *)ctx->dx;
                for (int k = 0; k < 5; ++k) a.ck[k] = b.ck[k];
                a.damp = b.damp;
                a.mode = mode;
                a.q_base = (float *)q_base;
                a.istride = ctx->istride;
                a.g = (float *)ctx->g_acc;
                a.n0 = n0 + done * dn;
                a.dn = dn;
                if (inj.n > 0) {
                    a.inj_start = (const int *)inj.fi_start;
                    a.inj_lz = (const int *)inj.fi_lz;
                    a.inj_lx = (const int *)inj.fi_lx;
                    a.inj_col = (const int *)inj.fi_col;
                    a.inj_interior = (const unsigned char *)inj.fi_int;
                    a.inj_run = (const int *)inj.fi_run;
                    a.inj_cidx = (const int64_t *)inj.fi_cidx;
                    a.inj_cu = (const float *)inj.fi_cu;
                    a.inj_cq = (const float *)inj.fi_cq;
                    a.inj_amp = (const float *)amp;
                    a.ninj = inj.n;
                }
                if (out && out->n > 0) {
                    a.rec_start = (const int *)out->fr_start;
                    a.rec_lz = (const int *)out->fr_lz;
                    a.rec_lx = (const int *)out->fr_lx;
                    a.rec_col = (const int *)out->fr_col;
                    a.rec_out = (float *)series;
                    a.rec_scale = out_scale;
                    a.nrec = out->n;
                }
                if (ctx->cpml) {  // the border recursion inside the launch (fwi_fused2d_pml.hip)
                    const PmlArgs<T> p = pml_args(ctx, sw, nullptr);
                    a.pml_npml = p.npml;
                    a.pml_rev = dn < 0;
                    for (int d = 0; d < 2; ++d) {  // z -> 0, x -> 1
                        a.pml_psi[d] = (float *)p.psi[d ? 2 : 0];
                        a.pml_zeta[d] = (float *)p.zeta[d ? 2 : 0];
                        a.pml_psi_out[d] = (float *)ctx->pml_spare_psi[d ? 2 : 0];
                        a.pml_zeta_out[d] = (float *)ctx->pml_spare_zeta[d ? 2 : 0];
                        a.pml_a[d] = (const float *)p.a[d ? 2 : 0];
                        a.pml_b[d] = (const float *)p.b[d ? 2 : 0];
                    }
                    for (int k = 0; k < 5; ++k) {
                        a.pml_dk[k] = (float)p.dk[k];
                        a.pml_dk1[k] = (float)p.dk1[k];
                    }
                    HIPCHK(ctx, launch_fused2d_cpml(g, a, ctx->stream));
                    for (int d : {0, 2}) {  // the written set becomes the current one of this sweep
                        std::swap(sw.pml_fw ? ctx->pml_psi_fw[d] : ctx->pml_psi[d], ctx->pml_spare_psi[d]);
                        std::swap(sw.pml_fw ? ctx->pml_zeta_fw[d] : ctx->pml_zeta[d], ctx->pml_spare_zeta[d]);
                    }
                } else {
                    HIPCHK(ctx, launch_fused2d(g, a, ctx->stream));
                }
                if (ctx->inc) {
                    // (u, v) written into the spare pair; the old u and v buffers become the next launch's output.
                    // The second u buffer of the sweep (the single-step kernels' output) is not involved.
                    void *ou = sw.f[sw.cur], *ov = sw.v;
                    sw.f[sw.cur] = spare[0];
                    sw.v = spare[1];
                    spare[0] = ou;
                    spare[1] = ov;
                    continue;
                }
                // the freshly written pair becomes current; the old pair is the next launch's output
                void *oc = sw.f[sw.cur], *op = sw.f[sw.cur ^ 1];
                sw.f[0] = spare[0];
                sw.f[1] = spare[1];
                sw.cur = 0;
                spare[0] = oc;
                spare[1] = op;
            }
            sw.prev_n = -1;
            return FWI_OK;
        } else {
            (void)sw; (void)spare; (void)n0; (void)dn; (void)count; (void)inj; (void)amp; (void)out; (void)series;
            (void)out_scale; (void)mode; (void)q_base;
            return ctx->fail(FWI_ESTATE, "fused 2-D path is fp32 only");
        }
    }

    static bool use_fused(const fwi_ctx *ctx, int nt) {
        return ctx->fused2d && nt % FUSED2D_STEPS == 0 && (ctx->ckpt == 0 || ctx->ckpt % FUSED2D_STEPS == 0);
    }

    // Steps taken by the fused kernel when nt is not a multiple of FUSED2D_STEPS (store-all mode only): the
    // first nt - nt % FUSED2D_STEPS; the remaining 1-3 steps go one per launch.  0 = not applicable.
    static int mixed_fused_steps(const fwi_ctx *ctx, int nt) {
        if (!ctx->fused2d || ctx->ckpt != 0 || nt % FUSED2D_STEPS == 0) return 0;
        return nt - nt % FUSED2D_STEPS;
    }

    static int upload_points(fwi_ctx *ctx, int32_t nsrc, const int32_t *src_idx, int32_t nrec,
                             const int32_t *rec_idx) {
        int rc = upload_set(ctx, ctx->src, nsrc, src_idx, 1.0 / std::pow(ctx->cfg.h, ctx->gd.ndim));
        if (rc) return rc;
        return upload_set(ctx, ctx->rec, nrec, rec_idx, 1.0);
    }

    // Steps n = n0, n0 + dn, ... (count steps) of the shared time loop (oracle:
    // Propagator._propagate): inject `amp` rows at point set `inj`, sample the new field at point
    // set `out` (nullptr = none) into `series`; q_out(n) / q_in(n) give the per-step imaging-term
    // pointers (nullptr = off).  Injection and sampling are fused into the step launch: one launch
    // per time step; the sampling of step n rides on the next launch, which reads that field anyway.
    template <class QOut, class QIn>
    static int run_steps(fwi_ctx *ctx, Sweep &sw, int n0, int dn, int count, const fwi_ctx::PointSet &inj,
                         const T *amp, const fwi_ctx::PointSet *out, T *series, T out_scale, QOut q_out,
                         QIn q_in) {
        const GridDesc &g = ctx->gd;
        for (int it = 0, n = n0; it < count; ++it, n += dn) {
            StepArgs<T> a = base_args(ctx, 0);
            a.u_cur = (const T *)sw.f[sw.cur];
            a.u_prev = (T *)sw.f[sw.cur ^ 1];
            a.v = (T *)sw.v;
            a.q_out = q_out(n);
            q_in(n, a.q_in, a.q_in2);
            if (inj.n > 0) {
                a.inj_start = (const int *)inj.s_start;
                a.inj_pidx = (const int64_t *)inj.s_pidx;
                a.inj_cidx = (const int64_t *)inj.s_cidx;
                a.inj_cu = (const T *)inj.s_cu;
                a.inj_cq = (const T *)inj.s_cq;
                a.inj_col = (const int *)inj.s_col;
                a.inj_run = (const int *)inj.s_run;
                a.inj_amp = amp + (size_t)n * inj.n;
            }
            if (out && sw.prev_n >= 0 && out->n > 0) {
                a.rec_pidx = (const int64_t *)out->pidx;
                a.rec_out = series + (size_t)sw.prev_n * out->n;
                a.rec_scale = out_scale;
                a.nrec = out->n;
            }
            if (ctx->cpml) {  // memory variables of the border advance with the newest field ...
                const PmlArgs<T> p = pml_args(ctx, sw, a.q_out);
                int axes = 7;
                if (ctx->xpml) {  // ... the x border's inside the step kernel itself (step3d_stream, XP)
                    axes = 3;
                    a.xp_mode = dn < 0 ? 2 : 1;
                    a.xp_partial = ctx->cfg.npml % 4 != 0;
                    a.xp_psi = p.psi[2];
                    a.xp_zeta = p.zeta[2];
                    a.xp_a = p.a[2];
                    a.xp_b = p.b[2];
                    for (int k = 0; k < 5; ++k) {
                        a.xp_dk[k] = p.dk[k];
                        a.xp_dk1[k] = p.dk1[k];
                    }
                }
                // z / y borders: ONE line launch for both BEFORE the step (all three phases of the recursion; fwi_pml.hip,
                // pml_line_t) hands their term to the step kernel, which adds it inside q; the slab phases around the step
                // kernel take whatever is left (the x border of contexts whose lanes cannot carry it, grids without lines)
                const int lines = axes & ctx->pml_lines;
                axes &= ~lines;
                if (lines) {
                    HIPCHK(ctx, launch_pml_lines<T>(g, p, dn < 0, ctx->stream, lines));
                    a.pml_tz = p.tz;
                    a.pml_ty = p.ty;
                }
                if (axes) {
                    HIPCHK(ctx, launch_pml<T>(g, p, 1, dn < 0, ctx->stream, axes));
                    HIPCHK(ctx, launch_pml<T>(g, p, 2, dn < 0, ctx->stream, axes));
                }
                HIPCHK(ctx, launch_step<T>(ctx->kernel, g, a, ctx->tune, ctx->stream));
                if (axes) HIPCHK(ctx, launch_pml<T>(g, p, 3, dn < 0, ctx->stream, axes));  // their term joins u' (and q)
            } else {
                HIPCHK(ctx, launch_step<T>(ctx->kernel, g, a, ctx->tune, ctx->stream));
            }
            sw.prev_n = n;
            sw.cur ^= 1;
        }
        return FWI_OK;
    }

    // PLACEMENT SEARCH (3-D CPML contexts past the cache-resident sizes; fwi_create).  The step kernel of such a context
    // streams seven arrays side by side, and how fast it runs depends on where they lie RELATIVE to each other: with every
    // device array of a 256^3 / npml 16 context carved from one slab, moving the 10 MiB `ty` alone by 4 MiB switched the
    // step kernel between 48.7 and 57.9 us (63.5 with everything on 64 MiB boundaries) -- same bytes by the FETCH / WRITE
    // counters, the line launch unmoved -- and with one hipMalloc per array the draw differs per process and per box
    // (72.9 ... 85.8 us per time step, profiles/r04_variance.log, r04_placement_probes.log).  No congruence rule we tried
    // predicts the slow positions, so the context MEASURES: the four small arrays (x border's psi / zeta, tz, ty) are
    // allocated with `place_pad` bytes of slack, and a coordinate search over their offsets (steps of 2 MiB) times a few
    // time steps of the real launch sequence on zeroed fields for each candidate and keeps a position only if it is
    // > 1 % faster twice.  ~30 candidates x ~1.5 ms: some tens of milliseconds per context, bit-identical results (the
    // arithmetic never sees an address).  FWI_PLACEMENT_TUNE=0 is the A/B hook.
    static int tune_placement(fwi_ctx *ctx) {
        const GridDesc &g = ctx->gd;
        const size_t fb = (size_t)g.ptot * sizeof(T), step = (size_t)2 << 20;
        const int ncand = (int)(ctx->place_pad / step) + 1;
        if (ncand < 2 || ctx->nmov == 0) return FWI_OK;
        // everything the trial steps read is zero, so everything they write is: the context is left as created
        for (void *f : {ctx->u[0], ctx->u[1], ctx->C, ctx->vf})
            if (f) HIPCHK(ctx, hipMemsetAsync(f, 0, fb, ctx->stream));
        for (int i = 0; i < ctx->nmov; ++i)  // (the movable arrays: the whole allocation, whichever position wins)
            HIPCHK(ctx, hipMemsetAsync(*ctx->mov_slot[i], 0, ctx->mov_bytes[i] + ctx->place_pad, ctx->stream));
        int rc = pml_zero(ctx);
        if (rc) return rc;
        hipEvent_t e0, e1;
        HIPCHK(ctx, hipEventCreate(&e0));
        HIPCHK(ctx, hipEventCreate(&e1));
        const fwi_ctx::PointSet nobody;
        auto none = [](int) -> T * { return nullptr; };
        auto noq = [](int, const T *&p, const T *&p2) { p = p2 = nullptr; };
        Sweep sw;
        sw.f[0] = ctx->u[0];
        sw.f[1] = ctx->u[1];
        sw.v = ctx->vf;
        int nstep = 0;
        auto timed = [&](int n, float *us) -> int {  // n time steps of the real launch sequence, no sources, no receivers
            sw.v = ctx->vf;  // (a movable array itself in increment form)
            HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
            const int r = run_steps(ctx, sw, nstep, 1, n, nobody, (const T *)nullptr, nullptr, (T *)nullptr, T(0), none, noq);
            if (r) return r;
            nstep += n;
            HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
            HIPCHK(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
            *us = 1e3f * ms / n;
            return FWI_OK;
        };
        float best = 0.f, t = 0.f, t2 = 0.f;
        int n = 8;
        // warm up (code objects' first use, clocks), then the position the context was created with -- twice, the better
        if ((rc = timed(24, &t)) || (rc = timed(8, &t))) goto done;
        n = std::max(4, std::min(12, (int)(1500.f / std::max(t, 1.f))));  // ~1.5 ms per trial
        if (t > 2500.f) goto done;  // (a search would take seconds on such a grid, whose borders are a small share anyway)
        if ((rc = timed(n, &t)) || (rc = timed(n, &t2))) goto done;
        best = std::min(t, t2);
        ctx->place_us[0] = ctx->place_us[1] = best;
        for (int i = 0; i < ctx->nmov && !rc; ++i) {
            void **const slot = ctx->mov_slot[i];
            char *const base = (char *)*slot - ctx->mov_shift[i];
            size_t keep = ctx->mov_shift[i];
            for (int c = 0; c < ncand && !rc; ++c) {
                const size_t sh = (size_t)c * step;
                if (sh == keep) continue;
                *slot = base + sh;
                if ((rc = timed(n, &t)) || t >= 0.99f * best) continue;
                // a candidate that looks > 1 % faster: once more, then the incumbent again (the clock drifts over a
                // search), and only a candidate that beats both readings is kept
                if ((rc = timed(n, &t2))) continue;
                t = std::max(t, t2);
                *slot = base + keep;
                if ((rc = timed(n, &t2))) continue;
                if (t < 0.99f * t2) {
                    best = t;
                    keep = sh;
                } else {
                    best = t2;
                }
            }
            *slot = base + keep;  // (also on an error: the slot and its recorded shift always agree)
            ctx->mov_shift[i] = keep;
        }
        if (!rc) ctx->place_us[1] = best;
    done:
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    }

    // the last step's field has no following launch to ride on
    static int flush_record(fwi_ctx *ctx, Sweep &sw, const fwi_ctx::PointSet &out, T *series, T out_scale) {
        if (sw.prev_n < 0) return FWI_OK;
        HIPCHK(ctx, launch_record<T>((const T *)sw.f[sw.cur], (const int64_t *)out.pidx,
                                     series + (size_t)sw.prev_n * out.n, out_scale, out.n, ctx->stream));
        return FWI_OK;
    }

    static int zero_fields(fwi_ctx *ctx, void *a, void *b) {
        HIPCHK(ctx, hipMemsetAsync(a, 0, (size_t)ctx->gd.ptot * sizeof(T), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(b, 0, (size_t)ctx->gd.ptot * sizeof(T), ctx->stream));
        return FWI_OK;
    }

    static int forward(fwi_ctx *ctx, int32_t nt, int32_t nsrc, const int32_t *src_idx,
                       const T *wavelet, int32_t nrec, const int32_t *rec_idx, int32_t save,
                       T *seis_out) {
        const GridDesc &g = ctx->gd;
        const int K = ctx->ckpt;  // 0 = store every step's imaging term, K > 0 = snapshot every K steps
        int rc;
        if ((rc = upload_points(ctx, nsrc, src_idx, nrec, rec_idx))) return rc;
        if (save && K == 0 && !ctx->q_store) {
            const size_t slots = ((size_t)ctx->cfg.nt_max + ctx->istride - 1) / ctx->istride;
            const size_t bytes = slots * g.npts * ctx->qes;
            size_t fr = 0, tot = 0;
            HIPCHK(ctx, hipMemGetInfo(&fr, &tot));
            if (bytes > fr)
                return ctx->fail(FWI_ENOMEM,
                                 "forward-term store needs %.1f GiB (nt_max=%d) but only %.1f GiB are free; "
                                 "set ckpt_interval (recomputation) or image_stride (decimated imaging)",
                                 bytes / 1073741824.0, ctx->cfg.nt_max, fr / 1073741824.0);
            HIPCHK(ctx, hipMalloc(&ctx->q_store, bytes));
        }
        if (save && K > 0 && !ctx->ckpt_ready) {
            // snapshots of (u^n, u^{n-1}) per segment, K + 1 imaging-term slots, the recomputation's field
            // pair(s).  All or nothing: a partial set would let a later adjoint run on null buffers.
            const int nseg = (ctx->cfg.nt_max + K - 1) / K;
            const size_t fb = (size_t)g.ptot * sizeof(T);
            struct { void **p; size_t bytes; bool zero; } want[] = {
                {&ctx->snap, (size_t)nseg * 2 * fb, false},
                {&ctx->q_store, (size_t)(K + 1) * g.npts * sizeof(T), false},  // K slots + carry
                // (padded fields are zeroed at allocation: in increment form fwd[1] only ever receives interior points and its
                // halo must read 0 -- found by the option fuzz reusing device memory of earlier contexts; the others are
                // overwritten whole before use, but fwi_check_padding looks at every padded field of the context)
                {&ctx->fwd[0], fb, true},
                {&ctx->fwd[1], fb, true}, {&ctx->fwv, ctx->inc ? fb : 0, true},
                {&ctx->fwx[0], ctx->fused2d ? fb : 0, true}, {&ctx->fwx[1], ctx->fused2d ? fb : 0, true},
                {&ctx->pml_psi_fw[0], ctx->pml_bytes[0], false}, {&ctx->pml_zeta_fw[0], ctx->pml_bytes[0], false},
                {&ctx->pml_psi_fw[1], ctx->pml_bytes[1], false}, {&ctx->pml_zeta_fw[1], ctx->pml_bytes[1], false},
                {&ctx->pml_psi_fw[2], ctx->pml_bytes[2], false}, {&ctx->pml_zeta_fw[2], ctx->pml_bytes[2], false},
                {&ctx->pml_snap, (size_t)nseg * 2 * (ctx->pml_bytes[0] + ctx->pml_bytes[1] + ctx->pml_bytes[2]), false}};
            ctx->pml_snap_stride = 2 * (ctx->pml_bytes[0] + ctx->pml_bytes[1] + ctx->pml_bytes[2]);
            size_t total = 0, fr = 0, tot = 0;
            for (auto &w : want) total += w.bytes;
            HIPCHK(ctx, hipMemGetInfo(&fr, &tot));
            hipError_t e = hipSuccess;
            if (total <= fr)
                for (auto &w : want) {
                    if (!w.bytes) continue;
                    if ((e = hipMalloc(w.p, w.bytes)) != hipSuccess) break;
                    if (w.zero && (e = hipMemsetAsync(*w.p, 0, w.bytes, ctx->stream)) != hipSuccess) break;
                }
            if (total > fr || e != hipSuccess) {
                (void)hipGetLastError();
                for (auto &w : want) {
                    if (*w.p) (void)hipFree(*w.p);
                    *w.p = nullptr;
                }
                return ctx->fail(FWI_ENOMEM,
                                 "checkpoint buffers need %.1f GiB (nt_max=%d, ckpt_interval=%d) but only %.1f GiB "
                                 "are free; a ckpt_interval near sqrt(2 nt_max) needs the least memory",
                                 total / 1073741824.0, ctx->cfg.nt_max, K, fr / 1073741824.0);
            }
            ctx->ckpt_ready = true;
        }
        // the pinned staging buffer at its final size for this shot, before any copy is queued on it
        if ((rc = stage_reserve(ctx, (size_t)nt * std::max(std::max(nsrc, nrec), 1) * sizeof(T)))) return rc;
        if ((rc = ensure(ctx, &ctx->wav, &ctx->cap_wav, (size_t)nt * std::max(nsrc, 1) * sizeof(T)))) return rc;
        if ((rc = ensure(ctx, &ctx->amp, &ctx->cap_amp, (size_t)nt * std::max(nrec, 1) * sizeof(T)))) return rc;
        if ((rc = ensure(ctx, &ctx->series, &ctx->cap_series,
                         (size_t)nt * std::max(std::max(nsrc, nrec), 1) * sizeof(T))))
            return rc;
        ctx->nt = nt;
        ctx->nsrc = nsrc;
        ctx->nrec = nrec;
        ctx->have_forward = false;
        ctx->have_q = false;
        ctx->have_syn = false;
        ctx->have_dev_residual = false;
        hipStream_t s = ctx->stream;
        if (nsrc && (rc = upload_amplitudes(ctx, ctx->wav, wavelet, nt, nsrc, ctx->src_sp))) return rc;
        if ((rc = zero_fields(ctx, ctx->u[0], ctx->u[1]))) return rc;
        Sweep sw;
        sw.f[0] = ctx->u[0];
        sw.f[1] = ctx->u[1];
        sw.v = ctx->vf;
        if (ctx->inc) HIPCHK(ctx, hipMemsetAsync(ctx->vf, 0, (size_t)g.ptot * sizeof(T), ctx->stream));
        if (ctx->cpml && (rc = pml_zero(ctx))) return rc;
        T *q_store = (T *)ctx->q_store;
        T *series = (T *)ctx->series;
        auto none = [](int) -> T * { return nullptr; };
        auto noq = [](int, const T *&p, const T *&p2) { p = p2 = nullptr; };
        const bool fused = use_fused(ctx, nt);  // 2-D: FUSED2D_STEPS time steps per launch
        // a step count that is not a multiple of FUSED2D_STEPS: the bulk fused, the last 1-3 steps one per launch
        const int nfused = mixed_fused_steps(ctx, nt);
        void *spare[2] = {ctx->fx[0], ctx->fx[1]};
        // 3-D: a forward sweep that keeps nothing for imaging runs two time steps per pass (the last step of an
        // odd count goes through the single-step kernel)
        const int npair = (ctx->pair3d && !save && base_args(ctx, 0).damp == 0) ? (nt & ~1) : 0;
        if ((fused || nfused || npair) && (rc = zero_fields(ctx, spare[0], spare[1]))) return rc;
        TimeLoop loop(ctx);
        if ((rc = loop.begin())) return rc;
        if (npair) {
            if ((rc = run_pairs(ctx, sw, spare, 0, npair, ctx->src, (const T *)ctx->wav, &ctx->rec, series, T(1))))
                return rc;
            if ((rc = run_steps(ctx, sw, npair, 1, nt - npair, ctx->src, (const T *)ctx->wav, &ctx->rec, series, T(1),
                                none, noq)))
                return rc;
        } else if (save && K > 0) {
            for (int n0 = 0, seg = 0; n0 < nt; n0 += K, ++seg) {
                T *sn = (T *)ctx->snap + (size_t)seg * 2 * g.ptot;  // (u^n0, u^{n0-1})
                HIPCHK(ctx, hipMemcpyAsync(sn, sw.f[sw.cur], (size_t)g.ptot * sizeof(T), hipMemcpyDeviceToDevice, s));
                HIPCHK(ctx, hipMemcpyAsync(sn + g.ptot, ctx->inc ? sw.v : sw.f[sw.cur ^ 1], (size_t)g.ptot * sizeof(T),
                                           hipMemcpyDeviceToDevice, s));  // (u^n0, u^{n0-1}) or, increment form, (u^n0, v^n0)
                if (ctx->cpml && (rc = pml_snapshot(ctx, seg, false, false))) return rc;
                const int cnt = std::min(K, nt - n0);
                if (fused)
                    rc = run_fused(ctx, sw, spare, n0, 1, cnt, ctx->src, (const T *)ctx->wav, &ctx->rec, series,
                                   T(1), 0, nullptr);
                else
                    rc = run_steps(ctx, sw, n0, 1, cnt, ctx->src, (const T *)ctx->wav, &ctx->rec, series, T(1),
                                   none, noq);
                if (rc) return rc;
            }
        } else if (fused) {
            if ((rc = run_fused(ctx, sw, spare, 0, 1, nt, ctx->src, (const T *)ctx->wav, &ctx->rec, series, T(1),
                                save ? 1 : 0, q_store)))
                return rc;
        } else {
            const int ks = ctx->istride;  // q^n is kept for n % ks == 0, in slot n / ks
            auto qo = [&](int n) -> T * {  // (slot addresses in bytes: the store may hold bf16)
                return (save && n % ks == 0) ? (T *)((char *)q_store + (size_t)(n / ks) * g.npts * ctx->qes) : nullptr;
            };
            if (nfused && (rc = run_fused(ctx, sw, spare, 0, 1, nfused, ctx->src, (const T *)ctx->wav, &ctx->rec, series,
                                          T(1), save ? 1 : 0, q_store)))
                return rc;
            if ((rc = run_steps(ctx, sw, nfused, 1, nt - nfused, ctx->src, (const T *)ctx->wav, &ctx->rec, series, T(1),
                                qo, noq)))
                return rc;
        }
        if ((rc = flush_record(ctx, sw, ctx->rec, series, T(1)))) return rc;
        if ((rc = loop.end())) return rc;
        if (fused || nfused || npair) {  // the buffer pairs may have changed roles: keep ownership consistent
            ctx->u[0] = sw.f[0];
            ctx->u[1] = sw.f[1];
            ctx->fx[0] = spare[0];
            ctx->fx[1] = spare[1];
            if (ctx->inc) ctx->vf = sw.v;
        }
        if ((rc = download_samples(ctx, (nrec && seis_out) ? seis_out : nullptr, ctx->series, nt, nrec, ctx->rec_sp,
                                   &ctx->pts_d, &ctx->cap_pts_d)))
            return rc;
        ctx->have_forward = true;
        ctx->have_syn = true;
        ctx->have_q = save != 0;
        return FWI_OK;
    }

    static int adjoint(fwi_ctx *ctx, const T *residual, int32_t image, T *adj_src_out) {
        const GridDesc &g = ctx->gd;
        const int nt = ctx->nt, K = ctx->ckpt;
        hipStream_t s = ctx->stream;
        int rc;
        if (ctx->nrec && residual && (rc = upload_amplitudes(ctx, ctx->amp, residual, nt, ctx->nrec, ctx->rec_sp)))
            return rc;  // (residual == nullptr: the one fwi_misfit_l2 left in ctx->amp)
        ctx->have_dev_residual = false;  // consumed: ctx->series is about to be overwritten
        ctx->have_syn = false;           // ... by this sweep's source-side series (fwi_misfit_l2 must not read them)
        const T rs = (T)(1.0 / std::pow(ctx->cfg.h, g.ndim));
        if ((rc = zero_fields(ctx, ctx->u[0], ctx->u[1]))) return rc;
        Sweep sw;
        sw.f[0] = ctx->u[0];
        sw.f[1] = ctx->u[1];
        sw.v = ctx->vf;
        if (ctx->inc) HIPCHK(ctx, hipMemsetAsync(ctx->vf, 0, (size_t)g.ptot * sizeof(T), ctx->stream));
        if (ctx->cpml && (rc = pml_zero(ctx))) return rc;
        T *q_store = (T *)ctx->q_store;
        T *series = (T *)ctx->series;
        const T *amp = (const T *)ctx->amp;
        auto none = [](int) -> T * { return nullptr; };
        auto noq = [](int, const T *&p, const T *&p2) { p = p2 = nullptr; };
        const bool fused = use_fused(ctx, nt);
        const int nfused = mixed_fused_steps(ctx, nt);
        void *spare[2] = {ctx->fx[0], ctx->fx[1]};
        if ((fused || nfused) && (rc = zero_fields(ctx, spare[0], spare[1]))) return rc;
        TimeLoop loop(ctx);
        if ((rc = loop.begin())) return rc;
        const T *q0 = nullptr;  // q^0, for the last pairing (mu^1 with q^0)
        // Lagged imaging: adjoint step n (which reads mu^{n+2} as u_cur and mu^{n+3} as u_prev) owns
        // the pairing P(n) = (mu^{n+2}, q^{n+1}).  Steps are taken in pairs: P(n+1) is deferred and
        // taken at step n together with P(n) -- u_prev there is still mu^{n+3} -- so the gradient
        // accumulator is read and written every other step only.  plan() fills, for the steps
        // n = hi .. lo (descending) of one run, the q pointers of P(n) and of a deferred P(n+1).
        std::vector<const T *> pq, pq2;
        auto plan = [&](int hi, int lo, auto qptr /* n -> q^{n+1} or nullptr */) {
            pq.assign(hi - lo + 1, nullptr);
            pq2.assign(hi - lo + 1, nullptr);
            const T *pending = nullptr;  // deferred pairing of the step above
            for (int n = hi; n >= lo; --n) {
                const T *mine = qptr(n);
                if (pending) {  // second step of a pair: take both (mine is valid whenever pending is)
                    pq[hi - n] = mine;
                    pq2[hi - n] = pending;
                    pending = nullptr;
                } else if (mine && n > lo && ctx->istride == 1) {  // (increment form too since round 4: u_prev = u - v)
                    pending = mine;  // first step of a pair: defer (every step images: the next one pairs)
                } else {
                    pq[hi - n] = mine;  // single (last step of an odd run, or nothing to pair)
                }
            }
        };
        bool imaged_all = false;  // the fused kernel pairs mu^{n+1} with q^n inside the launch (no lag)
        if (fused && image && K > 0) {
            // checkpointed, fused: recompute each segment's forward storing q into the slot buffer,
            // then sweep the segment backwards imaging against those slots (no carry: nothing lags).
            // The recomputation ping-pongs between its own two buffer pairs (fwd, fwx).
            Sweep fw;
            fw.f[0] = ctx->fwd[0];
            fw.f[1] = ctx->fwd[1];
            fw.v = ctx->fwv;
            fw.pml_fw = true;
            void *fspare[2] = {ctx->fwx[0], ctx->fwx[1]};
            const int nseg = (nt + K - 1) / K;
            for (int seg = nseg - 1; seg >= 0; --seg) {
                const int n0 = seg * K, cnt = std::min(K, nt - n0);
                const T *sn = (const T *)ctx->snap + (size_t)seg * 2 * g.ptot;
                fw.cur = 0;
                HIPCHK(ctx, hipMemcpyAsync(fw.f[0], sn, (size_t)g.ptot * sizeof(T), hipMemcpyDeviceToDevice, s));
                HIPCHK(ctx, hipMemcpyAsync(ctx->inc ? fw.v : fw.f[1], sn + g.ptot, (size_t)g.ptot * sizeof(T),
                                           hipMemcpyDeviceToDevice, s));
                if (ctx->cpml && (rc = pml_snapshot(ctx, seg, true, true))) return rc;
                T *qb = q_store - (size_t)n0 * g.npts;  // slot (n - n0) == qb + n * npts
                if ((rc = run_fused(ctx, fw, fspare, n0, 1, cnt, ctx->src, (const T *)ctx->wav, nullptr, nullptr,
                                    T(0), 1, qb)))
                    return rc;
                if ((rc = run_fused(ctx, sw, spare, n0 + cnt - 1, -1, cnt, ctx->rec, amp, &ctx->src, series, rs, 2,
                                    qb)))
                    return rc;
            }
            ctx->fwd[0] = fw.f[0];
            ctx->fwd[1] = fw.f[1];
            ctx->fwx[0] = fspare[0];
            ctx->fwx[1] = fspare[1];
            if (ctx->inc) ctx->fwv = fw.v;
            imaged_all = true;
        } else if (fused) {
            if ((rc = run_fused(ctx, sw, spare, nt - 1, -1, nt, ctx->rec, amp, &ctx->src, series, rs, image ? 2 : 0,
                                q_store)))
                return rc;
            imaged_all = true;
        } else if (image && K > 0) {
            // Checkpointed: per segment (last to first) restore the snapshot, recompute the forward
            // steps storing q into the K-slot buffer, then take the adjoint steps of that segment.
            // At a segment's first adjoint step q^{n+1} is the first q of the segment processed
            // before, kept in the carry slot.
            T *carry = q_store + (size_t)K * g.npts;
            const int nseg = (nt + K - 1) / K;
            for (int seg = nseg - 1; seg >= 0; --seg) {
                const int n0 = seg * K, cnt = std::min(K, nt - n0);
                const T *sn = (const T *)ctx->snap + (size_t)seg * 2 * g.ptot;
                HIPCHK(ctx, hipMemcpyAsync(ctx->fwd[0], sn, (size_t)g.ptot * sizeof(T), hipMemcpyDeviceToDevice, s));
                HIPCHK(ctx, hipMemcpyAsync(ctx->inc ? ctx->fwv : ctx->fwd[1], sn + g.ptot, (size_t)g.ptot * sizeof(T),
                                           hipMemcpyDeviceToDevice, s));
                Sweep fw;
                fw.f[0] = ctx->fwd[0];
                fw.f[1] = ctx->fwd[1];
                fw.v = ctx->fwv;
                fw.pml_fw = true;
                if (ctx->cpml && (rc = pml_snapshot(ctx, seg, true, true))) return rc;
                auto qo = [&](int n) -> T * { return q_store + (size_t)(n - n0) * g.npts; };
                if ((rc = run_steps(ctx, fw, n0, 1, cnt, ctx->src, (const T *)ctx->wav, nullptr, nullptr, T(0), qo,
                                    noq)))
                    return rc;
                const int hi = n0 + cnt - 1;
                plan(hi, n0, [&](int n) -> const T * {
                    if (n + 1 >= nt) return nullptr;
                    return (n + 1 < n0 + cnt) ? q_store + (size_t)(n + 1 - n0) * g.npts : carry;
                });
                auto qi = [&](int n, const T *&p, const T *&p2) { p = pq[hi - n]; p2 = pq2[hi - n]; };
                if ((rc = run_steps(ctx, sw, hi, -1, cnt, ctx->rec, amp, &ctx->src, series, rs, none, qi)))
                    return rc;
                HIPCHK(ctx, hipMemcpyAsync(carry, q_store, (size_t)g.npts * sizeof(T), hipMemcpyDeviceToDevice, s));
            }
            q0 = carry;
        } else {
            // steps nt-1 .. nfused one per launch (all of them when nfused == 0), lagged pairing; the pairing owed
            // at the lower end, (mu^{nfused+1}, q^{nfused}), is the "last pairing" of that run
            const int ks = ctx->istride;
            plan(nt - 1, nfused, [&](int n) -> const T * {
                return (image && n + 1 < nt && (n + 1) % ks == 0)
                           ? (const T *)((const char *)q_store + (size_t)((n + 1) / ks) * g.npts * ctx->qes)
                           : nullptr;
            });
            auto qi = [&](int n, const T *&p, const T *&p2) { p = pq[nt - 1 - n]; p2 = pq2[nt - 1 - n]; };
            if ((rc = run_steps(ctx, sw, nt - 1, -1, nt - nfused, ctx->rec, amp, &ctx->src, series, rs, none, qi)))
                return rc;
            q0 = (nfused % ks == 0) ? (const T *)((const char *)q_store + (size_t)(nfused / ks) * g.npts * ctx->qes) : nullptr;
            if (nfused) {
                // ... then the first nfused steps, FUSED2D_STEPS per launch (pairing inside the launch, no lag)
                if ((rc = flush_record(ctx, sw, ctx->src, series, rs))) return rc;
                sw.prev_n = -1;
                if (image && q0)
                    HIPCHK(ctx, launch_image<T>(g, (const T *)sw.f[sw.cur], q0, (T *)ctx->g_acc, ctx->qbf16, s));
                if ((rc = run_fused(ctx, sw, spare, nfused - 1, -1, nfused, ctx->rec, amp, &ctx->src, series, rs,
                                    image ? 2 : 0, q_store)))
                    return rc;
                imaged_all = true;
            }
        }
        if (fused || nfused) {  // the buffer pairs may have changed roles: keep ownership consistent
            ctx->u[0] = sw.f[0];
            ctx->u[1] = sw.f[1];
            ctx->fx[0] = spare[0];
            ctx->fx[1] = spare[1];
            if (ctx->inc) ctx->vf = sw.v;
        }
        if ((rc = flush_record(ctx, sw, ctx->src, series, rs))) return rc;
        if (image && !imaged_all)  // the last pairing: mu^1 with q^0
            HIPCHK(ctx, launch_image<T>(g, (const T *)sw.f[sw.cur], q0, (T *)ctx->g_acc, ctx->qbf16, s));
        if (image && ctx->qbf16 && ctx->nsrc > 0)
            // the store holds C L u rounded to bf16; the source's own share C w / h^D of the forward term is paired
            // exactly, in closed form: g(x_s) += cq_s sum_n mu^{n+1}(x_s) w_s^n, with mu at the sources = the series
            // this sweep has just recorded (scaled by rs)
            HIPCHK(ctx, launch_source_image<T>((const T *)ctx->series, (const T *)ctx->wav, (const int64_t *)ctx->src.cidx,
                                               (const T *)ctx->src.cq, (T *)ctx->g_acc, nt, ctx->nsrc, ctx->istride,
                                               (T)(1.0 / (double)rs), s));
        if ((rc = loop.end())) return rc;
        return download_samples(ctx, (adj_src_out && ctx->nsrc) ? adj_src_out : nullptr, ctx->series, nt, ctx->nsrc,
                                ctx->src_sp, nullptr, nullptr);
    }

    static int gradient_vec(fwi_ctx *ctx, int32_t wrt, void *dev) {
        const double scale = -(double)ctx->istride / (ctx->cfg.dt * ctx->cfg.dt);  // istride: quadrature weight
        HIPCHK(ctx, launch_finalize_gradient<T>(ctx->gd, (const T *)ctx->g_acc, (const T *)ctx->c_dev, (T *)dev, scale,
                                                wrt == FWI_WRT_VELOCITY, ctx->stream));
        return FWI_OK;
    }

    // d_obs -> ctx->amp, then amp := series - amp and J = 1/2 sum amp^2, all on the device
    static int misfit_l2(fwi_ctx *ctx, const T *d_obs, double *J_out) {
        const fwi_ctx::SpreadSet &sp = ctx->rec_sp;
        const size_t n = (size_t)ctx->nt * (sp.npts ? sp.npts : ctx->nrec);
        double ss = 0.0;
        if (n && ctx->nrec) {
            // off-grid receivers: the residual lives per POINT (against the gathered synthetics kept by the forward)
            // and is then scattered onto the nodes, where the adjoint sweep injects it
            void *resid = ctx->amp;
            int rc;
            if (sp.npts) {
                if ((rc = ensure(ctx, &ctx->pts_a, &ctx->cap_pts_a, n * sizeof(T)))) return rc;
                resid = ctx->pts_a;
            }
            if ((rc = upload_series(ctx, resid, d_obs, n * sizeof(T)))) return rc;
            HIPCHK(ctx, hipMemsetAsync(ctx->red, 0, sizeof(double), ctx->stream));
            HIPCHK(ctx, launch_residual_l2<T>((const T *)(sp.npts ? ctx->pts_d : ctx->series), (T *)resid, (int64_t)n,
                                              ctx->red, ctx->stream));
            if (sp.npts)
                HIPCHK(ctx, launch_scatter_series<T>((const T *)resid, (T *)ctx->amp, (const int *)sp.owner,
                                                     (const T *)sp.weight, ctx->nt, sp.npts, ctx->nrec, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(&ss, ctx->red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        *J_out = 0.5 * ss;
        ctx->have_dev_residual = true;
        return FWI_OK;
    }

    static int vec_dot(fwi_ctx *ctx, const void *x, const void *y, double *out) {
        HIPCHK(ctx, hipMemsetAsync(ctx->red, 0, sizeof(double), ctx->stream));
        HIPCHK(ctx, launch_dot<T>((const T *)x, (const T *)y, ctx->gd.npts, ctx->red, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out, ctx->red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return FWI_OK;
    }

    static int vec_absmax(fwi_ctx *ctx, const void *x, double *out) {
        HIPCHK(ctx, hipMemsetAsync(ctx->red, 0, sizeof(double), ctx->stream));
        HIPCHK(ctx, launch_absmax<T>((const T *)x, ctx->gd.npts, ctx->red, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out, ctx->red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return FWI_OK;
    }

    static int vec_axpby(fwi_ctx *ctx, void *y, double a, const void *x, double b) {
        HIPCHK(ctx, launch_axpby<T>((T *)y, a, (const T *)x, b, ctx->gd.npts, ctx->stream));
        return FWI_OK;
    }

    static int vec_clip(fwi_ctx *ctx, void *x, double lo, double hi) {
        HIPCHK(ctx, launch_clip<T>(ctx->gd, (T *)x, lo, hi, ctx->stream));
        return FWI_OK;
    }

    static int gradient(fwi_ctx *ctx, int32_t wrt, T *out) {
        const GridDesc &g = ctx->gd;
        const double scale = -(double)ctx->istride / (ctx->cfg.dt * ctx->cfg.dt);  // istride: quadrature weight
        HIPCHK(ctx, launch_finalize_gradient<T>(g, (const T *)ctx->g_acc, (const T *)ctx->c_dev,
                                                (T *)ctx->g_out, scale, wrt == FWI_WRT_VELOCITY, ctx->stream));
        int rc = download_compact(ctx, out, ctx->g_out);
        if (rc) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return FWI_OK;
    }

    static int dot(fwi_ctx *ctx, const T *a, const T *b, int64_t n, double *out) {
        void *da = nullptr, *db = nullptr;
        HIPCHK(ctx, hipMalloc(&da, (size_t)n * sizeof(T)));
        hipError_t e = hipMalloc(&db, (size_t)n * sizeof(T));
        if (e != hipSuccess) {
            (void)hipFree(da);
            return ctx->fail(FWI_ENOMEM, "hipMalloc: %s", hipGetErrorString(e));
        }
        int rc = FWI_OK;
        do {
            if ((e = hipMemcpyAsync(da, a, (size_t)n * sizeof(T), hipMemcpyHostToDevice, ctx->stream))) break;
            if ((e = hipMemcpyAsync(db, b, (size_t)n * sizeof(T), hipMemcpyHostToDevice, ctx->stream))) break;
            if ((e = hipMemsetAsync(ctx->red, 0, sizeof(double), ctx->stream))) break;
            if ((e = launch_dot<T>((const T *)da, (const T *)db, n, ctx->red, ctx->stream))) break;
            if ((e = hipMemcpyAsync(out, ctx->red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream))) break;
            e = hipStreamSynchronize(ctx->stream);
        } while (0);
        if (e != hipSuccess) rc = ctx->fail(FWI_EHIP, "fwi_dot: %s", hipGetErrorString(e));
        (void)hipFree(da);
        (void)hipFree(db);
        return rc;
    }
};

int create_impl(fwi_ctx *ctx) {
    const fwi_config &c = ctx->cfg;
    const GridDesc &g = ctx->gd;
    const size_t es = ctx->esize;
    HIPCHK(ctx, hipSetDevice(c.device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    const size_t fslack = ctx->place_kind == 2 ? ctx->place_pad : 0;  // (tune_placement may move C; the u buffers
                                                                       // change roles with others and stay where they are)
    for (int i = 0; i < 2; ++i) {
        HIPCHK(ctx, hipMalloc(&ctx->u[i], (size_t)g.ptot * es));
        HIPCHK(ctx, hipMemsetAsync(ctx->u[i], 0, (size_t)g.ptot * es, ctx->stream));
    }
    HIPCHK(ctx, hipMalloc(&ctx->C, (size_t)g.ptot * es + fslack));
    HIPCHK(ctx, hipMemsetAsync(ctx->C, 0, (size_t)g.ptot * es + fslack, ctx->stream));
    HIPCHK(ctx, hipMalloc(&ctx->c_dev, (size_t)g.npts * es));
    HIPCHK(ctx, hipMalloc(&ctx->g_acc, (size_t)g.npts * es));
    HIPCHK(ctx, hipMemsetAsync(ctx->g_acc, 0, (size_t)g.npts * es, ctx->stream));
    HIPCHK(ctx, hipMalloc(&ctx->g_out, (size_t)g.npts * es));
    HIPCHK(ctx, hipMalloc(&ctx->dz, (size_t)g.nz * es));
    HIPCHK(ctx, hipMalloc(&ctx->dy, (size_t)g.ny * es));
    HIPCHK(ctx, hipMalloc(&ctx->dx, (size_t)g.nx * es));
    HIPCHK(ctx, hipMalloc((void **)&ctx->red, 64));
    if (ctx->fused2d) {  // tile order of the fused 2-D kernel on grids of more than one round of tiles (on THIS device)
        std::vector<int> order;
        fused2d_tile_order(g, ctx->fused_ft, c.npml, order, ctx->cpml ? 1 : 0);
        if (!order.empty()) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->fused_order, order.size() * sizeof(int)));
            HIPCHK(ctx, hipMemcpy(ctx->fused_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    double sm = c.sigma_max;
    const int np_sponge = ctx->cpml ? 0 : c.npml;  // with the CPML the damping factors are off (A = B = 1)
    profile(ctx->pz, g.nz, np_sponge, sm, c.dt);
    profile(ctx->px, g.nx, np_sponge, sm, c.dt);
    if (g.ndim == 3) profile(ctx->py, g.ny, np_sponge, sm, c.dt); else ctx->py.assign(1, 0.0);
    if (ctx->cpml) {
        const int nd[3] = {g.nz, g.ny, g.nx};
        for (int d = 0; d < 3; ++d) {
            if (d == 1 && g.ndim == 2) continue;
            const int n = nd[d], nslab = std::min(n, 2 * c.npml);
            size_t cnt = (size_t)nslab;  // rows padded to cx (16-byte lanes of the z / y border kernels)
            for (int o = 0; o < 3; ++o)
                if (o != d) cnt *= (size_t)(o == 2 ? g.cx : nd[o]);
            ctx->pml_bytes[d] = cnt * es;
            const size_t slack = (d == 2 && ctx->place_kind == 1) ? ctx->place_pad : 0;  // (the x border's arrays can be moved: tune_placement)
            HIPCHK(ctx, hipMalloc(&ctx->pml_psi[d], ctx->pml_bytes[d] + slack));
            HIPCHK(ctx, hipMalloc(&ctx->pml_zeta[d], ctx->pml_bytes[d] + slack));
            if (ctx->fused2d) {  // the set the fused launch writes (see fwi_ctx::pml_spare_psi)
                HIPCHK(ctx, hipMalloc(&ctx->pml_spare_psi[d], ctx->pml_bytes[d]));
                HIPCHK(ctx, hipMalloc(&ctx->pml_spare_zeta[d], ctx->pml_bytes[d]));
                HIPCHK(ctx, hipMemsetAsync(ctx->pml_spare_psi[d], 0, ctx->pml_bytes[d], ctx->stream));
                HIPCHK(ctx, hipMemsetAsync(ctx->pml_spare_zeta[d], 0, ctx->pml_bytes[d], ctx->stream));
            }
            HIPCHK(ctx, hipMalloc(&ctx->pml_a[d], (size_t)n * es));
            HIPCHK(ctx, hipMalloc(&ctx->pml_b[d], (size_t)n * es));
            std::vector<double> pa(n), pb(n);  // cpml_profiles() of the oracle
            for (int i = 0; i < n; ++i) {
                const double dist = std::max(0.0, std::max((double)c.npml - i, (double)i - (n - 1 - c.npml)));
                const double x = dist / c.npml, sig = sm * x * x, alp = c.pml_alpha_max * (1.0 - x);
                pb[i] = std::exp(-(sig + alp) * c.dt);
                pa[i] = sig > 0.0 ? sig / (sig + alp) * (pb[i] - 1.0) : 0.0;
            }
            int rc2;
            if (c.dtype == FWI_F32) {
                if ((rc2 = upload_vec<float>(ctx, ctx->pml_a[d], pa)) || (rc2 = upload_vec<float>(ctx, ctx->pml_b[d], pb)))
                    return rc2;
            } else {
                if ((rc2 = upload_vec<double>(ctx, ctx->pml_a[d], pa)) || (rc2 = upload_vec<double>(ctx, ctx->pml_b[d], pb)))
                    return rc2;
            }
        }
    }
    if (ctx->cpml && ctx->pml_lines == 3) {
        // the handed-over border terms: (shell rows of z, ny, cx) and (nz + 1, shell rows of y, cx) -- one plane more than
        // the grid, the stream kernel's prefetch runs one plane past the end.  Zeroed once: rows off a thread's shell
        // re-read the first vectors of the arrays and discard them, which must not be NaNs of an earlier allocation.
        const size_t bz = (size_t)pml_shell_rows(g.nz, c.npml, g.r) * g.ny * g.cx * es;
        const size_t by = (size_t)(g.nz + 1) * pml_shell_rows(g.ny, c.npml, g.r) * g.cx * es;
        const size_t slack = ctx->place_kind == 1 ? ctx->place_pad : 0;
        HIPCHK(ctx, hipMalloc(&ctx->pml_tz, bz + slack));
        HIPCHK(ctx, hipMalloc(&ctx->pml_ty, by + slack));
        HIPCHK(ctx, hipMemsetAsync(ctx->pml_tz, 0, bz + slack, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->pml_ty, 0, by + slack, ctx->stream));
        if (ctx->place_kind == 1) {  // search order: the positions that decided the slab experiments first
            ctx->movable(&ctx->pml_ty, by);
            ctx->movable(&ctx->pml_zeta[2], ctx->pml_bytes[2]);
            ctx->movable(&ctx->pml_tz, bz);
            ctx->movable(&ctx->pml_psi[2], ctx->pml_bytes[2]);
        }
    }
    int rc;
    if (c.dtype == FWI_F32) {
        if ((rc = upload_vec<float>(ctx, ctx->dz, ctx->pz))) return rc;
        if ((rc = upload_vec<float>(ctx, ctx->dy, ctx->py))) return rc;
        if ((rc = upload_vec<float>(ctx, ctx->dx, ctx->px))) return rc;
    } else {
        if ((rc = upload_vec<double>(ctx, ctx->dz, ctx->pz))) return rc;
        if ((rc = upload_vec<double>(ctx, ctx->dy, ctx->py))) return rc;
        if ((rc = upload_vec<double>(ctx, ctx->dx, ctx->px))) return rc;
    }
    return FWI_OK;
}

}  // namespace

extern "C" {

int fwi_abi_version(void) { return FWI_ABI_VERSION; }

int fwi_device_count(int32_t *n_out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (n_out) *n_out = (e == hipSuccess) ? n : 0;
    return (e == hipSuccess && n > 0) ? FWI_OK : FWI_EHIP;
}

const char *fwi_last_error(const fwi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

const char *fwi_kernel_name(const fwi_ctx *ctx) {
    if (!ctx) return "";
    if (ctx->fused2d) return "step2d_fused";
    if (ctx->kernel == K_STREAM && ctx->gd.ndim == 2) return "step2d_tile";
    if (ctx->kernel == K_STREAM) return "step3d_stream";
    return "step_point";
}

int fwi_create(const fwi_config *cfg, fwi_ctx **out) {
    if (!cfg || !out) {
        g_create_error = "fwi_create: null argument";
        return FWI_EINVAL;
    }
    *out = nullptr;
    auto bad = [&](const char *m) {
        g_create_error = std::string("fwi_create: ") + m;
        return FWI_EINVAL;
    };
    if (cfg->struct_size != (int32_t)sizeof(fwi_config)) return bad("struct_size mismatch (ABI)");
    if (cfg->ndim != 2 && cfg->ndim != 3) return bad("ndim must be 2 or 3");
    if (cfg->nz < 1 || cfg->nx < 1 || (cfg->ndim == 3 && cfg->ny < 1)) return bad("empty grid");
    if (cfg->order != 2 && cfg->order != 4 && cfg->order != 8) return bad("order must be 2, 4 or 8");
    if (cfg->nt_max < 1) return bad("nt_max must be >= 1");
    if (cfg->npml < 0) return bad("npml must be >= 0");
    if (!(cfg->h > 0) || !(cfg->dt > 0)) return bad("h and dt must be > 0");
    if (cfg->dtype != FWI_F32 && cfg->dtype != FWI_F64) return bad("dtype must be FWI_F32 or FWI_F64");
    if (cfg->npml > 0 && !(cfg->sigma_max >= 0)) return bad("sigma_max must be >= 0 when npml > 0");
    if (cfg->kernel < FWI_KERNEL_AUTO || cfg->kernel > FWI_KERNEL_STREAM) return bad("unknown kernel id");
    if (cfg->ckpt_interval < 0) return bad("ckpt_interval must be >= 0");
    if (cfg->image_stride < 0) return bad("image_stride must be >= 0");
    if (cfg->image_stride > 1 && cfg->ckpt_interval > 0)
        return bad("image_stride > 1 and ckpt_interval > 0 cannot be combined");
    if (cfg->update_form != FWI_UPDATE_STANDARD && cfg->update_form != FWI_UPDATE_INCREMENT)
        return bad("unknown update_form");
    if (cfg->abc != FWI_ABC_SPONGE && cfg->abc != FWI_ABC_CPML) return bad("unknown abc");
    if (cfg->store_dtype != FWI_STORE_NATIVE && cfg->store_dtype != FWI_STORE_BF16) return bad("unknown store_dtype");
    if (cfg->store_dtype == FWI_STORE_BF16 && cfg->dtype != FWI_F32) return bad("store_dtype bf16 needs an fp32 context");
    if (cfg->launch_mode < FWI_LAUNCH_AUTO || cfg->launch_mode > FWI_LAUNCH_GRAPH) return bad("unknown launch_mode");
    if (!(cfg->pml_alpha_max >= 0)) return bad("pml_alpha_max must be >= 0");
    if (cfg->store_dtype == FWI_STORE_BF16 &&
        (cfg->ndim != 3 || cfg->order != 8 || cfg->kernel == FWI_KERNEL_POINT || cfg->ckpt_interval > 0 ||
         cfg->update_form != FWI_UPDATE_STANDARD || (cfg->abc == FWI_ABC_CPML && cfg->npml > 0)))
        return bad("store_dtype bf16: 3-D fp32 O(8) stream kernel, standard update form, sponge border, no checkpointing");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        g_create_error = "fwi_create: no HIP device available (this library has no CPU fallback)";
        return FWI_EHIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return bad("device ordinal out of range");

    fwi_ctx *ctx = new (std::nothrow) fwi_ctx;
    if (!ctx) {
        g_create_error = "fwi_create: out of host memory";
        return FWI_ENOMEM;
    }
    ctx->cfg = *cfg;
    // (FWI_XPITCH_EXTRA: the layout A/B hook of DESIGN.md's pitch sweep, read ONCE per process -- not on any hot path)
    static const int xpitch_extra = getenv("FWI_XPITCH_EXTRA") ? atoi(getenv("FWI_XPITCH_EXTRA")) : 0;
    ctx->gd = make_grid(cfg->ndim, cfg->nz, cfg->ny, cfg->nx, cfg->order, xpitch_extra);
    ctx->esize = cfg->dtype == FWI_F32 ? 4 : 8;
    ctx->ckpt = cfg->ckpt_interval;
    ctx->istride = cfg->image_stride > 1 ? cfg->image_stride : 1;
    ctx->inc = cfg->update_form == FWI_UPDATE_INCREMENT;
    ctx->cpml = cfg->abc == FWI_ABC_CPML && cfg->npml > 0;
    ctx->qbf16 = cfg->store_dtype == FWI_STORE_BF16;
    ctx->graph_mode = cfg->launch_mode == FWI_LAUNCH_GRAPH;  // (AUTO = stream: the measurement is with the comment below)
    ctx->qes = ctx->qbf16 ? 2 : (cfg->dtype == FWI_F32 ? 4 : 8);
    const bool can_stream = stream_supported(ctx->gd, cfg->dtype == FWI_F32);
    if (cfg->kernel == FWI_KERNEL_STREAM && !can_stream) {
        delete ctx;
        return bad("STREAM kernel: fp64 is 3-D only");
    }
    ctx->kernel = (cfg->kernel == FWI_KERNEL_POINT || !can_stream) ? K_POINT : K_STREAM;
    if (ctx->inc && !(cfg->ndim == 3 && cfg->dtype == FWI_F32)) {
        // the increment form exists in the 3-D fp32 stream kernel and in the point kernel
        if (cfg->kernel == FWI_KERNEL_STREAM) {
            delete ctx;
            return bad("update_form increment: the STREAM kernel takes it for 3-D fp32 only");
        }
        ctx->kernel = K_POINT;
    }
    if (ctx->kernel == K_STREAM) {
        // what a time step touches beside the three fields: the increment field; the CPML's memory variables (psi, zeta
        // over 2 npml planes per axis) and the handed-over border terms (two shells of npml + r)
        double extra = ctx->inc ? (double)ctx->gd.ptot * ctx->esize : 0.0;
        if (ctx->cpml && cfg->ndim == 3) {
            const double face[3] = {(double)ctx->gd.ny * ctx->gd.cx, (double)ctx->gd.nz * ctx->gd.cx, (double)ctx->gd.nz * ctx->gd.ny};
            for (int d = 0; d < 3; ++d) extra += 2.0 * std::min(2 * cfg->npml, d == 0 ? cfg->nz : d == 1 ? cfg->ny : cfg->nx) * face[d] * ctx->esize;
            for (int d = 0; d < 2; ++d) extra += 2.0 * (cfg->npml + ctx->gd.r) * face[d] * ctx->esize;
        }
        ctx->tune = stream_default_tuning(ctx->gd, cfg->dtype == FWI_F32, extra);
        if (cfg->zchunk > 0) ctx->tune.zchunk = cfg->zchunk;
        if (const char *pf = getenv("FWI_STREAM_PF")) {  // tuning hook: prefetch depth in planes
            const int v = atoi(pf);
            if (v >= 1 && v <= 3) ctx->tune.pf = v;
        }
        if (const char *ty = getenv("FWI_STREAM_TY")) {  // tuning hook: rows per workgroup
            const int v = atoi(ty);
            if (v == 4 || v == 8 || (v == 16 && cfg->ndim == 2)) ctx->tune.ty = v;
        }
    }
    // 2-D fp32 grids: advance FUSED2D_STEPS time steps per launch (fwi_fused2d.hip) whenever the step
    // count allows it (FWI_NO_FUSED2D is the tuning / comparison hook)
    // (CPML: the fused kernel carries the border recursion itself where the tiling admits it, fused2d_cpml_supported;
    // otherwise the slab kernels run between time steps: one step per launch, the tile kernel.  FWI_NO_FUSED2D_CPML is
    // the comparison hook)
    // Increment form in 2-D: the fused kernel carries it (same traffic as the standard form); the steps it cannot take
    // (step counts off the multiple of 4) go through the point kernel, which ctx->kernel names in that case.
    if (const char *sk = getenv("FWI_FUSED2D_SKIPD")) ctx->fused_skipd = atoi(sk) != 0;  // tuning / test hook
    const bool cpml_in_launch = ctx->cpml && !ctx->inc && ctx->kernel == K_STREAM &&
                                fused2d_cpml_supported(ctx->gd, cfg->npml) && !getenv("FWI_NO_FUSED2D_CPML");
    ctx->fused2d = cfg->ndim == 2 && cfg->dtype == FWI_F32 && (!ctx->cpml || cpml_in_launch) && !getenv("FWI_NO_FUSED2D") &&
                   (ctx->inc ? cfg->kernel == FWI_KERNEL_AUTO : ctx->kernel == K_STREAM);
    if (ctx->fused2d && !ctx->cpml && !ctx->inc) ctx->fused_ft = fused2d_pick_tile(ctx->gd);
    // 3-D fp32 stream contexts: two time steps per pass for forward sweeps without imaging (FWI_STREAM_PAIR=0 /
    // =1 is the tuning / comparison hook)
    if (cfg->ndim == 3 && cfg->dtype == FWI_F32 && ctx->kernel == K_STREAM && !ctx->inc && !ctx->cpml) {
        const char *pe = getenv("FWI_STREAM_PAIR");
        ctx->pair3d = (pe ? atoi(pe) != 0 : false) && cfg->npml == 0;  // (the two-step kernel exists undamped only)
        if (ctx->pair3d) {
            pair3d_default_tuning(ctx->gd, &ctx->pair_zc, &ctx->pair_tw);
            if (const char *zc = getenv("FWI_PAIR_ZCHUNK")) {
                const int v = atoi(zc);
                if (v >= 1) ctx->pair_zc = v;
            }
        }
    }
    // CPML in 3-D: the z / y borders run as ONE line launch before every step, which hands their term to the step kernel
    // (both axes or none); with it the 3-D fp32 O(8) stream kernel carries the x border's recursion in its own lanes.
    // Whatever is left runs as slab phases around the step kernel.
    if (ctx->cpml && cfg->ndim == 3) ctx->pml_lines = pml_line_axes(ctx->gd, cfg->npml);
    ctx->xpml = ctx->cpml && ctx->kernel == K_STREAM && ctx->pml_lines == 3 && !getenv("FWI_NO_STREAM_XPML") &&
                stream_xpml_supported(ctx->gd, ctx->tune, cfg->npml, cfg->dtype == FWI_F32);
    // launch_mode AUTO = stream launches.  hipGraph capture of the time loop was built and measured in round 4 (VERDICT r03
    // item 3) with the mode switched between sweeps of ONE context: -0.6 % (3-D CPML, two kernels per step) ... +1.4 %
    // (2-D 256^2) of the loop time, nothing near the 3 % that would justify a second launch path by default; the host is
    // far ahead of the GPU either way (1.3 ms to submit 500 launches that run 3.1 ms at 256^2).  A first probe that ran the
    // two modes on two contexts side by side had shown the graph 6 - 13 % ahead on the 3-D CPML: an artefact of that set-up
    // (the stream-mode context ran 6 % slower than it does alone), which is why the A/B is done on one context now.
    if (getenv("FWI_DEBUG_PML"))
        fprintf(stderr, "fwi: cpml=%d fused2d=%d x-in-kernel=%d line-axes=%d (ty %d zchunk %d)\n", (int)ctx->cpml,
                (int)ctx->fused2d, (int)ctx->xpml, ctx->pml_lines, ctx->tune.ty, ctx->tune.zchunk);
    // 3-D CPML contexts past the cache-resident sizes place their small arrays by measurement (Impl::tune_placement).
    // FWI_PLACEMENT_TUNE: "0" = off, "pad" = padded allocations without the search (A/B hooks), "force" = also on grids
    // below the size threshold (tests: the oracle comparisons run small grids)
    const char *pt = getenv("FWI_PLACEMENT_TUNE");
    const bool place_on = cfg->dtype == FWI_F32 && cfg->ndim == 3 && ctx->kernel == K_STREAM && !(pt && !strcmp(pt, "0")) &&
                          ((double)ctx->gd.ptot * ctx->esize >= 48e6 || (pt && !strcmp(pt, "force")));
    if (place_on && ctx->xpml) ctx->place_kind = 1;
    else if (place_on && ctx->inc && !ctx->cpml) ctx->place_kind = 2;
    const bool place = ctx->place_kind != 0;
    if (place) ctx->place_pad = (size_t)14 << 20;
    int rc = create_impl(ctx);
    const size_t vslack = ctx->place_kind ? ctx->place_pad : 0;
    if (rc == FWI_OK && ctx->inc &&
        (hipMalloc(&ctx->vf, (size_t)ctx->gd.ptot * ctx->esize + vslack) != hipSuccess ||
         hipMemset(ctx->vf, 0, (size_t)ctx->gd.ptot * ctx->esize + vslack) != hipSuccess))
        rc = ctx->fail(FWI_ENOMEM, "allocating the increment field failed");
    if (rc == FWI_OK && ctx->place_kind && ctx->inc) ctx->movable(&ctx->vf, (size_t)ctx->gd.ptot * ctx->esize);
    if (rc == FWI_OK && ctx->place_kind == 2) ctx->movable(&ctx->C, (size_t)ctx->gd.ptot * ctx->esize);
    if (rc == FWI_OK && (ctx->fused2d || ctx->pair3d)) {
        for (int i = 0; i < 2 && rc == FWI_OK; ++i) {
            if (hipMalloc(&ctx->fx[i], (size_t)ctx->gd.ptot * ctx->esize) != hipSuccess ||
                hipMemset(ctx->fx[i], 0, (size_t)ctx->gd.ptot * ctx->esize) != hipSuccess)
                rc = ctx->fail(FWI_ENOMEM, "allocating the fused 2-D buffer pair failed");
        }
    }
    if (rc == FWI_OK && place && !(pt && !strcmp(pt, "pad"))) rc = Impl<float>::tune_placement(ctx);
    if (rc) {
        g_create_error = "fwi_create: " + ctx->err;
        fwi_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return FWI_OK;
}

void fwi_destroy(fwi_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->graph) (void)hipGraphDestroy(ctx->graph);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    void *ptrs[] = {ctx->u[0], ctx->u[1], ctx->C, ctx->c_dev, ctx->dz, ctx->dy, ctx->dx, ctx->q_store,
                    ctx->g_acc, ctx->g_out, ctx->red, ctx->amp, ctx->series, ctx->wav, ctx->snap, ctx->fwd[0],
                    ctx->fwd[1], ctx->fx[0], ctx->fx[1], ctx->fwx[0], ctx->fwx[1], ctx->logical, ctx->vf, ctx->fwv};
    for (fwi_ctx::PointSet *ps : {&ctx->src, &ctx->rec})
        for (void *p : {ps->pidx, ps->cidx, ps->cu, ps->cq, ps->s_start, ps->s_pidx, ps->s_cidx, ps->s_cu,
                        ps->s_cq, ps->s_col, ps->s_run, ps->fi_run, ps->fi_start, ps->fi_lz, ps->fi_lx, ps->fi_col, ps->fi_int,
                        ps->fi_cidx, ps->fi_cu, ps->fi_cq, ps->fr_start, ps->fr_lz, ps->fr_lx, ps->fr_col, ps->pr_start,
                        ps->pr_ent})
            if (p) (void)hipFree(p);
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *v : ctx->vecs)
        if (v) (void)hipFree(v);
    // (the four movable arrays: back to the start of their allocations)
    for (int i = 0; i < ctx->nmov; ++i)
        if (*ctx->mov_slot[i]) *ctx->mov_slot[i] = (char *)*ctx->mov_slot[i] - ctx->mov_shift[i];
    for (int d = 0; d < 3; ++d)
        for (void *q : {ctx->pml_psi[d], ctx->pml_zeta[d], ctx->pml_a[d], ctx->pml_b[d], ctx->pml_psi_fw[d],
                        ctx->pml_zeta_fw[d], ctx->pml_spare_psi[d], ctx->pml_spare_zeta[d]})
            if (q) (void)hipFree(q);
    if (ctx->pml_snap) (void)hipFree(ctx->pml_snap);
    for (void *q : {ctx->pml_tz, ctx->pml_ty})
        if (q) (void)hipFree(q);
    if (ctx->fused_order) (void)hipFree(ctx->fused_order);
    for (fwi_ctx::SpreadSet *sp : {&ctx->src_sp, &ctx->rec_sp})
        for (void *q : {sp->pt_start, sp->owner, sp->weight})
            if (q) (void)hipFree(q);
    for (void *q : {ctx->pts_a, ctx->pts_d})
        if (q) (void)hipFree(q);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

#define DISPATCH(ctx, expr32, expr64) ((ctx)->cfg.dtype == FWI_F32 ? (expr32) : (expr64))

int fwi_set_model(fwi_ctx *ctx, const void *c_host) {
    if (!ctx) return FWI_EINVAL;
    if (!c_host) return ctx->fail(FWI_EINVAL, "fwi_set_model: null model");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::set_model(ctx, (const float *)c_host),
                    Impl<double>::set_model(ctx, (const double *)c_host));
}

int fwi_forward(fwi_ctx *ctx, int32_t nt, int32_t nsrc, const int32_t *src_idx, const void *wavelet,
                int32_t nrec, const int32_t *rec_idx, int32_t save, void *seis_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->have_model) return ctx->fail(FWI_ESTATE, "fwi_forward: call fwi_set_model first");
    if (nt < 1 || nt > ctx->cfg.nt_max)
        return ctx->fail(FWI_EINVAL, "fwi_forward: nt=%d outside [1, nt_max=%d]", nt, ctx->cfg.nt_max);
    if (nsrc < 0 || nrec < 0) return ctx->fail(FWI_EINVAL, "fwi_forward: negative point count");
    if ((nsrc && (!src_idx || !wavelet)) || (nrec && (!rec_idx || !seis_out)))
        return ctx->fail(FWI_EINVAL, "fwi_forward: null buffer");
    (void)hipSetDevice(ctx->cfg.device);
    ctx->src_sp.npts = ctx->rec_sp.npts = 0;  // node-based call
    return DISPATCH(ctx,
                    Impl<float>::forward(ctx, nt, nsrc, src_idx, (const float *)wavelet, nrec, rec_idx,
                                         save, (float *)seis_out),
                    Impl<double>::forward(ctx, nt, nsrc, src_idx, (const double *)wavelet, nrec,
                                          rec_idx, save, (double *)seis_out));
}

// nodes of point p = entries pt_start[p] .. pt_start[p + 1] of the node list (at most 8); weight per entry
static int set_spread(fwi_ctx *ctx, fwi_ctx::SpreadSet &sp, int32_t npts, int32_t nnodes, const int32_t *pt_start,
                      const void *weight, const char *what) {
    sp.npts = 0;
    if (npts < 0 || (npts > 0 && (!pt_start || !weight)))
        return ctx->fail(FWI_EINVAL, "fwi_forward_spread: bad %s point set", what);
    if (npts == 0) {
        if (nnodes != 0) return ctx->fail(FWI_EINVAL, "fwi_forward_spread: %s nodes without points", what);
        return FWI_OK;
    }
    std::vector<int> owner((size_t)nnodes);
    if (pt_start[0] != 0 || pt_start[npts] != nnodes)
        return ctx->fail(FWI_EINVAL, "fwi_forward_spread: %s pt_start must run from 0 to the node count", what);
    for (int p = 0; p < npts; ++p) {
        const int a = pt_start[p], b = pt_start[p + 1];
        if (b < a || b - a > 8)
            return ctx->fail(FWI_EINVAL, "fwi_forward_spread: %s point %d has %d nodes (0..8 allowed)", what, p, b - a);
        for (int m = a; m < b; ++m) owner[m] = p;
    }
    const size_t need = (size_t)std::max(nnodes, npts + 1) * 8 + 16;
    if (sp.cap < need) {
        for (void **q : {&sp.pt_start, &sp.owner, &sp.weight}) {
            if (*q) HIPCHK(ctx, hipFree(*q));
            *q = nullptr;
            HIPCHK(ctx, hipMalloc(q, need));
        }
        sp.cap = need;
    }
    HIPCHK(ctx, hipMemcpyAsync(sp.pt_start, pt_start, (size_t)(npts + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(sp.owner, owner.data(), (size_t)nnodes * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(sp.weight, weight, (size_t)nnodes * ctx->esize, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // `owner` goes out of scope
    sp.npts = npts;
    return FWI_OK;
}

int fwi_forward_spread(fwi_ctx *ctx, int32_t nt, int32_t nsrc_pts, int32_t nsrc_nodes, const int32_t *src_idx,
                       const int32_t *src_pt_start, const void *src_weight, const void *wavelet, int32_t nrec_pts,
                       int32_t nrec_nodes, const int32_t *rec_idx, const int32_t *rec_pt_start, const void *rec_weight,
                       int32_t save, void *seis_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->have_model) return ctx->fail(FWI_ESTATE, "fwi_forward_spread: call fwi_set_model first");
    if (nt < 1 || nt > ctx->cfg.nt_max)
        return ctx->fail(FWI_EINVAL, "fwi_forward_spread: nt=%d outside [1, nt_max=%d]", nt, ctx->cfg.nt_max);
    if (nsrc_nodes < 0 || nrec_nodes < 0) return ctx->fail(FWI_EINVAL, "fwi_forward_spread: negative node count");
    if ((nsrc_nodes && (!src_idx || !wavelet)) || (nrec_nodes && (!rec_idx || !seis_out)))
        return ctx->fail(FWI_EINVAL, "fwi_forward_spread: null buffer");
    (void)hipSetDevice(ctx->cfg.device);
    int rc = set_spread(ctx, ctx->src_sp, nsrc_pts, nsrc_nodes, src_pt_start, src_weight, "source");
    if (!rc) rc = set_spread(ctx, ctx->rec_sp, nrec_pts, nrec_nodes, rec_pt_start, rec_weight, "receiver");
    if (rc) {
        ctx->src_sp.npts = ctx->rec_sp.npts = 0;
        return rc;
    }
    return DISPATCH(ctx,
                    Impl<float>::forward(ctx, nt, nsrc_nodes, src_idx, (const float *)wavelet, nrec_nodes, rec_idx, save,
                                         (float *)seis_out),
                    Impl<double>::forward(ctx, nt, nsrc_nodes, src_idx, (const double *)wavelet, nrec_nodes, rec_idx,
                                          save, (double *)seis_out));
}

int fwi_adjoint(fwi_ctx *ctx, const void *residual, int32_t image, void *adj_src_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->have_forward) return ctx->fail(FWI_ESTATE, "fwi_adjoint: no forward run to adjoin");
    if (image && !ctx->have_q)
        return ctx->fail(FWI_ESTATE, "fwi_adjoint: imaging needs fwi_forward(save=1) first");
    if (ctx->nrec && !residual && !ctx->have_dev_residual)
        return ctx->fail(FWI_EINVAL, "fwi_adjoint: null residual (and no fwi_misfit_l2 residual on the device)");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::adjoint(ctx, (const float *)residual, image, (float *)adj_src_out),
                    Impl<double>::adjoint(ctx, (const double *)residual, image, (double *)adj_src_out));
}

int fwi_misfit_l2(fwi_ctx *ctx, const void *d_obs, double *J_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->have_forward) return ctx->fail(FWI_ESTATE, "fwi_misfit_l2: no forward run whose data to compare");
    if (!J_out || (ctx->nrec && !d_obs)) return ctx->fail(FWI_EINVAL, "fwi_misfit_l2: null argument");
    if (!ctx->have_syn)
        return ctx->fail(FWI_ESTATE, "fwi_misfit_l2: the synthetics of the last forward are gone (an fwi_adjoint has "
                                     "run since): call it between fwi_forward and fwi_adjoint");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::misfit_l2(ctx, (const float *)d_obs, J_out),
                    Impl<double>::misfit_l2(ctx, (const double *)d_obs, J_out));
}

int fwi_gradient(fwi_ctx *ctx, int32_t wrt, void *g_out) {
    if (!ctx) return FWI_EINVAL;
    if (!g_out) return ctx->fail(FWI_EINVAL, "fwi_gradient: null output");
    if (wrt != FWI_WRT_VELOCITY && wrt != FWI_WRT_SLOWNESS2)
        return ctx->fail(FWI_EINVAL, "fwi_gradient: unknown parametrisation %d", wrt);
    if (!ctx->have_model) return ctx->fail(FWI_ESTATE, "fwi_gradient: no model set");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::gradient(ctx, wrt, (float *)g_out),
                    Impl<double>::gradient(ctx, wrt, (double *)g_out));
}

int fwi_gradient_reset(fwi_ctx *ctx) {
    if (!ctx) return FWI_EINVAL;
    (void)hipSetDevice(ctx->cfg.device);
    HIPCHK(ctx, hipMemsetAsync(ctx->g_acc, 0, (size_t)ctx->gd.npts * ctx->esize, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

int fwi_gradient_add(fwi_ctx *dst, fwi_ctx *src) {
    if (!dst || !src) return FWI_EINVAL;
    if (dst == src) return dst->fail(FWI_EINVAL, "fwi_gradient_add: dst and src are the same context");
    if (dst->cfg.device != src->cfg.device || dst->cfg.dtype != src->cfg.dtype || dst->gd.npts != src->gd.npts ||
        dst->gd.nx != src->gd.nx || dst->gd.nz != src->gd.nz)
        return dst->fail(FWI_EINVAL, "fwi_gradient_add: contexts differ in device, dtype or shape");
    (void)hipSetDevice(dst->cfg.device);
    HIPCHK(dst, hipStreamSynchronize(src->stream));  // src's adjoint sweeps are complete
    int rc = DISPATCH(dst, Impl<float>::vec_axpby(dst, dst->g_acc, 1.0, src->g_acc, 1.0),
                      Impl<double>::vec_axpby(dst, dst->g_acc, 1.0, src->g_acc, 1.0));
    if (rc) return rc;
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    return FWI_OK;
}

int fwi_dot(fwi_ctx *ctx, const void *a, const void *b, int64_t n, double *out) {
    if (!ctx) return FWI_EINVAL;
    if (!a || !b || !out || n < 0) return ctx->fail(FWI_EINVAL, "fwi_dot: bad argument");
    if (n == 0) {
        *out = 0.0;
        return FWI_OK;
    }
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::dot(ctx, (const float *)a, (const float *)b, n, out),
                    Impl<double>::dot(ctx, (const double *)a, (const double *)b, n, out));
}

static void *vec_slot(fwi_ctx *ctx, int32_t slot) {
    return (slot >= 0 && slot < (int32_t)ctx->vecs.size()) ? ctx->vecs[slot] : nullptr;
}

#define VEC_OR_FAIL(ctx, name, slot)                                                            \
    void *name = vec_slot(ctx, slot);                                                           \
    if (!name) return (ctx)->fail(FWI_EINVAL, "%s: vector slot %d does not exist", __func__, (int)(slot))

int fwi_vec_create(fwi_ctx *ctx, int32_t count) {
    if (!ctx) return FWI_EINVAL;
    if (count < 0 || count > 256) return ctx->fail(FWI_EINVAL, "fwi_vec_create: count must be in [0, 256]");
    (void)hipSetDevice(ctx->cfg.device);
    for (void *v : ctx->vecs)
        if (v) (void)hipFree(v);
    ctx->vecs.assign(count, nullptr);
    const size_t bytes = (size_t)ctx->gd.npts * ctx->esize;
    for (int i = 0; i < count; ++i) {
        HIPCHK(ctx, hipMalloc(&ctx->vecs[i], bytes));
        HIPCHK(ctx, hipMemsetAsync(ctx->vecs[i], 0, bytes, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

int fwi_vec_upload(fwi_ctx *ctx, int32_t slot, const void *host) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, v, slot);
    if (!host) return ctx->fail(FWI_EINVAL, "fwi_vec_upload: null buffer");
    (void)hipSetDevice(ctx->cfg.device);
    int rc = DISPATCH(ctx, Impl<float>::upload_compact(ctx, v, host), Impl<double>::upload_compact(ctx, v, host));
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

int fwi_vec_download(fwi_ctx *ctx, int32_t slot, void *host) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, v, slot);
    if (!host) return ctx->fail(FWI_EINVAL, "fwi_vec_download: null buffer");
    (void)hipSetDevice(ctx->cfg.device);
    int rc = DISPATCH(ctx, Impl<float>::download_compact(ctx, host, v), Impl<double>::download_compact(ctx, host, v));
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

int fwi_vec_copy(fwi_ctx *ctx, int32_t dst, int32_t src) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, d, dst);
    VEC_OR_FAIL(ctx, s, src);
    (void)hipSetDevice(ctx->cfg.device);
    if (d != s)
        HIPCHK(ctx, hipMemcpyAsync(d, s, (size_t)ctx->gd.npts * ctx->esize, hipMemcpyDeviceToDevice, ctx->stream));
    return FWI_OK;
}

int fwi_vec_axpby(fwi_ctx *ctx, int32_t y, double a, int32_t x, double b) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, vy, y);
    VEC_OR_FAIL(ctx, vx, x);
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::vec_axpby(ctx, vy, a, vx, b), Impl<double>::vec_axpby(ctx, vy, a, vx, b));
}

int fwi_vec_dot(fwi_ctx *ctx, int32_t x, int32_t y, double *out) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, vx, x);
    VEC_OR_FAIL(ctx, vy, y);
    if (!out) return ctx->fail(FWI_EINVAL, "fwi_vec_dot: null output");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::vec_dot(ctx, vx, vy, out), Impl<double>::vec_dot(ctx, vx, vy, out));
}

int fwi_vec_absmax(fwi_ctx *ctx, int32_t x, double *out) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, vx, x);
    if (!out) return ctx->fail(FWI_EINVAL, "fwi_vec_absmax: null output");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::vec_absmax(ctx, vx, out), Impl<double>::vec_absmax(ctx, vx, out));
}

int fwi_vec_clip(fwi_ctx *ctx, int32_t x, double lo, double hi) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, vx, x);
    if (!(lo <= hi)) return ctx->fail(FWI_EINVAL, "fwi_vec_clip: lo > hi");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::vec_clip(ctx, vx, lo, hi), Impl<double>::vec_clip(ctx, vx, lo, hi));
}

int fwi_set_model_vec(fwi_ctx *ctx, int32_t slot) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, v, slot);
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::set_model_vec(ctx, v), Impl<double>::set_model_vec(ctx, v));
}

int fwi_gradient_vec(fwi_ctx *ctx, int32_t wrt, int32_t slot) {
    if (!ctx) return FWI_EINVAL;
    VEC_OR_FAIL(ctx, v, slot);
    if (wrt != FWI_WRT_VELOCITY && wrt != FWI_WRT_SLOWNESS2)
        return ctx->fail(FWI_EINVAL, "fwi_gradient_vec: unknown parametrisation %d", wrt);
    if (!ctx->have_model) return ctx->fail(FWI_ESTATE, "fwi_gradient_vec: no model set");
    (void)hipSetDevice(ctx->cfg.device);
    return DISPATCH(ctx, Impl<float>::gradient_vec(ctx, wrt, v), Impl<double>::gradient_vec(ctx, wrt, v));
}

int fwi_comm_unique_id(void *id_out) {
    static_assert(sizeof(ncclUniqueId) == FWI_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return FWI_EINVAL;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return FWI_ECOMM;
    memcpy(id_out, &id, sizeof id);
    return FWI_OK;
}

int fwi_comm_init(fwi_ctx *ctx, int32_t rank, int32_t nranks, const void *id) {
    if (!ctx) return FWI_EINVAL;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks)
        return ctx->fail(FWI_EINVAL, "fwi_comm_init: bad rank %d of %d", rank, nranks);
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->comm) {
        (void)ncclCommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&ctx->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        ctx->nranks = 1;
        return ctx->fail(FWI_ECOMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, ncclGetErrorString(r));
    }
    ctx->nranks = nranks;
    return FWI_OK;
}

int fwi_comm_info(fwi_ctx *ctx, int32_t *nranks_out, int32_t *rank_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->comm) return ctx->fail(FWI_ESTATE, "fwi_comm_info: call fwi_comm_init first");
    int n = 0, r = -1;
    NCCLCHK(ctx, ncclCommCount(ctx->comm, &n));
    NCCLCHK(ctx, ncclCommUserRank(ctx->comm, &r));
    if (nranks_out) *nranks_out = n;
    if (rank_out) *rank_out = r;
    return FWI_OK;
}

int fwi_comm_abort(fwi_ctx *ctx) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->comm) return FWI_OK;
    (void)hipSetDevice(ctx->cfg.device);
    ncclComm_t c = ctx->comm;
    ctx->comm = nullptr;
    ctx->nranks = 1;
    NCCLCHK(ctx, ncclCommAbort(c));
    return FWI_OK;
}

int fwi_allreduce_gradient(fwi_ctx *ctx) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->comm) return ctx->fail(FWI_ESTATE, "fwi_allreduce_gradient: call fwi_comm_init first");
    (void)hipSetDevice(ctx->cfg.device);
    NCCLCHK(ctx, ncclAllReduce(ctx->g_acc, ctx->g_acc, (size_t)ctx->gd.npts,
                               ctx->cfg.dtype == FWI_F32 ? ncclFloat32 : ncclFloat64, ncclSum, ctx->comm,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

static int allreduce_scalars(fwi_ctx *ctx, double *vals, int32_t n, ncclRedOp_t op, const char *who) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->comm) return ctx->fail(FWI_ESTATE, "%s: call fwi_comm_init first", who);
    if (!vals || n < 1 || n > 8) return ctx->fail(FWI_EINVAL, "%s: n must be in [1, 8]", who);
    (void)hipSetDevice(ctx->cfg.device);
    HIPCHK(ctx, hipMemcpyAsync(ctx->red, vals, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, ncclAllReduce(ctx->red, ctx->red, (size_t)n, ncclFloat64, op, ctx->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(vals, ctx->red, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

int fwi_allreduce_f64(fwi_ctx *ctx, double *vals, int32_t n) {
    return allreduce_scalars(ctx, vals, n, ncclSum, "fwi_allreduce_f64");
}

int fwi_allreduce_f64_max(fwi_ctx *ctx, double *vals, int32_t n) {
    return allreduce_scalars(ctx, vals, n, ncclMax, "fwi_allreduce_f64_max");
}

int fwi_last_loop_ms(fwi_ctx *ctx, double *ms_out) {
    if (!ctx || !ms_out) return FWI_EINVAL;
    if (!ctx->have_loop_time) return ctx->fail(FWI_ESTATE, "fwi_last_loop_ms: no time loop has run");
    (void)hipSetDevice(ctx->cfg.device);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = ms;
    return FWI_OK;
}

int fwi_set_launch_mode(fwi_ctx *ctx, int32_t mode) {
    if (!ctx) return FWI_EINVAL;
    if (mode < FWI_LAUNCH_AUTO || mode > FWI_LAUNCH_GRAPH) return ctx->fail(FWI_EINVAL, "fwi_set_launch_mode: unknown mode %d", mode);
    ctx->graph_mode = mode == FWI_LAUNCH_GRAPH;  // (AUTO = stream, see fwi_create)
    return FWI_OK;
}

int fwi_check_padding(fwi_ctx *ctx, int64_t *dirty_out) {
    if (!ctx || !dirty_out) return FWI_EINVAL;
    (void)hipSetDevice(ctx->cfg.device);
    unsigned long long *cnt = (unsigned long long *)(ctx->red + 6);
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
    for (void *f : {ctx->u[0], ctx->u[1], ctx->C, ctx->vf, ctx->fwv, ctx->fx[0], ctx->fx[1], ctx->fwd[0], ctx->fwd[1],
                    ctx->fwx[0], ctx->fwx[1]}) {
        if (!f) continue;
        HIPCHK(ctx, DISPATCH(ctx, launch_count_dirty_padding<float>(ctx->gd, (const float *)f, cnt, ctx->stream),
                             launch_count_dirty_padding<double>(ctx->gd, (const double *)f, cnt, ctx->stream)));
    }
    unsigned long long n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, cnt, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *dirty_out = (int64_t)n;
    return FWI_OK;
}

int fwi_placement_info(fwi_ctx *ctx, double *us_before_out, double *us_after_out, int64_t *shift_bytes_out) {
    if (!ctx) return FWI_EINVAL;
    if (us_before_out) *us_before_out = ctx->place_us[0];
    if (us_after_out) *us_after_out = ctx->place_us[1];
    if (shift_bytes_out)
        for (int i = 0; i < 8; ++i) shift_bytes_out[i] = i < ctx->nmov ? (int64_t)ctx->mov_shift[i] : 0;
    return FWI_OK;
}

int fwi_last_host_ms(fwi_ctx *ctx, double *submit_ms_out, double *graph_build_ms_out) {
    if (!ctx) return FWI_EINVAL;
    if (!ctx->have_loop_time) return ctx->fail(FWI_ESTATE, "fwi_last_host_ms: no time loop has run");
    if (submit_ms_out) *submit_ms_out = ctx->host_submit_ms;
    if (graph_build_ms_out) *graph_build_ms_out = ctx->host_graph_ms;
    return FWI_OK;
}

int fwi_synchronize(fwi_ctx *ctx) {
    if (!ctx) return FWI_EINVAL;
    (void)hipSetDevice(ctx->cfg.device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FWI_OK;
}

}  // extern "C"

namespace fwi {
void set_global_error(const std::string &msg) { g_create_error = msg; }
}  // namespace fwi
