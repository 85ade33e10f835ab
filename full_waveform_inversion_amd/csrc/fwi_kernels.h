// Internal launch interface between the C-ABI (fwi_api.hip) and the gfx950
// kernels (fwi_kernels.hip).  Not part of the public boundary (include/fwi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace fwi {

// Layout halo of every padded field: 4 cells on each side of every stencil
// axis, whatever the order (4 floats = 16 B keeps interior rows float4-aligned).
constexpr int HALO = 4;
// Padded extents: x to XALIGN cells plus one shared halo (tight rows: slack between rows costs HBM efficiency, make_grid),
// y to YALIGN rows.
constexpr int XALIGN = 4;
constexpr int YALIGN = 16;
// Zero planes appended behind the far z halo of 3-D fields: the stream kernel reads up to
// r + PF (<= 4 + 3) planes past the last interior plane without clamping.
constexpr int LOOKAHEAD = 4;

// Padded field geometry.  Element (z, y, x) of the interior lives at
// off0 + z*sz + y*sy + x.  2-D grids are ny = 1 with no y halo (sz == sy).
struct GridDesc {
    int ndim, nz, ny, nx;
    int cx;          // row stride of COMPACT arrays: nx rounded up to 4, so that rows stay 16-byte
                     // aligned for any nx; the cx - nx pad columns hold zeros
    int r;           // stencil radius (order / 2)
    int64_t sy, sz;  // padded strides in elements (x stride is 1)
    int64_t off0;    // padded offset of interior point (0, 0, 0)
    int64_t ptot;    // elements of one padded field
    int64_t npts;    // nz * ny * cx: elements of one compact array (pad columns included)
};

// `xpitch_extra`: floats of slack added to the (tight) x pitch -- the A/B hook behind DESIGN.md's pitch sweep; fwi_create
// reads FWI_XPITCH_EXTRA once per process and passes it here (it is not consulted on any other path).
GridDesc make_grid(int ndim, int nz, int ny, int nx, int order, int xpitch_extra = 0);
// Widest x tile of any step kernel in elements (64 lanes x one float4): the tail behind the last row of a padded field
// covers the edge loads of such a tile whose row ends early (values never used).
constexpr int MAX_TILE_X = 256;

template <typename T>
struct StepArgs {
    const T *u_cur;  // u^n, padded, read with halo
    T *u_prev;       // in: u^{n-1}; out: u^{n+1} (same padded buffer)
    T *v;            // increment form (fwi_config.update_form = 1) only: in v^n = u^n - u^{n-1}, out v^{n+1}; the
                     // step is v' = A (B v + q), u' = u + v' -- algebraically the same recursion, but the rounding
                     // of u' is relative to u instead of to the cancelling 2u - u_prev (fp32: ~4x smaller error
                     // growth, at 20 instead of 16 B/update: v is read and written, u' written).  nullptr = standard
    const T *C;      // dt^2 c^2 a_1 / h^2, padded
    const T *dz, *dy, *dx;  // per-axis damping d = sigma dt / 2 (lengths nz, ny, nx)
    T *q_out;        // compact (npts) forward term of this step, or nullptr
    const T *q_in;   // compact forward term to correlate u_cur with, or nullptr
    const T *q_in2;  // second pairing taken in the same launch: u_prev (before it is overwritten) with
                     // this term, so the gradient accumulator is read-modified-written every other step
    int q_bf16;      // q_out / q_in / q_in2 point at bf16 arrays (fp32 stream kernel only; fwi_config.store_dtype)
    T *g;            // compact gradient accumulator (used with q_in)
    T ck[5];         // ck[k] = a_k / a_1, k = 1..r (ck[0] unused): the factor a_1 / h^2 is folded into C
    int damp;        // npml > 0
    int npml;        // border width; with dz_scale lets the stream kernel form dz[z] without a load
    T dz_scale;      // sigma_max dt / (2 npml^2): d_z(z) = dz_scale * dist(z)^2

    // Convolutional PML of the x border INSIDE the 3-D stream kernel (fp32, standard form; xp_mode 0 = off, 1 = the
    // forward recursion, 2 = its transpose): memory variables in the slab kernels' layout (nz, ny, 2 npml), 1-D
    // coefficients over x, first-difference weights dk (of a memory variable) / dk1 (of the field), see PmlArgs.
    int xp_mode;
    int xp_partial;  // npml % 4 != 0: the kernel variant with masked stores in the lane astride the border's inner edge
    T *xp_psi, *xp_zeta;
    const T *xp_a, *xp_b;
    T xp_dk[5], xp_dk1[5];
    // Convolutional PML of the z / y border (3-D): its term T_d = D psi_d' + zeta_d' (adjoint: E alpha_d - D beta_d) is
    // formed BEFORE the step by the line launch (fwi_pml.hip, pml_line_t: it depends on the newest field and the memory
    // variables only) and handed over in arrays compact over the axis' SHELL -- the border and the r cells its term
    // reaches: (shell rows of z, ny, cx) for z, (nz, shell rows of y, cx) for y; pml_shell_rows / pml_shell_index.  The
    // step kernel adds them inside q = C (L u + sum_d T_d + ...): no second pass over u' (v', q), no re-read of C.
    // nullptr = none.
    const T *pml_tz, *pml_ty;

    // Point operations fused into the step kernels (all nullptr / 0 = none).
    // Injection into u_next (and q_out): entries sorted by workgroup tile,
    // inj_start[tile] .. inj_start[tile + 1] is the tile's slice.
    const int *inj_start;
    const int64_t *inj_pidx, *inj_cidx;  // padded / compact flat index per entry
    const T *inj_cu, *inj_cq;            // coefficient of the amplitude in u / in q
    const int *inj_col;                  // column of the entry in the amplitude row
    const int *inj_run;                  // entries of this entry's node that follow from here (>= 1) if it is the first
                                         // of its node's run, 0 otherwise (entries of a node are consecutive)
    const T *inj_amp;                    // amplitude row of this time step
    // Sampling of u_cur (= the previous step's result) by extra workgroups.
    const int64_t *rec_pidx;
    T *rec_out;
    T rec_scale;
    int nrec;
};

enum StencilKernel { K_POINT = 1, K_STREAM = 2 };

struct StreamTuning {
    int ty;      // rows per workgroup (4 or 8)
    int zchunk;  // planes marched per workgroup
    int pf;      // planes fetched ahead of use (1..3)
    int tile_x;  // x extent of a tile in elements: at most 64 lanes x one 16-byte vector (256 fp32 /
                 // 128 fp64); the 3-D kernel takes narrower, equal tiles when nx is not a multiple of that
};

// Workgroup tile of the stream kernel that owns grid point (z, y, x): the index
// into StepArgs::inj_start.  Must match the kernel's block decode.
int stream_tile_of(const GridDesc &g, const StreamTuning &t, int z, int y, int x);
int stream_num_tiles(const GridDesc &g, const StreamTuning &t);

// Same for the point kernel (64 x 4 point workgroups).
int point_tile_of(const GridDesc &g, int z, int y, int x);
int point_num_tiles(const GridDesc &g);

// True when the stream kernel supports this grid / dtype.
bool stream_supported(const GridDesc &g, bool is_f32);
// `extra_bytes`: what a time step touches beside the three padded fields (the increment form's fourth field, the CPML's
// memory variables and handed-over terms): it decides whether the step is Infinity-Cache resident, i.e. which regime the
// launch shape is chosen for.
StreamTuning stream_default_tuning(const GridDesc &g, bool is_f32, double extra_bytes = 0.0);

template <typename T>
hipError_t launch_step(int kernel, const GridDesc &g, const StepArgs<T> &a, const StreamTuning &t,
                       hipStream_t s);
// The 3-D stream kernel's dispatch for one dtype and stencil radius (fwi_stream3d.h); instantiated in
// fwi_stream3d_*.hip, one translation unit per dtype / order group.
template <typename T, int R>
hipError_t launch_stream_r(const GridDesc &g, const StepArgs<T> &a, const StreamTuning &t, hipStream_t s);

// out[i] = u[pidx[i]] * scale
template <typename T>
hipError_t launch_record(const T *u, const int64_t *pidx, T *out, T scale, int n, hipStream_t s);
// g[i] += u(center of point i) * q[i] over the whole grid (last imaging step)
template <typename T>
hipError_t launch_image(const GridDesc &g, const T *u, const T *q, T *gacc, int q_bf16, hipStream_t s);
// bf16 store: gacc[cidx[s]] += cq[s] * inv_rs * sum_{n % stride == 0} adj_series[n, s] * wav[n, s]
template <typename T>
hipError_t launch_source_image(const T *adj_series, const T *wav, const int64_t *cidx, const T *cq, T *gacc, int nt,
                               int nsrc, int stride, T inv_rs, hipStream_t s);
// out[i] = gacc[i] * scale * (wrt_velocity ? -2 / c[i]^3 : 1); 0 in the pad columns
template <typename T>
hipError_t launch_finalize_gradient(const GridDesc &g, const T *gacc, const T *c, T *out, double scale,
                                    int wrt_velocity, hipStream_t s);
// compact (row stride cx, pad zeroed) <-> contiguous (row stride nx) copies on the device
template <typename T>
hipError_t launch_repack(const GridDesc &g, T *dst, const T *src, int to_compact, hipStream_t s);
// Cpad (padded, halo untouched) = dt2 * c^2 from the compact velocity c; *bad += #invalid entries
template <typename T>
hipError_t launch_build_model(const GridDesc &g, const T *c, T *Cpad, double dt2, int *bad, hipStream_t s);
// *bad += the number of cells of the padded field `f` OUTSIDE the grid's interior (halo planes / rows, the shared x halo
// between rows, the look-ahead planes and the tail) that are not exactly zero.  The tight row pitch rests on nothing ever
// writing there (make_grid); tests/test_gpu_round4.py checks it after every kind of sweep.
template <typename T>
hipError_t launch_count_dirty_padding(const GridDesc &g, const T *f, unsigned long long *bad, hipStream_t s);
// *out += sum a[i] * b[i]  (out must be zeroed by the caller)
template <typename T>
hipError_t launch_dot(const T *a, const T *b, int64_t n, double *out, hipStream_t s);

// off-grid points: per-point (nt, npts) <-> per-node (nt, nnodes) time series; a point's nodes are entries
// pt_start[p] .. pt_start[p + 1] (at most 8), owner[m] is the point of entry m, w[m] its interpolation weight
template <typename T>
hipError_t launch_scatter_series(const T *pt, T *node, const int *owner, const T *w, int nt, int npts, int nnodes,
                                 hipStream_t s);
template <typename T>
hipError_t launch_gather_series(const T *node, T *pt, const int *pt_start, const T *w, int nt, int npts, int nnodes,
                                hipStream_t s);
// obs_inout := syn - obs_inout; *out += sum of its squares (out zeroed by the caller)
template <typename T>
hipError_t launch_residual_l2(const T *syn, T *obs_inout, int64_t n, double *out, hipStream_t s);

// ---- convolutional PML (fwi_pml.hip): slab kernels around the undamped step kernels ------------------------
// The SHELL of an axis of n cells: its two borders and the r cells inward that their term reaches (the rows whose
// update takes a CPML term of that axis).  When the two shells meet (n < 2 npml + 3 r, the line kernel's one-segment
// case) the shell is the whole axis.  Rows are numbered low border first.
__host__ __device__ inline int pml_shell_rows(int n, int npml, int r) { return n < 2 * npml + 3 * r ? n : 2 * (npml + r); }
__host__ __device__ inline bool pml_in_shell(int j, int n, int npml, int r) {
    return n < 2 * npml + 3 * r || j < npml + r || j >= n - npml - r;
}
__host__ __device__ inline int pml_shell_index(int j, int n, int npml, int r) {  // (of a row in the shell)
    return (n < 2 * npml + 3 * r || j < npml + r) ? j : j - (n - 2 * (npml + r));
}

template <typename T>
struct PmlArgs {
    const T *u_cur;   // the newest field (padded): u^n forward, mu^{j+2} in the adjoint sweep
    T *u_next;        // the field the step kernel has just written (phase 3 adds the border term to it)
    T *v;             // increment form: v' of the step, or nullptr
    const T *C;       // dt^2 c^2, padded
    T *q_out;         // compact forward term of this step (phase 3 adds the border term), or nullptr
    T *psi[3], *zeta[3];    // memory variables per axis (z, y, x), compact over that axis' border; [1] unused in 2-D
    const T *a[3], *b[3];   // 1-D coefficients per axis
    T ck[5];          // second-difference weights a_k / a_1 (ck[0] = centre): C carries a_1 / h^2
    T dk[5];          // first-difference weights d_k / h, k = 1..r: D of a memory variable
    T dk1[5];         // d_k h / a_1: D of the field inside the memory-variable recursions (memory variables x h^2 / a_1)
    int npml;
    T *tz, *ty;       // line form: where the z / y border's term goes (StepArgs::pml_tz / pml_ty), or nullptr
};
// phase 1, 2: advance the memory variables (before the step kernel); 3: add their term to u' (after it)
// `axes`: bit d set = run axis d (z = 1, y = 2, x = 4); an axis the step kernel carries itself is left out
template <typename T>
hipError_t launch_pml(const GridDesc &g, const PmlArgs<T> &p, int phase, int reverse, hipStream_t s, int axes = 7);
// Line form of the z / y border (3-D): ONE launch for both axes BEFORE the step kernel advances the memory variables
// and writes the border's term into PmlArgs::tz / ty (a thread marches its line through the border with the field and
// the new memory variables in register windows); the step kernel adds the term inside q (StepArgs::pml_tz / pml_ty).
// pml_line_axes: the axes (z = 1, y = 2) it takes for this grid -- the slab phases are then run without them.
int pml_line_axes(const GridDesc &g, int npml);
template <typename T>
hipError_t launch_pml_lines(const GridDesc &g, const PmlArgs<T> &p, int reverse, hipStream_t s, int axes);
// True when the 3-D stream kernel can carry the x border's recursion in its lanes (see step3d_stream, XP)
bool stream_xpml_supported(const GridDesc &g, const StreamTuning &t, int npml, bool is_f32);

// ---- 2-D temporal blocking (fwi_fused2d.hip): FUSED2D_STEPS time steps per launch ----------------
constexpr int FUSED2D_STEPS = 4;   // time steps advanced per launch
constexpr int FUSED2D_TILE = 64;   // interior tile edge (points) of a grid that fills the chip with such tiles; extended
                                   // edge = TILE + 2 STEPS r.  Smaller grids take 32- or 16-point tiles (Fused2dArgs::ft)

struct Fused2dArgs {
    const float *u_cur, *u_prev, *C;   // padded inputs: u^n, u^{n-1}, dt^2 c^2
    float *out_cur, *out_prev;         // padded outputs: u^{n+K}, u^{n+K-1} (a different buffer pair)
    const float *dz, *dx;              // damping profiles
    float ck[5];
    int damp;
    int mode;                          // 0 plain, 1 store q (SAVE_Q), 2 imaging against stored q (IMAGE)
    float *q_base;                     // q of step n lives at q_base + n * npts (compact) ...
    int istride;                       // ... or, with istride > 1, for n % istride == 0 only, at slot n / istride
    float *g;                          // compact gradient accumulator (mode 2)
    int n0, dn;                        // first step index of the launch and +1 / -1
    int xcd_remap;                     // set by launch_fused2d: XCD-contiguous tile numbering
    int inc;                           // increment form: u_prev / out_prev are the v field in / out
    int skipd;                         // interior tiles skip the damped update: -1 = by tile count, 0 / 1 = forced
    int ft;                            // interior tile edge: 64, 32 or 16 (0 = 64); fused2d_pick_tile()
    const int *tile_order;             // tile of workgroup b (device array, fused2d_tile_order) or nullptr = the
                                       // XCD-contiguous row-major order computed in the kernel
    // injection entries, sorted by tile (CSR): every entry whose point lies in the tile's EXTENDED region
    const int *inj_start, *inj_lz, *inj_lx, *inj_col;
    const unsigned char *inj_interior;  // 1 if the point is in the tile's interior (then q gets its share)
    const int *inj_run;                 // run length at the first entry of a node's run, 0 at the others
    const int64_t *inj_cidx;
    const float *inj_cu, *inj_cq;
    const float *inj_amp;               // (nt, ninj) amplitudes, row n used by step n
    int ninj;
    // sampling entries, sorted by tile: points in the tile's interior
    const int *rec_start, *rec_lz, *rec_lx, *rec_col;
    float *rec_out;                     // (nt, nrec), row n written by step n
    float rec_scale;
    int nrec;
    // convolutional PML inside the launch (fwi_fused2d_pml.hip; npml = 0: off): memory variables of the z (index 0)
    // and x (index 1) border in the layout of the slab kernels (fwi_pml.hip), 1-D coefficients, first-difference
    // weights (dk: of a memory variable, dk1: of the field, see PmlArgs) and the direction of the recursion
    int pml_npml, pml_rev;
    float *pml_psi[2], *pml_zeta[2];
    // where the launch WRITES the advanced memory variables of the cells its tiles own (a second set of arrays, swapped
    // with the first by the host after the launch): a tile also reads border cells of its halo, which a neighbouring
    // tile owns and, in a launch of several rounds of workgroups, may already have advanced
    float *pml_psi_out[2], *pml_zeta_out[2];
    const float *pml_a[2], *pml_b[2];
    float pml_dk[5], pml_dk1[5];
};

// Tiling of an axis of n cells by WHOLE tiles of ft cells with one overlap seam in the middle -- the fused 2-D kernel
// with the CPML inside, whose border tiles must be whole (the recursion reaches 2 r cells per sub-step, so a border cell
// may not sit in a tile's halo): the lower half of the tiles starts at t * ft, the upper half is anchored at the high
// end (origins rounded up to a 16-byte group; the last tile may overhang n by <= 3 cells like any partial tile), and the
// two tiles either side of the seam overlap by nt * ft - n cells in the interior of the grid, which both compute and
// the lower one stores.  `seam` = 0: the plain tiling t * ft.  fused2d_own: the first cell of tile t that t owns.
__host__ __device__ inline int fused2d_origin(int t, int n, int ft, int seam) {
    const int nt = (n + ft - 1) / ft;
    return (!seam || nt < 2 || t < nt / 2) ? t * ft : ((n - (nt - t) * ft + 3) & ~3);
}
__host__ __device__ inline int fused2d_own(int t, int n, int ft, int seam) {
    return t == 0 ? 0 : fused2d_origin(t - 1, n, ft, seam) + ft;  // (= the tile's origin except just above the seam)
}
int fused2d_num_tiles(const GridDesc &g, int ft = FUSED2D_TILE);
// Tile edge that minimises rounds of workgroups x extended tile area (a 512^2 grid makes 64 tiles of 64^2 -- a
// quarter of the chip -- but 256 of 32^2); FWI_FUSED2D_TILE overrides (tuning / tests).
int fused2d_pick_tile(const GridDesc &g);
// Grids of more than one round of tiles: the order in which the workgroups take the tiles.  Tiles whose extended
// region reaches into the absorbing border do more work per sub-step (the damped update; with the CPML the border
// recursion, ~2x an interior tile): they go FIRST, dealt evenly over the XCDs, corners before edges, so that the launch
// does not end on them; the interior tiles follow in XCD-contiguous row-major runs as before.  Empty result = one round
// or less (the kernel's own numbering is kept).
void fused2d_tile_order(const GridDesc &g, int ft, int npml, std::vector<int> &order, int seam = 0);
hipError_t launch_fused2d(const GridDesc &g, const Fused2dArgs &a, hipStream_t s);
// True when the fused kernel can carry the CPML of this grid: every border cell a tile sees lies deep inside that
// tile's extended region or against the outside of the grid (conditions at fused2d_cpml_supported).
bool fused2d_cpml_supported(const GridDesc &g, int npml);
hipError_t launch_fused2d_cpml(const GridDesc &g, const Fused2dArgs &a, hipStream_t s);

// ---- 3-D temporal blocking (fwi_pair3d.hip): two time steps per pass, fp32, forward sweeps without imaging ----
constexpr int PAIR3D_TY = 8;   // interior rows (= waves) per workgroup
struct Pair3dInj {             // one source term as seen by one workgroup
    int z, yoff, xoff;         // plane; row relative to the tile's first interior row; column relative to lane 0's
    int col;                   // column of the amplitude rows
    float cu;                  // coefficient of the amplitude in u
    int interior;              // the point is one of the workgroup's own (then step 2 injects it too)
};
struct Pair3dArgs {
    const float *u_cur, *u_prev, *C;   // u^n, u^{n-1}, dt^2 c^2 (padded)
    float *out1, *out2;                // u^{n+1}, u^{n+2} (another buffer pair)
    const float *dy, *dx;              // damping profiles (the z one is formed from the plane index)
    float ck[5];
    int damp, npml;
    float dz_scale;
    const int *inj_start;              // CSR over workgroups (the kernel's renumbered order) into inj
    const Pair3dInj *inj;
    const float *inj_amp0, *inj_amp1;  // amplitude rows of step n and n + 1
    const int64_t *rec_pidx;           // sampling of the previous pass' two fields (u_prev -> out0, u_cur -> out1)
    float *rec_out0, *rec_out1;
    float rec_scale;
    int nrec;
};
void pair3d_default_tuning(const GridDesc &g, int *zchunk, int *tw);
int pair3d_num_tiles(const GridDesc &g, int zchunk, int tw);
int pair3d_tile_of(const GridDesc &g, int zchunk, int tw, int z, int y, int x);
hipError_t launch_pair3d(const GridDesc &g, const Pair3dArgs &a, int zchunk, int tw, hipStream_t s);

// optimiser vector algebra: y = a x + b y; clamp; *out = max(*out, max|x|) (out zeroed by the caller)
template <typename T>
hipError_t launch_axpby(T *y, double a, const T *x, double b, int64_t n, hipStream_t s);
// clamps the grid points only (pad columns of a compact array stay zero)
template <typename T>
hipError_t launch_clip(const GridDesc &g, T *x, double lo, double hi, hipStream_t s);
template <typename T>
hipError_t launch_absmax(const T *x, int64_t n, double *out, hipStream_t s);

}  // namespace fwi
