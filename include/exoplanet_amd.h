/* exoplanet_amd.h -- C ABI of libexoplanet_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-leapfrog-step log-likelihood hot path of
 * exoplanet-dev/exoplanet.  Every entry point replaces one third-party Op that
 * the reference reaches through `exoplanet.compat.ops`
 * (/root/reference/src/exoplanet/compat.py:27,56) or through celerite2, or one
 * block of elementwise PyTensor graph between those Ops.  The binding a
 * maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), float64, C-contiguous (the column-form packing entry points take
 *     HOST arrays of device pointers and say so); ordinary hipMalloc memory (coarse-grained): the summed flux of
 *     several planets and the cotangents of timing shifts are accumulated with the hardware's fp64 atomics, which
 *     fine-grained / host-mapped allocations do not offer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     calls are asynchronous and stream-ordered, nothing is allocated inside;
 *   - return value: 0 = launched, EXO_ERR_* otherwise (no exceptions cross the
 *     boundary); numeric failure is in-band (NaN in -> NaN out, `flag` words);
 *   - the library keeps no global state (no caches, no environment reads): re-entrant,
 *     one process per GPU.
 */
#ifndef EXOPLANET_AMD_H
#define EXOPLANET_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXO_OK 0
#define EXO_ERR_INVALID_ARGUMENT 1
#define EXO_ERR_LAUNCH 2
#define EXO_ERR_WORKSPACE 3

/* ABI version, bumped whenever a signature, a layout or the set of flags below changes.
 * 10: EXO_FLAG_CADENCE_MAJOR, EXO_FLAG_SORTED_TIMES, the *_cm_f64 entry points, exo_transit_flux_fwd_jac_f64 / _jac_vjp_f64,
 *     exo_transit_sparse_scatter_f64, exo_sho_coefficients_multi_*; a larger exo_celerite_state_doubles (the lane order of
 *     batches of mixed pair kinds lives in the state: forward and reverse call must be the same build); flag bits a build
 *     does not know are EXO_ERR_INVALID_ARGUMENT.
 * 11: the sparse model -- exo_sparse_model, exo_transit_flux_sparse_model, exo_transit_flux_vjp_sparse_f64,
 *     exo_celerite_loglike_sparse_{fwd,vjp}_f64, exo_celerite_default_chunks; EXO_FLAG_SPARSE accepted by the Jacobian pair;
 *     exo_transit_flux_cols_vjp_f64; EXO_GP_MAX_J 16; a larger exo_transit_flux_workspace_bytes.
 * 12: exo_sparse_model.row_of_draw covers EVERY per-draw array of a sparse celerite call (coefficients, pair kinds, per-draw diag,
 *     loglike, gloglike, the cotangents written back: all in the caller's order, nothing to permute); exo_sparse_model_order.
 * 13: the MERGED sparse model -- exo_sparse_merge_workspace_bytes, exo_sparse_merge_layout, exo_sparse_model_merge_f64,
 *     exo_sparse_model_merged, exo_sparse_model_merge_vjp_f64: several lists per draw (planets, occultations) as one.
 * 14: EXO_GP_PREPARE_ADJOINT, a flag or-ed into n_chunks of a celerite pair (the adjoint scan beside the forward chunk kernel). */
#define EXO_ABI_VERSION 14
int32_t exo_abi_version(void);

/* ---------------------------------------------------------------------------
 * ops.kepler(M, ecc) -> (sinf, cosf)
 * replaces exoplanet_core's Kepler Op; reference call sites
 *   src/exoplanet/orbits/keplerian.py:333  (ecc pre-broadcast to M's shape)
 *   src/exoplanet/orbits/keplerian.py:818
 * ecc outside [0,1) -> NaN (docstring keplerian.py:58).
 * ------------------------------------------------------------------------- */
int exo_kepler_f64(const double* M, const double* ecc, double* sinf, double* cosf,
                   int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * ops.quad_solution_vector(b, r) -> s[..., 3]   (+ ds/db, ds/dr, nullable)
 * replaces exoplanet_core's quad_solution_vector Op; reference call site
 *   src/exoplanet/light_curves/limb_dark.py:24
 * Takes |b| (tests/light_curves_test.py:24-27 feed b < 0); dsdb carries sign(b).
 * s, dsdb, dsdr are [n][3].  Pass dsdb = dsdr = NULL for value only.
 * ------------------------------------------------------------------------- */
int exo_quad_solution_vector_f64(const double* b, const double* r, double* s,
                                 double* dsdb, double* dsdr, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * ops.contact_points(a, e, cosw, sinw, cosi, sini, L) -> (M_left, M_right, flag)
 * replaces exoplanet_core's contact_points Op; reference call site
 *   src/exoplanet/orbits/keplerian.py:744-753
 * flag != 0 means "no contact found"; the caller then evaluates every cadence
 * (keplerian.py:771-775).
 * ------------------------------------------------------------------------- */
int exo_contact_points_f64(const double* a, const double* e, const double* cosw,
                           const double* sinw, const double* cosi, const double* sini,
                           const double* L, double* M_left, double* M_right,
                           int32_t* flag, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * Fused transit flux: everything between the time array and the flux array of
 *   LimbDarkLightCurve.get_light_curve   (src/exoplanet/light_curves/limb_dark.py:99-232)
 *   KeplerianOrbit._get_true_anomaly / _get_position / _rotate_vector
 *                                        (src/exoplanet/orbits/keplerian.py:283-334,380-409)
 *   KeplerianOrbit.in_transit mask       (keplerian.py:729-731,765-769)
 *   SecondaryEclipseLightCurve blend     (src/exoplanet/light_curves/secondary_eclipse.py:45-70)
 * for `n_draw` independent parameter sets (posterior draws / chains) at once.
 *
 * Per-(draw, planet) parameter record, EXO_NPAR doubles, all lengths in units
 * of the stellar radius (the O(P) algebra that produces them stays on the host
 * side, in autograd):
 * ------------------------------------------------------------------------- */
#define EXO_NPAR 20
#define EXO_P_N 0       /* mean motion 2 pi / period            keplerian.py:146     */
#define EXO_P_TP 1      /* t_periastron = t0 - M0/n             keplerian.py:277     */
#define EXO_P_ECC 2     /* eccentricity (0 for ecc=None)                              */
#define EXO_P_COSW 3    /* cos(omega)   (1 for ecc=None)        keplerian.py:303-308 */
#define EXO_P_SINW 4    /* sin(omega)   (0 for ecc=None)                              */
#define EXO_P_COSI 5    /* cos(incl)                            keplerian.py:227,312 */
#define EXO_P_SINI 6    /* sin(incl)  (only its sign is used: los > 0 test)          */
#define EXO_P_AOR 7     /* a / R_star                                                 */
#define EXO_P_ROR 8     /* r_planet / R_star                                          */
#define EXO_P_T0 9      /* reference transit time (window phase) keplerian.py:731    */
#define EXO_P_PERIOD 10 /* period                                                     */
#define EXO_P_TS 11     /* first contact  - t0 (<= 0), without texp keplerian.py:755-762 */
#define EXO_P_TE 12     /* fourth contact - t0 (>= 0)                                 */
#define EXO_P_FRATIO 13 /* secondary: sbr * ror^2                secondary_eclipse.py:68 */
#define EXO_P_TS2 14    /* secondary-eclipse window start - t0 (may exceed +-P/2)    */
#define EXO_P_TE2 15    /* secondary-eclipse window end   - t0                        */
#define EXO_P_CLIGHT 16 /* speed of light in stellar radii per day (constants.py:36 / R_star): only read
                           with EXO_FLAG_LIGHT_DELAY                                  */
/* slots 17 .. 19: reserved, written as 0 by the packing kernel, not read */

/* flags */
#define EXO_FLAG_PER_PLANET 1u /* flux is [n_draw][n_cad][n_planet] instead of summed [n_draw][n_cad] */
#define EXO_FLAG_WINDOW 2u     /* skip cadences outside [TS,TE] (+- texp/2): use_in_transit semantics */
#define EXO_FLAG_SECONDARY 4u  /* also evaluate the occultation of the planet; ld is [n_draw][6]      */
/* 8u is EXO_PACK_CIRCULAR (record packing only) */
#define EXO_FLAG_EXACT_SCAN 16u /* classify cadences with the fp64 Kepler solve instead of the conservative
                                   fp32 pre-filter.  Results are identical either way (accepted cadences are
                                   always evaluated in fp64); the flag exists to measure / verify that.      */

#define EXO_FLAG_SPARSE 32u     /* run-enumeration sweeps only (sorted t, scalar or no texp, no timing tables, no
                                   EXO_FLAG_EXACT_SCAN; EXO_ERR_INVALID_ARGUMENT otherwise): the flux array is NOT
                                   touched (pass NULL); the output is the runs of cadences in which a planet can
                                   overlap the disk and a compact value array, both inside `workspace`
                                   (exo_transit_flux_sparse_layout); every other cadence has flux exactly 0.  */

#define EXO_FLAG_LIGHT_DELAY 64u /* light-travel delay (keplerian.py:411-470, _get_retarded_position with z0 = 0):
                                   a body is seen where it was at t - delay, the delay from its line-of-sight
                                   position, velocity and acceleration at t; a second Kepler solve per sample, in
                                   the same kernel.  Needs slot EXO_P_CLIGHT and the VALUE of EXO_P_SINI (both get
                                   cotangents then).  Run-enumeration sweeps only (see EXO_FLAG_SPARSE).        */

#define EXO_FLAG_CADENCE_MAJOR 128u /* the dense SUMMED flux arrays of the call -- flux (out), gflux (in) -- are
                                      [n_cad][n_draw] instead of [n_draw][n_cad]: the layout the celerite kernels read
                                      with every wave's accesses contiguous (exo_celerite_loglike_obs_*_cm_f64).
                                      Run-enumeration sweeps only (see EXO_FLAG_SPARSE); not with
                                      EXO_FLAG_PER_PLANET (EXO_ERR_INVALID_ARGUMENT).                            */

#define EXO_FLAG_SORTED_TIMES 256u /* the caller has CHECKED that t is non-decreasing (the sweep otherwise checks it on
                                     the device, every call, in a launch of its own before the searches): windows and runs
                                     then come out of one launch.  The word is taken for NEIGHBOURING cadences only: the
                                     launch still looks at 65 evenly spaced cadences, and a series that does not ascend
                                     through them (a NaN included) is swept cadence by cadence, as without the flag.
                                     Disorder finer than that under this flag gives undefined results; without the flag
                                     unsorted times are merely slow (every cadence solved).  A flag is part of a captured
                                     launch: it must hold for whatever the time buffer contains at every replay.       */

/* a sweep called with a flag bit outside this set returns EXO_ERR_INVALID_ARGUMENT */
#define EXO_FLAG_SWEEP_ALL (EXO_FLAG_PER_PLANET | EXO_FLAG_WINDOW | EXO_FLAG_SECONDARY | EXO_FLAG_EXACT_SCAN | \
                            EXO_FLAG_SPARSE | EXO_FLAG_LIGHT_DELAY | EXO_FLAG_CADENCE_MAJOR | EXO_FLAG_SORTED_TIMES)

#define EXO_MAX_PLANETS 16
#define EXO_MAX_SUBEXP 63

/* Forward.
 *   t          [n_cad]                  shared by all draws
 *   texp       NULL (n_texp = 0), scalar (n_texp = 1) or per cadence (n_texp = n_cad)
 *   stencil_dt [n_sub], stencil_w [n_sub]   exposure stencil in units of texp
 *                                       (limb_dark.py:181-197); n_sub = 1 if no texp
 *   params     [n_draw][n_planet][EXO_NPAR]
 *   ld         [n_draw][3] Green's-basis coefficients c of get_cl (limb_dark.py:11-18),
 *              or [n_draw][6] = primary c then secondary c with EXO_FLAG_SECONDARY
 *   flux       out, see EXO_FLAG_PER_PLANET
 *   workspace  exo_transit_flux_workspace_bytes() bytes of scratch                 */
int exo_transit_flux_fwd_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                             const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                             const double* params, const double* ld, int64_t n_draw,
                             int32_t n_planet, uint32_t flags, double* flux, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Bytes of scratch the fused entry points need (active-cadence lists of the scan
 * kernel + the deterministic two-stage gradient reduction).                     */
int64_t exo_transit_flux_workspace_bytes(int64_t n_cad, int64_t n_draw, int32_t n_planet);

/* Where the sparse output of an EXO_FLAG_SPARSE sweep lives inside `workspace` (same n_cad, n_draw,
 * n_planet as the sweep).  out[0..4] = byte offsets of nrun, runs, pre_all, vals, and r_max:
 *   list = (draw * n_planet + planet) * n_ev + event   (n_ev = 2 with EXO_FLAG_SECONDARY: 0 transits,
 *                                                       1 occultations; else n_ev = 1)
 *   nrun    int32 [n_list]                 windows of the list that reach into the series
 *   runs    int32 [n_list][r_max][4]       (lo, a, b, hi): cadences [lo, hi) of window k
 *   pre_all int32 [n_list][r_max + 1]      exclusive prefix sums of hi - lo
 *   vals    double [n_draw][n_planet][n_cad]  flux of cadence i of window k of (draw, planet):
 *           vals[draw][planet][(event ? pre_all[list of event 0][nrun] : 0) + pre_all[list][k] + i - lo]
 * (per planet, EXO_FLAG_PER_PLANET or not).  If t is not sorted, or a window cannot be bounded, the
 * list is the whole series cut into a few runs.                                                   */
int exo_transit_flux_sparse_layout(int64_t n_cad, int64_t n_draw, int32_t n_planet, int64_t* out);

/* The sparse output as a MODEL for the celerite entries (exo_celerite_loglike_sparse_*_f64, which see for the layout): segments
 * of cadences + their values, per draw.  exo_transit_flux_sparse_model fills the descriptor with pointers into `workspace`
 * (that of an EXO_FLAG_SPARSE sweep with the same n_cad, n_draw, n_planet; flags: EXO_FLAG_SECONDARY as the sweep had it).
 * One list per draw (one planet, no occultations): the runs are the segments.  Otherwise EXO_ERR_INVALID_ARGUMENT -- callers
 * merge the lists first (exo_sparse_model_merge_f64 below) or keep the dense model.                                                                                      */
typedef struct exo_sparse_model {
  const int32_t* nseg;
  const int32_t* seg;
  const int32_t* off;
  const double* vals;
  int64_t seg_row, off_row, val_row;
  int32_t seg_step, hi_at;
  const int32_t* row_of_draw;   /* NULL, or [n_draw], a permutation: the ORDER in which the celerite kernels take the draws -- their
                                   draw d (a lane; 64 consecutive ones a wave) is row row_of_draw[d] of nseg / seg / off / vals / gvals
                                   AND of every other per-draw array of the call: coefficients, pair kinds, a per-draw diag, loglike,
                                   gloglike, gcoef_*, gdiag, gdiag_sum.  All arrays stay in the caller's order; nothing is permuted
                                   (ABI 12; ABI 11 applied it to the model alone).  Why: a wave pays for a transit while ANY of its 64
                                   draws is inside one; taking the draws sorted by transit timing (exo_sparse_model_order) keeps a
                                   wave's transits together -- C3: 3.6 -> 3.3 ms of GP kernels.  exo_transit_flux_sparse_model sets it
                                   to NULL.                                                                                     */
} exo_sparse_model;
int exo_transit_flux_sparse_model(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw,
                                  int32_t n_planet, uint32_t flags, exo_sparse_model* out);
/* The MERGED model (ABI 13): what exo_transit_flux_sparse_model refuses.  A draw with several lists -- n_planet planets, and with
 * EXO_FLAG_SECONDARY a transit and an occultation list each -- becomes ONE ascending list of disjoint segments: the union of
 * the lists' runs, a cadence's value the SUM of the values the lists hold for it (the reference sums the planets per cadence,
 * limb_dark.py:228-230; a planet's transit and occultation never share a cadence, secondary_eclipse.py:67-70).  It lives in a
 * caller-owned `merge_ws` (exo_sparse_merge_workspace_bytes; layout: exo_sparse_merge_layout out[0..4] = byte offsets of
 * nseg int32 [n_draw], seg int32 [n_draw][cap][2] = (lo, hi), off int32 [n_draw][cap + 1], vals double [n_draw][n_cad], and cap):
 *   exo_sparse_model_merge_f64       two launches on `stream` (segments: a block per draw, every run ranked among the draw's
 *                                    runs by binary search, a prefix-maximum scan of the run ends; values: a thread per merged
 *                                    cadence); `workspace` = the EXO_FLAG_SPARSE sweep's, same n_cad / n_draw / n_planet / flags;
 *                                    fills `out` (may be NULL)
 *   exo_sparse_model_merged          the descriptor alone (no launch): for the reverse call
 *   exo_sparse_model_merge_vjp_f64   gvals (the sweep's value layout: what exo_transit_flux_vjp_sparse_f64 takes) from gmvals
 *                                    [n_draw][n_cad], the cotangent the celerite reverse entry wrote for the merged values
 * n_cad < 2^31, n_draw <= 65535.  Deterministic (no atomics): a merged value is summed in list order.                        */
int64_t exo_sparse_merge_workspace_bytes(int64_t n_cad, int64_t n_draw, int32_t n_planet);
int exo_sparse_merge_layout(int64_t n_cad, int64_t n_draw, int32_t n_planet, int64_t* out);
int exo_sparse_model_merge_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                               uint32_t flags, void* merge_ws, int64_t merge_ws_bytes, exo_sparse_model* out, void* stream);
int exo_sparse_model_merged(const void* merge_ws, int64_t merge_ws_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                            exo_sparse_model* out);
int exo_sparse_model_merge_vjp_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                                   uint32_t flags, const void* merge_ws, int64_t merge_ws_bytes, const double* gmvals,
                                   double* gvals, void* stream);
/* order [n_draw] (device, int32): the draws of `model` (its rows 0 .. n_draw - 1; row_of_draw ignored) by ascending mean spacing of
 * their segments -- the period, in cadences: neighbouring periods keep their transits together all along the series -- ties by the
 * start of the first segment, then by index: what to pass as row_of_draw.  One launch, no host synchronisation.  n_draw <=
 * EXO_SPARSE_ORDER_MAX_DRAWS (EXO_ERR_INVALID_ARGUMENT beyond: sort on the host side, or pass NULL).                          */
#define EXO_SPARSE_ORDER_MAX_DRAWS 4096
int exo_sparse_model_order(const exo_sparse_model* model, int64_t n_draw, int32_t* order, void* stream);
/* Reverse sweep for a cotangent given in the VALUE layout of the sparse output -- gvals [n_draw][n_planet][n_cad], gvals of
 * (draw, planet) at the positions of that planet's values: what exo_celerite_loglike_sparse_vjp_f64 writes -- instead of a
 * dense gflux.  flags must carry EXO_FLAG_SPARSE (and whatever the forward sweep carried); otherwise as
 * exo_transit_flux_vjp_f64 without flux_out.  reuse_runs != 0: `workspace` is the forward sweep's, untouched since, for these
 * very t / params (the windows and runs in it are reused: no enumeration launch).                                       */
int exo_transit_flux_vjp_sparse_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                    const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                    const double* params, const double* ld, int64_t n_draw, int32_t n_planet,
                                    uint32_t flags, const double* gvals, double* gparams, double* gld, double* flux_dot,
                                    void* workspace, int64_t workspace_bytes, int32_t reuse_runs, void* stream);

/* A light curve whose cotangent is not known yet (the mean of a GP: limb_dark.py:99-232 feeding celerite2's
 * log_likelihood; the cotangent is the GP's gradient with respect to its mean): the forward sweep and, once the cotangent
 * exists, its VJP WITHOUT a second sweep.  exo_transit_flux_fwd_jac_f64 is exo_transit_flux_fwd_f64 (same arguments, same flux,
 * bit for bit) that also leaves, for every solved cadence, the sixteen derivatives of its flux -- ten record slots, six
 * limb-darkening coefficients -- in `jac` (exo_transit_flux_jac_doubles(n_cad, n_draw, n_planet) doubles, caller-owned);
 * exo_transit_flux_jac_vjp_f64 contracts them with the cotangent (rows or EXO_FLAG_CADENCE_MAJOR, as the flux was) into what
 * exo_transit_flux_vjp_f64 returns.  `workspace` of the second call is the forward call's, untouched in between (it holds
 * the runs, the values and their cadences).  Run-enumeration sweeps (see EXO_FLAG_SPARSE) of the summed flux, without timing
 * tables, EXO_FLAG_LIGHT_DELAY or EXO_FLAG_PER_PLANET: EXO_ERR_INVALID_ARGUMENT otherwise -- callers keep
 * the two-sweep route for those.  With EXO_FLAG_SPARSE (both calls): flux may be NULL and is not written, and gflux is
 * the cotangent in the value layout (see exo_transit_flux_vjp_sparse_f64).  Worth it when a cadence is several samples (an exposure stencil: each of them a Kepler
 * solve, the row still sixteen doubles); bit-reproducible.                                                              */
int64_t exo_transit_flux_jac_doubles(int64_t n_cad, int64_t n_draw, int32_t n_planet);
int exo_transit_flux_fwd_jac_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                                 const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                                 int64_t n_draw, int32_t n_planet, uint32_t flags, double* flux, double* jac,
                                 int64_t jac_doubles, void* workspace, int64_t workspace_bytes, void* stream);
int exo_transit_flux_jac_vjp_f64(const double* gflux, int64_t n_cad, int64_t n_draw, int32_t n_planet, uint32_t flags,
                                 const double* jac, void* workspace, int64_t workspace_bytes, double* gparams, double* gld,
                                 double* flux_dot, void* stream);

/* White-noise Gaussian likelihood of ONE observed series against the light curves of n_draw parameter sets, value and
 * gradient, without a dense flux array -- what a sampler step needs when there is no correlated-noise model (the
 * reference's `pm.Normal("obs", mu=light_curve, sigma=yerr, observed=y)`, docs/tutorials: the (draw, cadence) arrays of
 * flux, residual and their cotangents never exist).  With f[d][n] the summed flux of draw d (exactly 0 outside the runs
 * of the sparse output, see EXO_FLAG_SPARSE), w_n = ivar[n_ivar == 1 ? 0 : n]:
 *   chi2[d]   = sum over the cadences n solved for draw d of  w_n ((f[d][n] - obs[n])^2 - obs[n]^2)
 *             = sum_n w_n (f[d][n] - obs[n])^2  -  sum_n w_n obs[n]^2      (the second sum is the caller's constant)
 *   gparams, gld = d chi2[d] / d (params, ld)
 * One planet without occultation or exposure stencil: ONE evaluation per solved cadence (the cotangent 2 w (f - obs) is
 * formed inside it).  Otherwise three sweeps' worth of launches in one call: values into the sparse output, residuals and
 * cotangents on it (a draw's planets may transit at once: every value sees the draw's total flux at its cadence),
 * gradient sweep.  Sorted t, one
 * exposure time (or none), no timing tables (EXO_ERR_INVALID_ARGUMENT otherwise); flags: EXO_FLAG_SECONDARY,
 * EXO_FLAG_WINDOW, EXO_FLAG_LIGHT_DELAY.  Bit-reproducible.                                                          */
int exo_transit_chi2_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                             const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                             int64_t n_draw, int32_t n_planet, uint32_t flags, const double* obs, const double* ivar,
                             int64_t n_ivar, double* chi2, double* gparams, double* gld, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* The same likelihood for an orbit with transit-timing variations (tables as for exo_transit_flux_ttv_vjp_f64, which
 * see; transits only, no light delay: EXO_FLAG_WINDOW is the one flag), plus
 *   gshift [n_draw][n_planet][n_edge + 1] = d chi2[d] / d ttv_shift
 * -- the gradient of a timing fit (TTVOrbit with `ttvs` or `transit_times` as parameters: ttv.py:71-156) without a
 * (draw, cadence) array.  n_cad >= 1.  One planet and no exposure stencil: one evaluation per solved cadence. */
int exo_transit_chi2_ttv_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp, const double* stencil_dt,
                                 const double* stencil_w, int32_t n_sub, const double* params, const double* ld,
                                 int64_t n_draw, int32_t n_planet, uint32_t flags, const double* ttv_edges,
                                 const double* ttv_shift, int32_t n_edge, const double* obs, const double* ivar,
                                 int64_t n_ivar, double* chi2, double* gparams, double* gld, double* gshift, void* workspace,
                                 int64_t workspace_bytes, void* stream);

/* Diagnostic for the scan kernel's conservative fp32 cadence classifier: the fp32
 * estimate of (cos E - e, sqrt(1-e^2) sin E) for mean anomaly M (fp64 phase) and
 * eccentricity ecc, widened back to double.  The classifier's safety margin assumes
 * |error| <= 8e-4; tests/test_gpu_scan_filter.py checks that here.                 */
int exo_selftest_orbit_pos_f32(const double* M, const double* ecc, double* cx, double* sx, int64_t n,
                               void* stream);

/* Profiling hook shared by the fused entry points below: if `ev_start` /
 * `ev_stop` (hipEvent_t passed as void*, may be NULL) are given, they are
 * recorded on `stream` immediately before the first / after the last kernel of the
 * sweep (scan + heavy [+ reduce]; the small memset of the gradient buffer stays
 * outside), so a caller can time the kernels alone without a profiler attached. */
int exo_transit_flux_fwd_ev_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                const double* params, const double* ld, int64_t n_draw,
                                int32_t n_planet, uint32_t flags, double* flux, void* workspace,
                                int64_t workspace_bytes, void* stream, void* ev_start, void* ev_stop);

/* Reverse (recompute-forward): given gflux (same shape as flux) accumulate
 *   gparams [n_draw][n_planet][EXO_NPAR]  (slots T0, PERIOD, TS.. are 0; SINI, CLIGHT too without EXO_FLAG_LIGHT_DELAY)
 *   gld     [n_draw][3 or 6]
 * and, if flux_out != NULL, also write the forward value in the same pass
 * (value + gradient in one sweep over t: 24 B per (draw, cadence)); if
 * flux_dot != NULL, flux_dot[n_draw] receives sum_n gflux * flux per draw (the
 * scalar whose gradient gparams / gld are).                                    */
int exo_transit_flux_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                             const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                             const double* params, const double* ld, int64_t n_draw,
                             int32_t n_planet, uint32_t flags, const double* gflux,
                             double* flux_out, double* gparams, double* gld, double* flux_dot,
                             void* workspace, int64_t workspace_bytes, void* stream);
int exo_transit_flux_vjp_ev_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                const double* params, const double* ld, int64_t n_draw,
                                int32_t n_planet, uint32_t flags, const double* gflux,
                                double* flux_out, double* gparams, double* gld, double* flux_dot,
                                void* workspace, int64_t workspace_bytes, void* stream, void* ev_start,
                                void* ev_stop);

/* The whole user-level step of a standard-parameterisation orbit in one call (round 5): the constructor's COLUMNS in -- as
 * exo_pack_records_cols_f64, which see below: cols / draw_stride / planet_stride / defaults / ld_cols / ld_draw_stride are HOST
 * arrays of device pointers, strides and defaults; pack_flags: EXO_PACK_CIRCULAR, EXO_FLAG_WINDOW, EXO_FLAG_SECONDARY (as in
 * flags) -- the records (params, ld: written, the caller's buffers), the value + VJP sweep of exo_transit_flux_vjp_f64 (flux_out
 * nullable or, with EXO_FLAG_SPARSE, unused; gparams, gld, flux_dot out), and, fold != 0, the packing VJP behind it:
 *   gcols[k] [n_draw][n_planet], gld_cols[k] [n_draw]  (HOST arrays of device pointers, NULL entry = not wanted)
 *                = gscale[d] (or 1, gscale == NULL) x d sum_n gflux flux / d column
 * On a run-enumeration sweep with EXO_FLAG_SORTED_TIMES the packing rides on the windows + enumeration launch (every list's wave
 * packs its own record: no packing launch in front); the packing VJP is a launch of its own behind the sweep (folded into the
 * sweep's last kernel it was measured slower: one thread's serial chain at the tail of every block).  Anything else runs the three
 * calls one after the other: same results, bit for bit.  ev_start / ev_stop as exo_transit_flux_vjp_ev_f64.  Reference:
 * keplerian.py:75-281 + limb_dark.py:99-232 and their gradients, one likelihood-gradient evaluation of a sampler.      */
int exo_transit_flux_cols_vjp_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                                  const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                                  uint32_t pack_flags, const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                  const double* stencil_dt, const double* stencil_w, int32_t n_sub, int64_t n_draw,
                                  int32_t n_planet, uint32_t flags, const double* gflux, double* flux_out, double* params,
                                  double* ld, double* gparams, double* gld, double* flux_dot, int32_t fold,
                                  const double* gscale, double* const* gcols, double* const* gld_cols, void* workspace,
                                  int64_t workspace_bytes, void* stream, void* ev_start, void* ev_stop);

/* ---------------------------------------------------------------------------
 * The same two sweeps for an orbit with transit-timing variations
 *   TTVOrbit._get_model_dt / _warp_times   (src/exoplanet/orbits/ttv.py:158-187)
 * Every time -- cadence or sub-exposure -- is measured from its nearest labelled
 * transit before it reaches the orbit: planet p of draw d has
 *   ttv_edges [n_draw][n_planet][n_edge]      ascending bin edges (ttv.py:158-166:
 *                                             midpoints between consecutive transit times,
 *                                             closed by half a period either side); rows
 *                                             with fewer edges are padded with +inf
 *   ttv_shift [n_draw][n_planet][n_edge + 1]  per bin k = #{edges < t} (searchsorted,
 *                                             ttv.py:171-172): transit time of the bin
 *                                             (ttv.py:167-170) MINUS the record's t0
 * and a time t acts as t - ttv_shift[k(t)]: mean anomaly (t - shift - TP) N, window
 * phase t - shift - T0.  The reverse sweep adds
 *   gshift    [n_draw][n_planet][n_edge + 1]  d sum(gflux * flux) / d ttv_shift
 * Sorted times, one exposure time (or none), no occultations: the windows are enumerated per
 * timing bin (run-enumeration sweep; EXO_FLAG_SPARSE is accepted), and gshift is summed run by
 * run in a fixed order -- bit-reproducible -- unless a transit lies across a bin edge (that
 * list's samples look their bins up one by one).  Otherwise, and for such lists: hardware fp64
 * atomics, one per wave and cadence run -- the last bits of gshift depend on scheduling;
 * everything else is as reproducible as without timing variations.
 * EXO_FLAG_WINDOW: the caller's windows are tested on the warped mid-exposure time
 * (keplerian.py:729-731 with ttv.py:181-187).
 * ------------------------------------------------------------------------- */
#define EXO_MAX_TTV_EDGES 65536
int exo_transit_flux_ttv_fwd_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                 const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                 const double* params, const double* ld, int64_t n_draw,
                                 int32_t n_planet, uint32_t flags, const double* ttv_edges,
                                 const double* ttv_shift, int32_t n_edge, double* flux,
                                 void* workspace, int64_t workspace_bytes, void* stream);
int exo_transit_flux_ttv_vjp_f64(const double* t, int64_t n_cad, const double* texp, int64_t n_texp,
                                 const double* stencil_dt, const double* stencil_w, int32_t n_sub,
                                 const double* params, const double* ld, int64_t n_draw,
                                 int32_t n_planet, uint32_t flags, const double* ttv_edges,
                                 const double* ttv_shift, int32_t n_edge, const double* gflux,
                                 double* flux_out, double* gparams, double* gld, double* gshift,
                                 double* flux_dot, void* workspace, int64_t workspace_bytes,
                                 void* stream);

/* ---------------------------------------------------------------------------
 * Stellar reflex radial velocity from the same Kepler solve
 *   KeplerianOrbit.get_radial_velocity   (src/exoplanet/orbits/keplerian.py:633-677)
 * for n_draw parameter sets:   rv[d][n][p] = AMP (COSW cos f - SINW sin f + ECC COSW),
 * f the true anomaly at mean anomaly (t_n - TP) N.  Per-(draw, planet) record, EXO_RV_NPAR
 * doubles; AMP is the caller's K (keplerian.py:660-669; ECC = 0, COSW = 1, SINW = 0 for a
 * circular orbit, :658-659) or, for the mass-based form (:671-676, with :572-578 and :283-322),
 * conv sin(incl) K0 m_planet.  rv / grv are [n_draw][n_cad][n_planet] (the reference returns
 * one column per planet); gparams [n_draw][n_planet][EXO_RV_NPAR].
 * ------------------------------------------------------------------------- */
#define EXO_RV_NPAR 6
#define EXO_RV_N 0     /* mean motion                      */
#define EXO_RV_TP 1    /* t_periastron                     */
#define EXO_RV_ECC 2   /* eccentricity (NaN result outside [0, 1)) */
#define EXO_RV_COSW 3
#define EXO_RV_SINW 4
#define EXO_RV_AMP 5   /* semi-amplitude                   */
int exo_radial_velocity_fwd_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw,
                                int32_t n_planet, double* rv, void* stream);
int exo_radial_velocity_vjp_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw,
                                int32_t n_planet, const double* grv, double* gparams, void* stream);

/* ---------------------------------------------------------------------------
 * Position / velocity vectors in the observer frame from the same solve.  Replaces, for n_draw parameter sets, the
 * sub-graph behind
 *   KeplerianOrbit.get_{star,planet,relative}_position   (src/exoplanet/orbits/keplerian.py:472-542 -> :380-409)
 *   KeplerianOrbit.get_{star,planet,relative}_velocity   (keplerian.py:580-631 -> :572-578)
 *   KeplerianOrbit.get_relative_angles                   (keplerian.py:544-570: rho, theta from X, Y)
 *   KeplerianOrbit.get_{star,planet,relative}_acceleration   (keplerian.py:679-706)
 * In the orbital plane  position (flags = 0): (u, v) = (1 - e^2) / (1 + e cos f) (cos f, sin f);  flags =
 * EXO_OV_VELOCITY: (u, v) = (-sin f, cos f + e);  flags = EXO_OV_ACCELERATION: (u, v) = -(1 + e cos f)^2 / (1 - e^2)
 * (cos f, sin f);  then _rotate_vector (keplerian.py:283-322) and the amplitude:
 *   x1 = COSW u - SINW v, y1 = SINW u + COSW v;  x2 = x1, y2 = COSI y1, Z = -SINI y1;
 *   X = COSO x2 - SINO y2, Y = SINO x2 + COSO y2;   out[d][n][p] = AMP (X, Y, Z)
 * AMP: a_star / a_planet / -a (times parallax au_per_R_sun, keplerian.py:404-406) for positions, K0 m for velocities,
 * (K0 m)^2 / a for accelerations;
 * COSO = 1, SINO = 0 when the orbit has no Omega; ECC = 0, COSW = 1, SINW = 0 when circular.
 * out / gout [n_draw][n_cad][n_planet][3]; params / gparams [n_draw][n_planet][EXO_OV_NPAR].
 * ------------------------------------------------------------------------- */
#define EXO_OV_NPAR 10
#define EXO_OV_N 0
#define EXO_OV_TP 1
#define EXO_OV_ECC 2    /* NaN result outside [0, 1) */
#define EXO_OV_COSW 3
#define EXO_OV_SINW 4
#define EXO_OV_COSI 5
#define EXO_OV_SINI 6
#define EXO_OV_AMP 7
#define EXO_OV_COSO 8
#define EXO_OV_SINO 9
#define EXO_OV_VELOCITY 1u
#define EXO_OV_ACCELERATION 2u
int exo_orbit_vector_fwd_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, double* out, void* stream);
int exo_orbit_vector_vjp_f64(const double* t, int64_t n_cad, const double* params, int64_t n_draw, int32_t n_planet,
                             uint32_t flags, const double* gout, double* gparams, void* stream);

/* ---------------------------------------------------------------------------
 * celerite GP log-likelihood, value + VJP, for n_draw independent (kernel,
 * residual) pairs.  Replaces what celerite2 (a dependency of the reference,
 * /root/reference/setup.py:36; the user's model calls it, the reference tree
 * itself has no call site) does behind
 *     gp = GaussianProcess(kernel, t=t, diag=diag);  gp.log_likelihood(y - model)
 * i.e. celerite2's `factor` + `solve_lower` + `norm` Ops and their reverse Ops.
 *
 *   t            [n]               sorted, shared by all draws
 *   resid        [n_draw][n]       y - mean model, per draw
 *   diag         [n_diag][n]       white-noise variance (yerr^2 + jitter); n_diag = 1 (shared) or n_draw
 *   coef_real    [n_draw][n_real][2]     (a, c)        of celerite2 Term.get_coefficients()
 *   coef_complex [n_draw][n_complex][4]  (a, b, c, d)  -- a "pair slot": two state indices
 *   pair_kind    NULL, or [n_draw][n_complex] int32: 0 = the slot is a complex term (a, b, c, d);
 *                1 = the slot holds TWO REAL terms (a1, c1, a2, c2).  An SHO term is one complex
 *                term for Q >= 1/2 and two real ones for Q < 1/2 (celerite2 terms.SHOTerm): with
 *                the kind per draw a batch of draws may straddle Q = 1/2 with a static state width
 *   J = n_real + 2 n_complex <= EXO_GP_MAX_J
 *   loglike      [n_draw]          out; -inf if the matrix is not positive definite
 *   state        NULL (value only, sequential recurrences) or exo_celerite_state_doubles() doubles,
 *                16-byte aligned: what the reverse pass re-reads.  Opaque to the caller: whatever
 *                wrote it (this library, this ABI version, the same n_chunks) must read it back
 *   n_chunks     how the series is cut for the time-parallel recurrences: 0 = the library's plan,
 *                1 = sequential recurrences, > 1 = that many chunks (tuning / tests).  The three
 *                calls of a pair -- exo_celerite_state_doubles, forward, reverse -- must be given
 *                the same value (and n, n_draw, n_real, n_complex): the plan is a pure function of
 *                them, the library keeps NO state between calls (EXO_ERR_WORKSPACE if state_doubles
 *                is smaller than exo_celerite_state_doubles() of the same arguments).
 *                EXO_GP_PREPARE_ADJOINT may be or-ed into it (ABI 14), again in every call of the pair: the FORWARD call then
 *                also runs everything of the reverse pass that does not need the cotangent -- the adjoint elements of the
 *                chunks and the scan over them, a dozen or two short, latency-bound launches -- for a cotangent of one, on a
 *                stream of the library's own BESIDE the forward chunk kernel (forked from and joined back into `stream` by
 *                events, so it is legal inside a stream capture and a replayed graph keeps the fork), and the REVERSE call
 *                starts at its chunk kernel, scaling by the actual cotangent (the adjoint is linear in it; a cotangent of
 *                exactly 1 gives the bits of the unflagged pair).  For callers that know the reverse call will follow (a
 *                sampler's value-and-gradient step) -- and that have MEASURED it: the scan's kernels are bound by memory
 *                latency, and beside a chunk kernel that saturates HBM with its checkpoint stores they slow down by about
 *                what the overlap hides (C5 -2 %, J = 10 -3 %; C3 +2.5 %): the Python host side leaves
 *                it off.  A forward call with the flag and no reverse call after it has only done work nobody reads.
 *
 * With a state buffer and n >= 64 the recurrences run in parallel over TIME (docs/DESIGN_r1_r4.md 3.5): the
 * series is cut into chunks, chunk "filtering elements" and a short per-draw scan over them give
 * the recurrence state entering every chunk (and, in the reverse pass, its adjoint), and the
 * ordinary recurrences then run inside all chunks at once.  J <= 6: one lane per (draw, chunk) and
 * a CHECKPOINTED factorisation -- the forward pass stores the recurrence state every 4 cadences (J <= 2) or every 2
 * (J = 3 .. 6), the reverse pass recomputes the cadences in between (which is why it takes the series again);
 * J > 6: a draw on 8 (J = 7, 8) or 16 lanes, (d, z, W, F) of every cadence saved and the S rows at every 2nd to 4th.
 * Same results as the
 * sequential recurrences (1e-14 relative in loglike); draws whose terms do not admit the filter
 * form (a <= 0 or |b d| > a c for some term) or are ill-conditioned are redone by the sequential
 * kernels on the device.
 * ------------------------------------------------------------------------- */
#define EXO_GP_PREPARE_ADJOINT 0x40000000   /* or-ed into n_chunks: see above */
#define EXO_GP_MAX_J 16   /* every state width takes the time-parallel path (1 .. 6: one lane per (draw, chunk); 7, 8: a draw on 8 lanes; 9 .. 16,
                             round 6: on a DPP row of 16 lanes, the scans on 256-thread blocks); the sequential recurrences are the
                             fallback for flagged draws and for series too short to cut */
int64_t exo_celerite_state_doubles(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex,
                                   int32_t n_chunks);
/* the number of chunks the default plan (n_chunks = 0) cuts the series into, for callers that pass it explicitly; sparse != 0:
 * what the sparse entries (exo_celerite_loglike_sparse_*_f64) do best with -- twice as many for J <= 2 (their waves are uneven: two
 * rounds of them balance).  1 = sequential recurrences.                                                               */
int32_t exo_celerite_default_chunks(int64_t n, int64_t n_draw, int32_t n_real, int32_t n_complex, int32_t sparse);
int exo_celerite_loglike_fwd_f64(const double* t, const double* resid, const double* diag, int64_t n_diag,
                                 int64_t n, const double* coef_real, int32_t n_real,
                                 const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                 int64_t n_draw, double* loglike, double* state, int64_t state_doubles,
                                 int32_t n_chunks, void* stream);
/* Reverse: given gloglike [n_draw], the series again and the saved state, write
 *   gresid [n_draw][n], gdiag [n_draw][n] (nullable), gdiag_sum [n_draw] (nullable,
 *   = sum_n d loglike / d diag_n, the cotangent of a scalar jitter),
 *   gcoef_real [n_draw][n_real][2], gcoef_complex [n_draw][n_complex][4] (a pair slot of kind 1
 *   receives the cotangents of (a1, c1, a2, c2)).                                  */
int exo_celerite_loglike_vjp_f64(const double* t, const double* resid, const double* diag, int64_t n_diag,
                                 int64_t n, const double* coef_real, int32_t n_real,
                                 const double* coef_complex, int32_t n_complex, const int32_t* pair_kind,
                                 int64_t n_draw, const double* gloglike, const double* state,
                                 int64_t state_doubles, int32_t n_chunks, double* gresid, double* gdiag,
                                 double* gdiag_sum, double* gcoef_real, double* gcoef_complex, void* stream);

/* The same pair for the usual case of one observed series against a per-draw model:
 *   resid[d][n] = obs[n] - model[d][n]
 * is formed inside the kernels (obs [n], model [n_draw][n]), and the reverse entry returns
 * gmodel = d loglike / d model = -(d loglike / d resid): neither the residual nor the sign
 * flip of its cotangent exists as an array.  Everything else as above; a state buffer
 * written by one pair must be read back by the same pair.                                */
int exo_celerite_loglike_obs_fwd_f64(const double* t, const double* obs, const double* model,
                                     const double* diag, int64_t n_diag, int64_t n,
                                     const double* coef_real, int32_t n_real,
                                     const double* coef_complex, int32_t n_complex,
                                     const int32_t* pair_kind, int64_t n_draw, double* loglike,
                                     double* state, int64_t state_doubles, int32_t n_chunks, void* stream);
int exo_celerite_loglike_obs_vjp_f64(const double* t, const double* obs, const double* model,
                                     const double* diag, int64_t n_diag, int64_t n,
                                     const double* coef_real, int32_t n_real,
                                     const double* coef_complex, int32_t n_complex,
                                     const int32_t* pair_kind, int64_t n_draw, const double* gloglike,
                                     const double* state, int64_t state_doubles, int32_t n_chunks,
                                     double* gmodel, double* gdiag, double* gdiag_sum, double* gcoef_real,
                                     double* gcoef_complex, void* stream);

/* The obs pair with the model (and, from the reverse entry, its cotangent) CADENCE-MAJOR:
 *   model_cm [n][n_draw], gmodel_cm [n][n_draw]
 * -- the layout the light-curve sweep writes under EXO_FLAG_CADENCE_MAJOR.  A lane of the kernels is a
 * draw: with the draws innermost every access of a wave to the series is 512 contiguous bytes instead of 64
 * pieces of 64 rows (C3, 1024 draws x 150 000 cadences: forward chunk kernel 0.86 -> 0.64 ms, reverse
 * 2.02 -> 1.54 ms).  diag / gdiag stay [n_diag][n] / [n_draw][n].  Same arithmetic, same results.     */
int exo_celerite_loglike_obs_fwd_cm_f64(const double* t, const double* obs, const double* model_cm,
                                        const double* diag, int64_t n_diag, int64_t n,
                                        const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex,
                                        const int32_t* pair_kind, int64_t n_draw, double* loglike,
                                        double* state, int64_t state_doubles, int32_t n_chunks, void* stream);
int exo_celerite_loglike_obs_vjp_cm_f64(const double* t, const double* obs, const double* model_cm,
                                        const double* diag, int64_t n_diag, int64_t n,
                                        const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex,
                                        const int32_t* pair_kind, int64_t n_draw, const double* gloglike,
                                        const double* state, int64_t state_doubles, int32_t n_chunks,
                                        double* gmodel_cm, double* gdiag, double* gdiag_sum, double* gcoef_real,
                                        double* gcoef_complex, void* stream);

/* The obs pair with the model SPARSE (round 5): a transit light curve is zero at ~97 % of the cadences, and as the mean of
 * a GP (limb_dark.py:99-232 feeding celerite2's log_likelihood) it used to cross HBM as a dense (draw, cadence) array five
 * times per value + gradient -- written, scattered, read by three kernels -- and its cotangent, of which the light curve's
 * reverse pass reads the same 3 %, twice more.  Here the model is what the EXO_FLAG_SPARSE sweep produces: per draw a few
 * SEGMENTS of cadences, ascending and disjoint, and the values of exactly those cadences:
 *   segment k of draw d   cadences [lo, hi),  lo = seg[d * seg_row + k * seg_step],  hi = seg[d * seg_row + k * seg_step + hi_at],
 *                         k < nseg[d]
 *   model[d][n]           vals[d * val_row + off[d * off_row + k] + (n - lo)]  for lo <= n < hi,  0 outside every segment
 * and the reverse entry writes gvals = d loglike / d vals at the same positions (nothing elsewhere: positions of `gvals` that
 * no segment covers are left as they were).  exo_transit_flux_sparse_model() fills the descriptor from a sweep's workspace.
 * n < 2^31.  Same arithmetic as the dense entries on the same model: same log-likelihood, bit for bit.                 */
int exo_celerite_loglike_sparse_fwd_f64(const double* t, const double* obs, const exo_sparse_model* model,
                                        const double* diag, int64_t n_diag, int64_t n,
                                        const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex,
                                        const int32_t* pair_kind, int64_t n_draw, double* loglike,
                                        double* state, int64_t state_doubles, int32_t n_chunks, void* stream);
int exo_celerite_loglike_sparse_vjp_f64(const double* t, const double* obs, const exo_sparse_model* model,
                                        const double* diag, int64_t n_diag, int64_t n,
                                        const double* coef_real, int32_t n_real,
                                        const double* coef_complex, int32_t n_complex,
                                        const int32_t* pair_kind, int64_t n_draw, const double* gloglike,
                                        const double* state, int64_t state_doubles, int32_t n_chunks,
                                        double* gvals, double* gdiag, double* gdiag_sum, double* gcoef_real,
                                        double* gcoef_complex, void* stream);

/* ---------------------------------------------------------------------------
 * O(N) companions of the likelihood (celerite2's GaussianProcess.dot_tril and the kernel product of
 * GaussianProcess.predict; the reference's docs pair it with celerite2: docs/index.rst:14,48-49):
 *   dot_tril: z[d] = L x[d] with K + diag = L L^T (prior samples: x ~ N(0, I));  NaN from the first
 *             cadence at which the matrix is not positive definite
 *   predict:  mu[d][m] = sum_n k(|tq[m] - t[n]|) alpha[d][n], t and tq sorted, in O(N + M) by one
 *             forward and one backward sweep -- the conditional mean at new times is this with
 *             alpha = (K + diag)^-1 (y - mean) = -d loglike / d resid
 * Sequential in time, one lane per draw: utilities, not part of the per-step path.  Coefficients,
 * pair kinds and diag as for exo_celerite_loglike_fwd_f64.
 * ------------------------------------------------------------------------- */
int exo_celerite_dot_tril_f64(const double* t, const double* diag, int64_t n_diag, int64_t n,
                              const double* coef_real, int32_t n_real, const double* coef_complex,
                              int32_t n_complex, const int32_t* pair_kind, int64_t n_draw, const double* x,
                              double* z, void* stream);
int exo_celerite_predict_f64(const double* t, int64_t n, const double* alpha, const double* coef_real,
                             int32_t n_real, const double* coef_complex, int32_t n_complex,
                             const int32_t* pair_kind, int64_t n_draw, const double* tq, int64_t m,
                             double* mu, void* stream);

/* ---------------------------------------------------------------------------
 * Record packing: the O(planets) algebra of KeplerianOrbit.__init__
 * (src/exoplanet/orbits/keplerian.py:133-281,849-934), get_cl
 * (src/exoplanet/light_curves/limb_dark.py:11-18), the in-transit windows
 * (keplerian.py:733-763) and the secondary flux ratio
 * (src/exoplanet/light_curves/secondary_eclipse.py:67-68) for the standard
 * transit parameterisation, as one kernel + one reverse kernel (in the
 * reference, and in torch, this is ~170 launch-bound elementwise nodes).
 *
 *   orbit_in  [n_draw][n_planet][EXO_NIN]   see EXO_IN_* (R_sun, M_sun, days)
 *   ld_in     [n_draw][2] = (u1, u2), or [n_draw][4] with EXO_FLAG_SECONDARY
 *   flags     EXO_PACK_CIRCULAR (ecc=None: e = 0, omega unused, keplerian.py:182-185),
 *             EXO_FLAG_WINDOW (fill TS/TE[, TS2/TE2]; else +-inf), EXO_FLAG_SECONDARY
 *   params    out [n_draw][n_planet][EXO_NPAR];  ld out [n_draw][3|6]
 * ------------------------------------------------------------------------- */
#define EXO_NIN 10
#define EXO_IN_PERIOD 0
#define EXO_IN_T0 1
#define EXO_IN_B 2
#define EXO_IN_ECC 3
#define EXO_IN_OMEGA 4
#define EXO_IN_R 5
#define EXO_IN_MSTAR 6
#define EXO_IN_RSTAR 7
#define EXO_IN_MPLANET 8
#define EXO_IN_SBR 9
#define EXO_PACK_CIRCULAR 8u
int exo_pack_records_f64(const double* orbit_in, const double* ld_in, int64_t n_draw, int32_t n_planet,
                         uint32_t flags, double* params, double* ld, void* stream);
/* Reverse: cotangents of params / ld -> cotangents of orbit_in / ld_in (window and
 * T0 / PERIOD slots carry no gradient: they only select cadences).              */
int exo_pack_records_vjp_f64(const double* orbit_in, const double* ld_in, int64_t n_draw,
                             int32_t n_planet, uint32_t flags, const double* gparams,
                             const double* gld, double* gorbit_in, double* gld_in, void* stream);
/* Column form of the two calls above, for callers whose parameters are separate arrays (one tensor per
 * constructor argument of KeplerianOrbit): no packing pass in front, no unpacking pass behind.
 *   cols, draw_stride, planet_stride, defaults   HOST arrays of EXO_NIN entries: input k of (draw, planet) is
 *       cols[k][draw * draw_stride[k] + planet * planet_stride[k]]  (device memory; stride 0 = broadcast), or
 *       defaults[k] where cols[k] is NULL (the reference's constructor defaults: m_star = r_star = 1, m_planet = 0)
 *   ld_cols, ld_draw_stride   HOST arrays of 2 (u1, u2) or, with EXO_FLAG_SECONDARY, 4 device pointers / strides
 * Reverse: gscale (optional, [n_draw]) multiplies the record cotangents as they are read -- the chain rule through
 * a per-draw scalar such as L[d] = sum_n gbar[d][n] flux[d][n] without a pass of its own; gcols[k] / gld_cols[k]
 * (HOST arrays; NULL entry = not wanted) receive the cotangents DENSELY, [n_draw][n_planet] / [n_draw] doubles each:
 * the caller sums over whatever it broadcast.                                                                  */
int exo_pack_records_cols_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                              const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                              int64_t n_draw, int32_t n_planet, uint32_t flags, double* params, double* ld,
                              void* stream);
int exo_pack_records_cols_vjp_f64(const double* const* cols, const int64_t* draw_stride, const int64_t* planet_stride,
                                  const double* defaults, const double* const* ld_cols, const int64_t* ld_draw_stride,
                                  int64_t n_draw, int32_t n_planet, uint32_t flags, const double* gparams,
                                  const double* gld, const double* gscale, double* const* gcols,
                                  double* const* gld_cols, void* stream);

/* The No-U-Turn sampler's tree building for a batch of chains (exoplanet_amd/sampling.py: NUTS; the reference hands
 * its models to PyMC's NUTS, docs/tutorials/data-and-models.md:289) on (chains, parameters) arrays that stay on the
 * device.  A transition grows every chain's trajectory by doublings; per doubling
 *   phase 2: every chain draws its direction, the sub-tree starts at that end of its trajectory;
 *   per leaf, phase 0: half kick and drift of the sub-tree's moving end into the scratch rows qn, ph (the caller then
 *            evaluates log-density and gradient at qn into lpn, gn), phase 1: second half kick, energy error,
 *            divergence, multinomial candidate, momentum sums, checkpoint write / generalised turning checks, which
 *            chains go on (the leaf index is kept on the device: a captured leaf needs nothing from the host);
 *   phase 3: a valid sub-tree joins the trajectory (proposal, new end, sums, weights, the trajectory's turning check).
 * ptrs: HOST array of 40 device pointers -- sub-tree: qe, pe, ge [D][n]; eps [D]; on [D] (bytes); H0, logw [D]; psum,
 * sq, sg [D][n]; slp [D]; turn, div [D] (bytes); acc, accn [D]; ckp, cks [n_slot][D][n]; then mass, qn, ph, gn [D][n];
 * lpn [D]; trajectory: ql, pl, gl, qr, pr, gr, tsum [D][n]; logW [D]; propq, propg [D][n]; proplp [D]; active,
 * diverged, going [D] (bytes); depth [D]; eps_abs [D]; R [2 + leaves][D] (uniform random numbers of the doubling: row 0
 * directions, row 1 the merge, row 2 + k leaf k); leaf (int32 [1]).                                               */
int exo_nuts_f64(const void* const* ptrs, int64_t n_chain, int32_t n_param, int32_t n_slot, double max_energy_error,
                 int32_t phase, void* stream);

/* Timing tables of a TTVOrbit whose transits are all labelled and given as offsets `ttvs` from the linear ephemeris
 * (orbits/ttv.py:99-170): transit k of planet p of a draw at tt_k = t0 + period k + ttv_k, k < n_transit[p];
 *   edges [n_draw][n_planet][n_edge]      tt_0 - period/2, midpoints of neighbours, tt_last + period/2, then +inf
 *                                         (ttv.py:158-166); n_edge = max_p n_transit[p] + 1
 *   shift [n_draw][n_planet][n_edge + 1]  bin j -> transit max(0, min(j - 1, n - 1)) (ttv.py:167-170): period k + ttv_k
 * -- the tables the exo_transit_*_ttv_* entry points take.  period, t0: device arrays read at
 * [draw * draw_stride + planet * planet_stride] (strides in elements, 0 = broadcast); ttv, ttv_draw_stride, n_transit:
 * HOST arrays [n_planet] of device pointers (row of a draw: n_transit[p] contiguous doubles), draw strides, counts.
 * Reverse: gshift -> gttv[p] [n_draw][n_transit[p]] and gperiod [n_draw][n_planet], dense, either may be NULL (per
 * planet for gttv); the edges carry no gradient (searchsorted, ttv.py:174) and t0 none through the tables.       */
int exo_ttv_tables_f64(const double* period, int64_t period_draw_stride, int64_t period_planet_stride, const double* t0,
                       int64_t t0_draw_stride, int64_t t0_planet_stride, const double* const* ttv,
                       const int64_t* ttv_draw_stride, const int32_t* n_transit, int64_t n_draw, int32_t n_planet,
                       int32_t n_edge, double* edges, double* shift, void* stream);
int exo_ttv_tables_vjp_f64(const double* period, int64_t period_draw_stride, int64_t period_planet_stride, const double* t0,
                           int64_t t0_draw_stride, int64_t t0_planet_stride, const double* const* ttv,
                           const int64_t* ttv_draw_stride, const int32_t* n_transit, int64_t n_draw, int32_t n_planet,
                           int32_t n_edge, const double* gshift, double* const* gttv, double* gperiod, void* stream);

/* ---------------------------------------------------------------------------
 * celerite2's SHOTerm -> the pair slot of the GP entry points above (one slot per term and draw, two state indices):
 * the term's parameters in any of celerite2's parameterisations -- amplitude S0 or (EXO_SHO_SIGMA) sigma, frequency w0
 * or (EXO_SHO_RHO) the undamped period rho, damping Q or (EXO_SHO_TAU) tau -- to coef[n][4] and kind[n]:
 *   Q >= 1/2: kind 0, one complex term (a, b, c, d) = (S0 w0 Q, a / f, w0 / 2Q, c f),  f = sqrt(max(4 Q^2 - 1, eps))
 *   Q <  1/2: kind 1, two real terms (a1, c1, a2, c2) = (a (1 + 1/f) / 2, c (1 - f), a (1 - 1/f) / 2, c (1 + f)),
 *             f = sqrt(max(1 - 4 Q^2, eps))
 * decided per element on the device (a batch may straddle Q = 1/2).  One launch, and one for the reverse
 * (gcoef[n][4] -> gamp, gfreq, gdamp[n]); in the reference this algebra is celerite2's Python, in torch ~35 kernels.
 * ------------------------------------------------------------------------- */
#define EXO_SHO_SIGMA 1u
#define EXO_SHO_RHO 2u
#define EXO_SHO_TAU 4u
int exo_sho_coefficients_f64(const double* amp, const double* freq, const double* damp, uint32_t flags, double eps,
                             int64_t n, double* coef, int32_t* kind, void* stream);
int exo_sho_coefficients_vjp_f64(const double* amp, const double* freq, const double* damp, uint32_t flags, double eps,
                                 int64_t n, const double* gcoef, double* gamp, double* gfreq, double* gdamp,
                                 void* stream);

/* A dense flux array KEPT ACROSS STEPS (the dense light curve a sampler asks LimbDarkLightCurve.get_light_curve for at every step:
 * src/exoplanet/light_curves/limb_dark.py:163-170, 228-230).  `workspace` holds the sparse output of an EXO_FLAG_SPARSE sweep (same n_cad, n_draw,
 * n_planet; flags: EXO_FLAG_SECONDARY as in that sweep, nothing else).  clear == 0: the summed flux of the cadences in its runs
 * is written into flux[n_draw][n_cad] (planets in order: the bits of the dense sweep), every other entry left alone;
 * clear != 0: those entries are zeroed.  A caller that keeps `flux` and `workspace` from step to step -- zero-initialised
 * both before the first -- calls: scatter(clear) -> sparse sweep -> scatter(write), and holds, after every step, the dense
 * array the dense sweep would have produced, for the price of the sparse one (no fill of every cadence: 1.2 GB at the
 * headline shape).  The library keeps no state: the promise that `flux` is zero outside the runs of `workspace` is the
 * caller's. */
int exo_transit_sparse_scatter_f64(const void* workspace, int64_t workspace_bytes, int64_t n_cad, int64_t n_draw, int32_t n_planet,
                                   uint32_t flags, int32_t clear, double* flux, void* stream);

/* A SUM of SHO terms (celerite2's terms.TermSum, term_1 + term_2 + ...; celerite2 is a dependency of the reference,
 * /root/reference/setup.py:36, its terms are what the reference's GP models are built from), all of it in one launch each way: term k reads its own
 * amp[k], freq[k], damp[k] (host arrays of n_terms <= EXO_SHO_MAX_TERMS device pointers to n doubles each) under its own
 * flags[k], and owns slot k of coef[n][n_terms][4] and kind[n][n_terms] -- the pair-slot arrays the celerite entry points
 * take (no concatenation; the reverse reads gcoef[n][n_terms][4] in place and writes every term's gamp / gfreq / gdamp[n]). */
#define EXO_SHO_MAX_TERMS 8
int exo_sho_coefficients_multi_f64(const double* const* amp, const double* const* freq, const double* const* damp,
                                   const uint32_t* flags, int32_t n_terms, double eps, int64_t n, double* coef, int32_t* kind,
                                   void* stream);
int exo_sho_coefficients_multi_vjp_f64(const double* const* amp, const double* const* freq, const double* const* damp,
                                       const uint32_t* flags, int32_t n_terms, double eps, int64_t n, const double* gcoef,
                                       double* const* gamp, double* const* gfreq, double* const* gdamp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EXOPLANET_AMD_H */
