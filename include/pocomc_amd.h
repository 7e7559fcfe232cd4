/* pocomc_amd -- C ABI of the MI355X (gfx950) engine for pocoMC's flow-preconditioned
 * particle MCMC step.
 *
 * The reference (minaskar/pocomc v1.2.6) has no FFI: its seam is two duck-typed
 * Python contracts:
 *   - the MCMC-kernel contract  kernel(state_dict, function_dict, option_dict)
 *     called from Sampler._mutate (pocomc/sampler.py:568-617), and
 *   - the Flow contract  forward / inverse / log_prob / sample / fit
 *     (pocomc/flow.py:99-384, pocomc/tools.py:336-349).
 * Every entry point below replaces one piece of reference arithmetic on that
 * path and cites it.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers to contiguous row-major arrays owned
 *     by the caller; the library allocates nothing and keeps no state between
 *     calls (except the last error string, per thread);
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work;
 *   - return 0 on success, non-zero on error; pmc_last_error() returns the text;
 *   - "f32"/"f64" in a parameter comment is the element type.  The reference
 *     keeps MCMC state in float64 numpy and the flow in float32 torch
 *     (pocomc/tools.py:279-292); so does this library.
 */
#ifndef POCOMC_AMD_H
#define POCOMC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_ABI_VERSION 9

const char* pmc_last_error(void);
int pmc_abi_version(void);
/* 16 hex digits: sha256 over the Makefile, the .hip and .h files of csrc/ and this header (sorted by name, concatenated) at build
 * time, "+debug" appended by DEBUG_HOOKS builds.  The Python loader recomputes it from the tree and refuses a library
 * that was not built from the sources it sits next to. */
const char* pmc_build_id(void);

/* ------------------------------------------------------------------ flow */

/* Device image of one MAF (pocomc/flow.py:55-68 -> zuko.flows.MAF).  Built by
 * the host side (pocomc_amd/maf_spec.py) and filled by pmc_maf_pack(). */
typedef struct pmc_maf {
    const float* packed;      /* packed weights, T * pk_per_transform floats */
    const int32_t* meta;      /* [8 hdr][T*D feat_of_rank][T*D rank_of_feat][nQ quad meta] */
    int32_t D, H, T;          /* features, hidden width, transforms */
    int32_t Hp, Dp;           /* padded hidden slots / ranks (multiples of 16) */
    int32_t nT, nXT, nOT;     /* hidden, input and output tiles */
    int64_t pk_per_transform;
    int32_t tri_ok;           /* every degree group fits one 16-slot tile */
    int32_t n_out;            /* hyper-network outputs per feature: 2 = affine (MAF), 23 = 8-bin spline (NSF) */
    const uint16_t* lane16;   /* NULL, or the 16-bit image of the lane-per-walker sweep's helper fragments (pmc_maf_pack_lane16):
                               * the inverse of the wide flows then multiplies everything left of the diagonal tile with 16-bit
                               * operands and float32 accumulation -- an opt-in precision, see PMC_INVERSE_TRIANGULAR_LANE16 */
    int32_t lane16_fmt;       /* 1 bfloat16, 2 float16 (0: no image) */
    int32_t reserved;         /* 0, or PMC_MAF_VARIANT_* bits: schedule variants of the inverse sweeps that must agree with
                               * the default bit for bit (the cross-checks of tests/test_gpu_flow.py) */
} pmc_maf_t;

#define PMC_MAF_VARIANT_LEFT_LOOKING 1  /* two-wave spline sweep: the burst wave forms no output partials ahead of time */
#define PMC_MAF_VARIANT_LANE_FOUR 2     /* lane-per-walker sweep of a flow of >= 16 hidden tiles: no fifth wavefront */

#define PMC_INVERSE_AUTO 0
#define PMC_INVERSE_TRIANGULAR 1   /* one sweep over the degree groups */
#define PMC_INVERSE_NAIVE 2        /* the reference's D fixed-point passes (zuko) */
#define PMC_INVERSE_TRIANGULAR_SOLO 6 /* the D <= 64 sweep, one wavefront per 16 rows: the left-looking cross-check of the two-wave sweeps (affine and spline flows) */
#define PMC_INVERSE_TRIANGULAR_LANE 8 /* lane-per-walker chain wavefront + three or four helper wavefronts per 16-64 rows (AUTO: affine flows with >= 16 hidden tiles or D > 64) */
#define PMC_INVERSE_TRIANGULAR_LANE16 9 /* the lane-per-walker sweep with 16-bit helper operands (pmc_maf_t.lane16 required): the chain --
                                        * diagonal tiles, newest ranks, univariate map, log-determinant -- stays float32, the
                                        * left-looking products of the three helper wavefronts run on v_mfma_f32_16x16x16_bf16 / _f16
                                        * with float32 accumulation; two walker subsets share a workgroup at D = 128 (BASELINE config
                                        * 5: 5000 walkers in one round).  AUTO takes it when the image is attached and the flow is
                                        * one the lane sweep is preferred for; _LANE (8) always multiplies in float32 */
#define PMC_INVERSE_TRIANGULAR_DUO 7  /* the same sweep with a second, burst wavefront per 16 rows, right-looking (AUTO: affine flows of < 16 hidden tiles, spline flows with D <= 64) */

/* packed[i] = idx[i] >= 0 ? flat[idx[i]] : 0   (canonical fp32 params -> kernel layout) */
int pmc_maf_pack(const float* flat, const int32_t* pack_idx, float* packed, int64_t n_packed, void* stream);

/* Flow.forward, pocomc/flow.py:99-114: data -> latent.  x,z f32 [n][D]; ladj f32 [n] or NULL;
 * log_prob f32 [n] or NULL (Flow.log_prob, flow.py:134-147). */
int pmc_maf_forward(const pmc_maf_t* m, const float* x, float* z, float* ladj, float* log_prob,
                    int64_t n, void* stream);

/* bf16 matrix-core variant of pmc_maf_forward for the affine flows (BASELINE config 5: "8-layer MAF bf16"):
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation, the univariate map and the log-determinant in fp32.  The weight
 * fragments are a bf16 image of the fp32 master parameters: image u16 [T * image_per_transform], built by
 * pmc_maf_pack_bf16 through MAFSpec.pack_index_bf16 (image[i] = bf16(flat[idx[i]]) or 0; round to nearest even);
 * biases are read from the fp32 image of `m`.  idx i64 [n] or NULL: row gather like pmc_maf_loss_grad.
 * Precision: activations and weights carry 8 mantissa bits -- the parity tests state the tolerance against the fp32
 * oracle. */
int pmc_maf_pack_bf16(const float* flat, const int32_t* pack_idx, uint16_t* image, int64_t n, void* stream);
int pmc_maf_forward_bf16(const pmc_maf_t* m, const uint16_t* image, int64_t image_per_transform, const float* x,
                         float* z, float* ladj, float* log_prob, int64_t n, const int64_t* idx, void* stream);

/* Flow.inverse, pocomc/flow.py:116-132: latent -> data with the log-determinant of the
 * inverse map.  z,x f32 [n][D]; ladj f32 [n] or NULL. */
int pmc_maf_inverse(const pmc_maf_t* m, const float* z, float* x, float* ladj, int64_t n,
                    int algo, void* stream);
/* Which sweep PMC_INVERSE_AUTO (and with it the MCMC step) launches for this flow: 1 / 0.  _is_duo: the two-wave sweep of
 * the affine flows with D <= 64 for a call of n rows; _is_lane: the lane-per-walker sweep; _is_nsf2: the two-wave spline
 * sweep.  (bench.py names the kernel its roofline line is about with them.) */
int pmc_maf_inverse_auto_is_duo(const pmc_maf_t* m, int64_t n);
int pmc_maf_inverse_auto_is_lane(const pmc_maf_t* m);
int pmc_maf_inverse_auto_is_nsf2(const pmc_maf_t* m);

/* 16-bit helper image of the lane-per-walker inverse sweep (Flow.inverse of the wide flows, flow.py:116-132, in the opt-in
 * precision of BASELINE config 5): image u16 [pmc_maf_lane16_elems(m)], derived on the device from m->packed (call again
 * after every pmc_maf_pack); fmt 1 = bfloat16, 2 = float16, round to nearest even.  Attach it as m->lane16 / lane16_fmt. */
int64_t pmc_maf_lane16_elems(const pmc_maf_t* m);
int pmc_maf_pack_lane16(const pmc_maf_t* m, int fmt, uint16_t* image, void* stream);

/* Training-side device image (host side: MAFSpec.train_index() / train_jobs()).  One minibatch is two launches:
 * a chain kernel (one workgroup per 16 rows: forward, loss, the backward sweep of the data gradients) that keeps every
 * activation and every delta of the batch in the scratch arrays below, and a weight-gradient kernel tiled over the
 * WEIGHT matrices (one workgroup per 16 x 16 tile with an unmasked entry, contracting over all rows of the batch) that
 * writes the canonical gradient -- no atomics, no per-workgroup copies of the gradient, a fixed summation order. */
typedef struct pmc_maf_train {
    const float* packedT;     /* transposed weight fragments, filled by pmc_maf_pack with the packT map */
    const int32_t* gmap;      /* canonical index of every element of a gradient tile / bias row, -1 = masked / padding */
    int64_t pkT_per_transform;
    int64_t gmap_per_transform;
    const int32_t* jobs;      /* device int32 [n_jobs][8] (MAFSpec.train_jobs): {kind_a, off_a, kind_b, off_b, gmap offset of
                               * the weight tile | -1, gmap offset of the 16 bias entries | -1, 0, 0}; kinds 0 xt_scratch,
                               * 1 act_scratch, 2 delta_scratch, 3 par_scratch; offsets in floats inside a row set's block */
    int32_t n_jobs;
    int32_t max_sets;         /* row sets of 16 the scratch arrays hold; a larger batch is taken in chunks of that many */
    int32_t n_sq_partial;     /* capacity of sq_partial, >= n_jobs */
    int32_t table_waves;      /* the wave count `tables` was built for: must equal pmc_maf_train_waves(m) */
    const int32_t* tables;    /* device int32 (MAFSpec.train_tables): [16][8] cost ranks of the hidden tiles every wave of the
                               * chain workgroup owns (-1 ends a row), [T][D] rank in transform t + 1 of the feature at rank r of
                               * transform t, [T][D] likewise for t - 1 */
    float* xt_scratch;        /* [max_sets][T + 1][Dp * 16] the input of every transform, then z */
    float* act_scratch;       /* [max_sets][T][3][Hp * 16] hidden activations h0 h1 h2 */
    float* delta_scratch;     /* [max_sets][T][3][Hp * 16] their gradients da0 da1 da2 */
    float* par_scratch;       /* [max_sets][T][par_per_transform] the hyper-network's outputs (affine: nOT tiles of (shift,
                               * raw); spline: nXT panels of 23 tiles), overwritten by their gradients */
    int64_t par_per_transform;/* floats: 256 * nOT (affine) or 256 * 23 * nXT (spline) */
    float* loss_partial;      /* [max_sets] */
    float* sq_partial;        /* [n_sq_partial] per-workgroup sums of squared gradient entries */
    const float* wsum;        /* NULL: c_n uses the sum of THIS call's weights; else f32 [1] (device) with the sum of
                               * the whole batch's weights, e.g. all-reduced over the ranks of a sharded batch */
} pmc_maf_train_t;

/* One minibatch of Flow.fit, pocomc/flow.py:297-323: loss and parameter gradient.
 *   loss += sum_n c_n * (-log_prob(x_n));  c_n = 1 (w == NULL, flow.py:309) or
 *   c_n = w_n * wmul / sum(w of the batch) (flow.py:311-312 with wmul = 1000).
 * The batch is rows idx[0..n) of x / w (idx i64 device, or NULL for rows 0..n).  x f32 [.][D];
 * grad f32 [n_params] in the canonical layout is OVERWRITTEN at every unmasked entry (masked
 * entries are never touched: allocate it zeroed); loss f32 [1] is ACCUMULATED.
 * tr->sq_partial[0 .. n_jobs) receives the sums of squares that pmc_maf_train_epoch clips with. */
/* Wavefronts per chain workgroup for this flow (16; 8 for spline flows of hidden width <= 64). */
int pmc_maf_train_waves(const pmc_maf_t* m);
int pmc_maf_loss_grad(const pmc_maf_t* m, const pmc_maf_train_t* tr, const float* x, const float* w,
                      const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * bf16 matrix-core training of the wide affine flows (BASELINE config 5: D = 128, 8 transforms, H = 512; flow.py:297-323
 * with the hyper-networks of flow.py:46-90).  A 512-row batch is too few rows for "one workgroup per 16 rows": the
 * layers are dense products [out x in] . [in x rows] on v_mfma_f32_16x16x32_bf16, a workgroup per 32 x 32 output tile
 * (its four wavefronts split the contraction), one launch per dependent layer (8 T + 4 per 512 rows).  fp32 master parameters, fp32
 * accumulation, fp32 univariate map / log-determinant / loss; activations and their gradients are stored as bf16 in
 * both orientations ([row][unit] feeds the next layer and the data gradients, [unit][row] the weight gradients).
 *
 * image  u16 [T * image_per_transform]: per transform, row-major and zero where masked / padded,
 *          W0f [HK][DK]  W0b = W0f^T  W1f [HK][HK]  W1b  W2f  W2b  W3f [OK][HK] (row 2 * feature + {shift, raw})  W3b
 *          with hidden units in slot order (MAFSpec.slot_unit), features in canonical order,
 *          DK = ceil32(D), HK = ceil32(Hp), OK = 2 * DK;
 * image_idx i32, same shape: canonical index of every element or -1 (MAFSpec.wide_index(); it is both the gather map
 *          that builds the image -- pmc_maf_pack_bf16 -- and, in its W?f parts, the scatter map of the weight gradients);
 * bias   f32 [T * bias_per_transform]: b0 b1 b2 [HK each] b3 [OK] in the same orders, bias_idx i32 likewise
 *          (pmc_maf_pack builds it);
 * scratch: pmc_maf_wide_scratch_bytes(m) bytes of device memory; wsum as in pmc_maf_train_t.
 * pmc_maf_wide_refresh re-derives image and bias from the parameters (after every optimizer step). */
typedef struct pmc_maf_wide {
    uint16_t* image; const int32_t* image_idx; int64_t image_per_transform;
    float* bias; const int32_t* bias_idx; int64_t bias_per_transform;
    void* scratch; int64_t scratch_bytes;
    const float* wsum;
} pmc_maf_wide_t;
int64_t pmc_maf_wide_scratch_bytes(const pmc_maf_t* m);
int pmc_maf_wide_refresh(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* params, void* stream);
/* Same contract as pmc_maf_loss_grad (loss accumulated, grad overwritten at the unmasked entries); affine flows only. */
int pmc_maf_loss_grad_bf16(const pmc_maf_t* m, const pmc_maf_wide_t* wd, const float* x, const float* w,
                           const int64_t* idx, float wmul, float* grad, float* loss, int64_t n, void* stream);
/* pmc_maf_train_epoch with the bf16 loss / gradient: per batch  loss+grad -> clip -> AdamW -> refresh of the bf16 image
 * (the fp32 kernel images of opt are NOT refreshed: repack them once after the fit).  sq_scratch f32 [PMC_ADAMW_SCRATCH]. */
struct pmc_adamw;
int pmc_maf_train_epoch_bf16(const pmc_maf_t* m, const pmc_maf_wide_t* wd, struct pmc_adamw* opt, const float* x,
                             const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                             float* sq_scratch, void* stream);

/* Optimizer state of pmc_maf_train_epoch: torch.optim.AdamW (flow.py:268) on the canonical
 * parameter vector + the two kernel images that are refreshed after every step. */
typedef struct pmc_adamw {
    float* params; float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n_params;
    const int32_t* pack_idx; float* packed; int64_t n_packed;       /* pmc_maf_t.packed */
    const int32_t* packT_idx; float* packedT; int64_t n_packedT;    /* pmc_maf_train_t.packedT */
    double lr, beta1, beta2, eps, weight_decay;
    double max_norm;          /* clip_grad_norm_ (flow.py:318); <= 0 disables */
    int64_t step;             /* optimizer steps taken so far; advanced by the call */
    /* optional inverse of pack_idx / packT_idx in CSR form: the image positions parameter i is copied to are
     * scatter_dst[scatter_ptr[i] .. scatter_ptr[i+1]), position d < n_packed in `packed`, d - n_packed in `packedT`.
     * With both non-NULL the AdamW kernel refreshes the two images itself (no separate gather launch per step). */
    const int32_t* scatter_ptr;   /* device int32 [n_params + 1] */
    const int32_t* scatter_dst;   /* device int32 [scatter_ptr[n_params]] */
    float* snapshot;              /* NULL, or device f32 [n_params]: pmc_maf_train_epoch writes the parameters behind the epoch's
                                   * LAST optimizer step there as well (the state Flow.fit restores at an early stop,
                                   * flow.py:364-374, without a separate copy launch per epoch) */
} pmc_adamw_t;

/* One epoch of the training loop, pocomc/flow.py:297-323: for every batch of `batch_size` rows
 * (rows perm[b0 .. b0+nb) of x and w, or consecutive rows when perm == NULL; the last batch may
 * be short): loss/gradient, global-norm clip, AdamW step, refresh of both kernel images.
 * loss f32 [1] accumulates the sum of the batch losses (flow.py:321).  Everything is enqueued on
 * `stream`; nothing synchronises. */
int pmc_maf_train_epoch(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                        const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                        void* stream);
/* The same with a gate: the epoch's first optimizer step waits for gate_event (a hipEvent_t, or NULL) -- its loss / gradient
 * launches, which only read the parameters, do not.  Lets the validation pass of the previous epoch (flow.py:327-348), which
 * reads the kernel images that step rewrites, run on another stream next to this epoch's first batch: a batch's chain kernel
 * occupies 32 of the 256 compute units. */
int pmc_maf_train_epoch_gated(const pmc_maf_t* m, const pmc_maf_train_t* tr, pmc_adamw_t* opt, const float* x,
                              const float* w, const int64_t* perm, int64_t n, int64_t batch_size, float* loss,
                              void* gate_event, void* stream);

/* Options of Flow.fit.
 * Weight regularisation, flow.py:314-315 with regularization_loss :387-421 as its docstring defines it (the function
 * itself lacks its `return`, so the reference raises TypeError when the option is set): the batch loss gains
 *   R = sum over the hyper-networks' weight matrices of |W| / laplace_scale + W^2 / (2 gaussian_scale^2)
 * (a scale <= 0 switches its term off).  loss f32 [1] += mult * R; grad (or NULL) += dR/dparams, to be called between
 * the loss/gradient of a batch and its clipped optimizer step.  is_weight u8 [n]: 1 for entries of weight matrices
 * (masked-out ones included: they are entries of zuko's `weight` parameters too), 0 for biases.
 * scratch f32 [PMC_ADAMW_SCRATCH]. */
int pmc_weight_penalty(const float* params, const uint8_t* is_weight, float* grad, int64_t n, double laplace_scale,
                       double gaussian_scale, float mult, float* loss, float* scratch, void* stream);
/* Noise augmentation, flow.py:305 / :334: out f32 [n][D] = x + scale * N(0, 1), Philox keyed by (seed, pass, row). */
int pmc_add_noise_f32(const float* x, int64_t n, int32_t D, float scale, uint64_t seed, uint64_t pass, float* out,
                      void* stream);
/* The same for rows row0 .. row0 + n of a larger set (a rank's shard of a data-parallel fit): row r draws what row
 * row0 + r of the whole set draws. */
int pmc_add_noise_rows_f32(const float* x, int64_t n, int32_t D, float scale, uint64_t seed, uint64_t pass, uint64_t row0,
                           float* out, void* stream);
/* out f32 [1] = mean_j ||x[row] - x[j]||: what flow.py:241-245 scales the noise with (its `torch.mean(min_dist)` is
 * the mean of the LAST row's distance vector, not of the nearest-neighbour distances computed above it). */
int pmc_mean_distance_f32(const float* x, int64_t n, int32_t D, int64_t row, float* out, void* stream);

/* The validation pass of one epoch, flow.py:327-348, in one call: for every batch (rows perm[b0 .. b0+nb) or
 * consecutive rows) loss += sum_n c_n * (-log_prob(x_n)) with c_n as in pmc_maf_loss_grad.
 * logp_scratch f32 [n] (one forward launch evaluates every row; the batches only shape the weight normalisation). */
int pmc_maf_valid_epoch(const pmc_maf_t* m, const float* x, const float* w, const int64_t* perm, int64_t n,
                        int64_t batch_size, float* logp_scratch, float* loss, void* stream);

/* out += sum_n -(logp_n * c_n)  (validation loss, flow.py:336-341);  out += sum_n v_n. */
int pmc_neg_weighted_sum(const float* logp, const float* w, const float* wsum, float wmul, float* out,
                         int64_t n, void* stream);
int pmc_sum_f32(const float* v, float* out, int64_t n, void* stream);

/* torch.nn.utils.clip_grad_norm_(max_norm) (flow.py:318; max_norm <= 0 disables) followed by one
 * torch.optim.AdamW step (flow.py:268, :319).  step counts from 1.
 * sq_scratch f32 [PMC_ADAMW_SCRATCH]. */
#define PMC_ADAMW_SCRATCH 256
int pmc_adamw_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   double lr, double beta1, double beta2, double eps, double weight_decay,
                   double max_norm, int64_t step, float* sq_scratch, void* stream);

/* ---------------------------------------------------------------- scaler */

/* pocomc/scaler.py Reparameterize with the Sampler's settings (diagonal affine,
 * scale=True; sampler.py:309-313). */
typedef struct pmc_scaler {
    const double* low;        /* [D] */
    const double* high;       /* [D] */
    const double* mu;         /* [D] */
    const double* sigma;      /* [D] */
    const int32_t* kind;      /* [D] 0 none, 1 low only, 2 high only, 3 both (scaler.py:463-489) */
    const int32_t* bc;        /* [D] bit 0 periodic, bit 1 reflective; NULL = no boundary conditions */
    const double* log_width;  /* [D] log(high-low) where kind == 3 (scaler.py:421, :424), else 0 */
    int32_t D;
    int32_t logit;            /* 0 probit, 1 logit */
    int32_t scale;            /* apply the affine part (scaler.py:219) */
    int32_t reserved;
    double sum_log_sigma;     /* sum(log(sigma)), scaler.py:306 */
} pmc_scaler_t;

/* Reparameterize.inverse (scaler.py:204-226) + the boundary-condition round trip of
 * mcmc.py:94-97 + the finite mask of mcmc.py:100-102.
 * u_in: f32 [n][D] (flow output) or NULL;  u_in64: f64 [n][D] or NULL (exactly one non-NULL)
 * u_out f64 [n][D] (u', re-derived from x' when boundary conditions apply), x f64 [n][D],
 * logdetj f64 [n], finite int32 [n] (1 = logdetj and every x finite).
 * x_colmajor: optional second copy of x as f64 [D][n] (what the host likelihood reads as an (n, D)
 * Fortran-ordered array: numpy's inner loops then run over n instead of over D), or NULL. */
int pmc_scaler_inverse(const pmc_scaler_t* s, const float* u_in, const double* u_in64, double* u_out,
                       double* x, double* x_colmajor, double* logdetj, int32_t* finite, int64_t n, void* stream);

/* Reparameterize.forward (scaler.py:180-202), no input check.  x f64 [n][D] -> u f64 [n][D]. */
int pmc_scaler_forward(const pmc_scaler_t* s, const double* x, double* u, int64_t n, void* stream);

/* ----------------------------------------------------------------- prior */

/* pocomc/prior.py: a product of frozen scipy.stats distributions.  Unlike the likelihood it is not a
 * user black box: the families below are evaluated on the device, everything else stays on the host. */
#define PMC_PRIOR_UNIFORM 1   /* scipy.stats.uniform(loc, scale) */
#define PMC_PRIOR_NORM 2      /* scipy.stats.norm(loc, scale) */
typedef struct pmc_prior {
    const int32_t* family;    /* [D] */
    const double* loc;        /* [D] */
    const double* scale;      /* [D] */
    int32_t D;
    int32_t reserved;
} pmc_prior_t;

/* Prior.logpdf, prior.py:70-100, with the gating of mcmc.py:105-107: logp[k] = -inf where finite[k] == 0
 * (finite may be NULL).  x f64 [n][D], logp f64 [n]. */
int pmc_prior_logpdf(const pmc_prior_t* pr, const double* x, const int32_t* finite, double* logp, int64_t n,
                     void* stream);

/* Completion word of a kernel in pinned host memory: the last workgroup to finish stores `value` to `*flag`
 * after every workgroup's results are visible system-wide; the host then spins on the word (pmc_wait_flag)
 * instead of waking up through the runtime's stream / event synchronisation (~10-20 us per wait).
 * Ordering: every workgroup waits for the acknowledgement of its own stores (agent-scope release), draws a
 * ticket (agent-scope acq_rel); the last one stores the word with a system-scope release.  The results the
 * host reads behind the word must therefore live in memory the device does not cache -- pinned (coherent,
 * fine-grained) host memory, as hipHostMalloc returns it by default; the host reads the word with acquire
 * semantics (pmc_wait_flag does). */
typedef struct pmc_done {
    int64_t* flag;            /* pinned, device-accessible host memory */
    int64_t value;
    uint32_t* ticket;         /* device, zeroed once by the caller (only kernels without their own ticket use it) */
} pmc_done_t;
/* spin until *flag == value; non-zero return after timeout_s seconds */
int pmc_wait_flag(const int64_t* flag, int64_t value, double timeout_s);

/* Cache warmer for the likelihood's input (host side only; csrc/host_prefetch.hip).  n_threads helper threads, pinned to
 * cpus[i] (or unpinned when cpus == NULL) -- cores that share the L3 with the thread that calls the likelihood.  A job:
 * wait until the completion word *flag (i64, pinned host memory, written by the kernels as for pmc_wait_flag) reaches
 * value, then read buf[0 .. bytes) once, back to front, so that the likelihood finds x' in the cache hierarchy instead of
 * in DRAM.  Best effort: jobs are dropped when the helpers are more than 16 behind, a job gives up after timeout_s. */
void* pmc_prefetcher_create(int32_t n_threads, const int32_t* cpus);
int pmc_prefetcher_submit(void* prefetcher, const void* flag, int64_t value, const void* buf, int64_t bytes,
                          double timeout_s);
void pmc_prefetcher_destroy(void* prefetcher);

/* pmc_scaler_inverse and pmc_prior_logpdf in ONE launch (the step's pre-phase is a chain of small
 * latency-bound kernels; each launch costs ~15-20 us end to end): additionally
 * logp f64 [n] <- Prior.logpdf(x') on the finite rows, -inf elsewhere (mcmc.py:105-107).
 * prior == NULL and logp == NULL: exactly pmc_scaler_inverse.
 * finite_copy / logp_copy: optional second destinations.  Together with x_colmajor they may point into
 * PINNED HOST memory (device-accessible, e.g. hipHostMalloc): the kernel then writes what the host
 * callbacks read straight over PCIe while it computes, and no device-to-host copy follows it. */
int pmc_scaler_inverse_prior(const pmc_scaler_t* s, const pmc_prior_t* prior, const float* u_in,
                             const double* u_in64, double* u_out, double* x, double* x_colmajor,
                             double* logdetj, int32_t* finite, double* logp, int32_t* finite_copy,
                             double* logp_copy, const pmc_done_t* done, int64_t n, void* stream);

/* ------------------------------------------------------------ MCMC step */

#define PMC_KIND_TPCN 0   /* t-preconditioned Crank-Nicolson proposal (mcmc.py:77-85, :394-402) */
#define PMC_KIND_RWM 1    /* random-walk proposal (mcmc.py:251-253, :561-563) */

/* Random variates of one step: replay arrays (parity tests; recorded from the
 * reference's legacy numpy stream) or, when they are NULL, a counter-based
 * Philox4x32-10 keyed by (seed, step, particle). */
typedef struct pmc_rng {
    const double* gamma;      /* [n] standard-gamma draws G_k (mcmc.py:80) or NULL */
    const double* normal;     /* [n][D] N(0,1) draws (mcmc.py:85) or NULL */
    const double* uniform;    /* [n] U(0,1) draws (mcmc.py:137) or NULL */
    uint64_t seed;
    uint64_t step;
    uint64_t offset;          /* global index of this shard's first particle */
} pmc_rng_t;

/* Proposal, mcmc.py:77-85 (tpCN) / :251-253 (RWM).
 * cur32: f32 [n][D] (theta of the preconditioned kernels lives in float32, it is the
 * numpy view of a torch float32 tensor, tools.py:336-340) or NULL; cur64: f64 [n][D]
 * (u of pcn/rwm) or NULL.  mu f64 [D], inv_cov f64 [D][D], chol f64 [D][D] (lower).
 * Outputs: prop64 f64 [n][D], prop32 f32 [n][D] (what the flow consumes, tools.py:344),
 * quad f64 [n] = diff^T inv_cov diff (current), quad_prop f64 [n] (proposed); the last
 * two may be NULL for RWM.  cn_a = (1 - sigma**2)**0.5, evaluated by the caller exactly as
 * mcmc.py:85 does (ignored for RWM). */
int pmc_propose(int kind, const float* cur32, const double* cur64, const double* mu,
                const double* inv_cov, const double* chol, double nu, double sigma, double cn_a,
                const pmc_rng_t* rng, double* prop64, float* prop32, double* quad,
                double* quad_prop, int64_t n, int32_t D, void* stream);

/* Arrays of one particle population (current or proposed). */
typedef struct pmc_state {
    float* theta32;           /* f32 [n][D] current theta (preconditioned kernels) or NULL */
    double* u;                /* f64 [n][D] */
    double* x;                /* f64 [n][D] */
    double* logdetj;          /* f64 [n] */
    double* logl;             /* f64 [n] */
    double* logp;             /* f64 [n] */
    float* logdetj_flow;      /* f32 [n] or NULL */
} pmc_state_t;

typedef struct pmc_proposal {
    const double* theta64;    /* f64 [n][D] proposed theta or NULL (pcn/rwm) */
    const double* u;          /* f64 [n][D] */
    const double* x;          /* f64 [n][D] */
    const double* logdetj;    /* f64 [n] */
    const double* logl;       /* f64 [n]  (-inf where not evaluated, mcmc.py:118) */
    const double* logp;       /* f64 [n]  (-inf where not finite, mcmc.py:107) */
    const float* logdetj_flow;/* f32 [n] or NULL */
    const double* quad;       /* f64 [n] current quadratic form (tpCN) or NULL */
    const double* quad_prop;  /* f64 [n] proposed quadratic form (tpCN) or NULL */
} pmc_proposal_t;

/* Metropolis ratio + accept + reductions: mcmc.py:124-156 (and the three variants).
 * sums f64 [D+4] <- { sum(alpha), sum(logl+logp) after accept, sum(logl+logp+logdetj) after accept,
 *                     number accepted, sum_k theta[k][0..D) after accept (cur theta32 or u) }
 * kind: PMC_KIND_TPCN adds the Student-t terms -A+B (mcmc.py:124-129); preconditioned != 0 adds the
 * flow log-determinants and moves theta (preconditioned_pcn / preconditioned_rwm), otherwise u is the
 * moved variable (pcn / rwm).
 * alpha_out f64 [n] and accept_out int32 [n] may be NULL.  workspace: pmc_accept_workspace_bytes(). */
int64_t pmc_accept_workspace_bytes(int64_t n, int32_t D);
int pmc_accept(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta, double nu,
               const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums,
               void* workspace, int64_t n, int32_t D, void* stream);

/* pmc_accept without the memset that re-arms the workspace (the kernel leaves the ticket word at zero itself;
 * the caller zeroes the workspace ONCE after allocating it and never aborts a launch), and with an optional
 * second destination of the sums -- pinned host memory, so that no device-to-host copy follows the kernel.
 * prop->logl / prop->logp may likewise point into pinned host memory (read straight over PCIe). */
int pmc_accept_armed(int kind, int preconditioned, pmc_state_t* cur, const pmc_proposal_t* prop, double beta,
                     double nu, const pmc_rng_t* rng, double* alpha_out, int32_t* accept_out, double* sums,
                     double* sums_copy, const pmc_done_t* done, void* workspace, int64_t n, int32_t D, void* stream);

/* All buffers of one step in one place, for the composite entry points below. */
typedef struct pmc_step {
    int32_t kind;             /* PMC_KIND_TPCN | PMC_KIND_RWM */
    int32_t preconditioned;
    int64_t n;
    int32_t D;
    int32_t inverse_algo;
    const pmc_maf_t* maf;     /* NULL unless preconditioned */
    const pmc_scaler_t* scaler;
    pmc_state_t cur;          /* current population (device) */
    const double* mu;         /* device [D];  */
    const double* inv_cov;    /* device [D][D] */
    const double* chol;       /* device [D][D] */
    double* p_theta64;        /* proposal buffers (device): see pmc_propose / pmc_scaler_inverse / pmc_accept */
    float* p_theta32;
    float* p_u32;
    float* p_ldjf;
    double* p_u;
    double* p_x;
    double* p_xT;             /* optional column-major copy of x' (NULL: the host gets the row-major x') */
    double* p_logdetj;
    int32_t* p_fin;
    double* quad;
    double* p_quad;
    double* p_logl;
    double* p_logp;
    double* alpha;
    int32_t* accept;
    double* sums;             /* device [D+4] */
    void* ws;                 /* pmc_accept_workspace_bytes() */
    const double* h_mu;       /* pinned host [D] or NULL: uploaded to mu at the start of pmc_step_pre */
    double* h_x;              /* pinned host [n*D] */
    int32_t* h_fin;           /* pinned host [n] */
    const double* h_logl;     /* pinned host [n] */
    const double* h_logp;     /* pinned host [n] */
    double* h_sums;           /* pinned host [D+4] */
    int32_t* h_accept;        /* pinned host [n] or NULL */
    void* ev_inv0;            /* optional hipEvent_t recorded right before / after the flow-inverse launch */
    void* ev_inv1;
    const pmc_prior_t* prior; /* non-NULL: logp' is evaluated on the device in pmc_step_pre and copied to h_logp_out */
    double* h_logp_out;       /* pinned host [n] (may alias h_logp) */
    /* Philox variates drawn ahead of time (throughput mode, rng->normal == NULL): the draws of step k+1 do not
     * depend on step k, so pmc_step_pre(k) enqueues pmc_rng_fill(k+1) behind its own kernels and the GPU
     * generates them while the host evaluates the likelihood; the kernels of step k+1 then read them like
     * replayed variates.  Two buffer sets, used alternately; rng_ready (host int64, -1 at start) holds the step
     * whose variates sit in set [step & 1].  All NULL: the kernels draw inline. */
    double* rng_normal[2];    /* device f64 [n][D] each */
    double* rng_gamma[2];     /* device f64 [n] each (tpCN) */
    double* rng_uniform[2];   /* device f64 [n] each */
    int64_t* rng_ready;       /* host */
    void* ev_pre_done;        /* optional hipEvent_t recorded when x', finite and logp' are complete (the variates of the
                               * next step are generated behind it: wait for this event, not for the stream) */
    int64_t* h_done;          /* pinned host int64 [2] or NULL: with host_direct, [0] <- step + 1 when pmc_step_pre's results
                               * are in host memory, [1] <- step + 1 when pmc_step_post's are (pmc_wait_flag) */
    uint32_t* done_ticket;    /* device uint32 [1], zeroed once */
    int32_t no_fuse;          /* bit 0: launch the proposal and the flow inverse separately; bit 1: keep the scaler (+ prior)
                               * a launch of its own instead of the epilogue of the fused proposal + inverse launch */
    int32_t host_direct;      /* 1: h_x (column-major, p_xT == NULL), h_fin, h_logp_out and h_mu are device-accessible
                               * pinned memory that the kernels read / write themselves -- no copies in pmc_step_pre */
    /* Adaptation on the device (mcmc.py:152-156 and the variants :314-318, :476-480, :627-631): with adapt_state
     * non-NULL, pmc_step_post lets the accept kernel's last block update  {sigma, (1-sigma^2)^0.5, mu[D]}  from the
     * step's sums, and pmc_step_pre reads sigma / cn_a / mu from there instead of its arguments and h_mu -- so
     * the host can enqueue pmc_step_pre(k+1) right behind pmc_step_post(k) without waiting for the sums.  The
     * step-dependent factors are passed in (they depend on the step number only); the arithmetic keeps numpy's
     * operation order, so a host that applies the same update to the same sums holds the same sigma and mu. */
    double* adapt_state;      /* device f64 [2 + D] or NULL */
    int32_t adapt_mode;       /* PMC_ADAPT_* ; 0: the accept kernel leaves adapt_state alone */
    int32_t adapt_pad;
    double adapt_c_sigma;     /* 1/(i+1)^0.75 (tpCN kinds) or 1/(i+1) (RWM kinds) of the update that follows this step */
    double adapt_c_mu;        /* 1/(i+1) */
    double adapt_cap;         /* min(2.38/sqrt(D), 0.99) */
    double adapt_n_total;     /* number of walkers the sums run over */
    const double* adapt_other[7];  /* device [D+4] sums of the other row ranges of the walker set (their accepts precede this
                               * one on the stream): added to this step's sums before the update and before the host copy */
    int32_t adapt_n_other;
    int32_t adapt_pad2;
    /* "every row is clean": with h_clean non-NULL pmc_step_pre leaves there, before the completion word h_done[0], the
     * number of rows whose x' is not finite or whose device-evaluated logp' is not finite (mcmc.py:100-109's two masks) --
     * 0 lets the host skip both mask scans and hand the whole block to the likelihood.  -1: the launch sequence that ran
     * does not count (no device prior, or x' not handed over by the kernels); the host then scans h_fin / logp' as before. */
    int64_t* h_clean;         /* pinned host int64 [1] or NULL */
    uint32_t* clean_count;    /* device uint32 [1], zeroed once by the caller */
    /* 1: in the HOST copy of x' (h_x with host_direct) a row that does not reach the likelihood -- x' or the device-evaluated
     * logp' not finite, the rows h_clean counts -- carries the walker's current x (cur.x) instead of x'.  The host can then
     * hand the whole block to a row-wise likelihood and overwrite those rows' values with -inf (mcmc.py:118-121 does that to
     * the rows it left out) instead of gathering the other rows first (x'[mask], mcmc.py:117: 280 us for 6.5e3 x 50 doubles).
     * The device copy p_x keeps x'.  Needs h_clean / clean_count and a device prior. */
    int32_t fill_rejected;
    int32_t fill_pad;
} pmc_step_t;

#define PMC_ADAPT_TPCN 1      /* sigma <- |min(sigma + c (mean alpha - 0.234), cap)|      (mcmc.py:152, :476) */
#define PMC_ADAPT_PRWM 2      /* sigma <- sigma + c (mean alpha - 0.234)                   (mcmc.py:314) */
#define PMC_ADAPT_RWM 3       /* sigma <- |sigma + c (mean alpha - 0.234)|                 (mcmc.py:627) */
#define PMC_ADAPT_MU 8        /* or-ed in: mu <- mu + c_mu (float32(mean theta) - mu)      (mcmc.py:156) */

/* The Philox variates of one step into arrays, exactly the values the kernels draw inline for the same
 * (seed, step, offset): normal f64 [n][D] (stream 1, pair j/2), gamma f64 [n] = standard gamma of shape
 * `gamma_shape` (stream 0; NULL or gamma_shape <= 0: skipped), uniform f64 [n] (stream 2; NULL: skipped). */
int pmc_rng_fill(const pmc_rng_t* rng, double gamma_shape, double* normal, double* gamma, double* uniform,
                 int64_t n, int32_t D, void* stream);
int pmc_event_synchronize(void* ev);

/* Proposal (pmc_propose, preconditioned kernels: cur32 = theta) and flow inverse (pmc_maf_inverse) of the
 * proposed theta' in ONE launch: every wave proposes for its 16 walkers and sweeps them through the inverse
 * flow without a global round trip.  prop64 f64 [n][D] = theta', quad / quad_prop f64 [n] (tpCN),
 * u_out f32 [n][D], ladj f32 [n] or NULL.  Affine flows with D <= 64 only (error otherwise). */
int pmc_propose_inverse(int kind, const float* cur32, const double* mu, const double* inv_cov, const double* chol,
                        double nu, double sigma, double cn_a, const pmc_rng_t* rng, double* prop64, double* quad,
                        double* quad_prop, const pmc_maf_t* maf, float* u_out, float* ladj, int64_t n, void* stream);

/* mcmc.py:77-102 in one call: [H2D mu] -> propose -> flow inverse -> scaler inverse -> D2H x', finite. */
int pmc_step_pre(const pmc_step_t* s, const pmc_rng_t* rng, double nu, double sigma, double cn_a, void* stream);
/* mcmc.py:124-156 in one call: H2D logl', logp' -> accept + reductions -> [D2H sums, accept mask]. */
int pmc_step_post(const pmc_step_t* s, const pmc_rng_t* rng, double beta, double nu, int want_mask,
                  int copy_sums, void* stream);
/* The adaptation of pmc_step_t.adapt_state as a launch of its own, for walker sets whose sums come in parts
 * (row ranges stepped one after the other, mcmc.LanedEngine; ranks, after the all-reduce):
 *   total[j] = parts[0][j] + parts[1][j] + ...   (j < D + 4, n_parts <= 8, added in this order)
 * -> total_out (device, may be NULL) and h_sums (pinned host, may be NULL); then, if adapt_state != NULL, the
 * update selected by adapt_mode (PMC_ADAPT_*) with the factors of pmc_step_t.adapt_*; finally done->flag. */
int pmc_adapt_update(const double* const* parts, int32_t n_parts, int32_t D, double* total_out, double* h_sums,
                     double* adapt_state, int32_t adapt_mode, double c_sigma, double c_mu, double cap, double n_total,
                     const pmc_done_t* done, void* stream);
int pmc_stream_synchronize(void* stream);

/* The all-reduce of a sharded step (one exchange of D + 4 doubles per step so that sigma, mu and the stop
 * decision of mcmc.py:152-180 are global) inside this library: one process per GPU on one node, a mailbox per rank in its
 * HBM shared through hipIpc handles (xGMI peer stores + sequence words), sums formed in RANK ORDER by every rank -- the same
 * bits everywhere, whatever the arrival order.
 *   comm = pmc_comm_create(rank, world <= 8, width >= D + 4);  pmc_comm_handle(comm, h64) -> 64 bytes the host language
 *   exchanges between the ranks' processes (e.g. torch.distributed.all_gather_object);  pmc_comm_connect(comm, world x 64 B).
 *   pmc_comm_adapt_update = pmc_adapt_update with the parts' total summed over the ranks first (every rank calls it in
 *   the same order; timeout_s bounds the wait for a peer: on a timeout sums[0] is NaN and `done` is written all the same).
 *   pmc_pipeline_set_comm(pipeline, comm): the pipelined step of a sharded walker set.
 * pmc_comm_create fails (NULL) where the device cannot give it UNCACHED memory -- the protocol is only valid for mailboxes
 * no device caches.  pmc_comm_create_host: the same communicator with the mailboxes in pinned, coherent HOST memory (a
 * POSIX shared-memory object per rank, registered with every rank's HIP runtime; the 64-byte handle is its name): every
 * store and poll crosses PCIe instead of xGMI -- the tier for nodes without hipIpc peer mappings, slower, same bits.  All
 * ranks of a communicator are of one kind (pmc_comm_kind: 0 device, 1 host). */
void* pmc_comm_create(int32_t rank, int32_t world, int32_t width);
void* pmc_comm_create_host(int32_t rank, int32_t world, int32_t width);
int pmc_comm_kind(void* comm);
/* host mailboxes: drop the shared-memory NAME once every rank has connected (mappings stay valid; nothing is left in /dev/shm
 * if a process dies later); no-op for device mailboxes */
int pmc_comm_unlink(void* comm);
int pmc_comm_handle(void* comm, void* out64);
int pmc_comm_connect(void* comm, const void* handles);
void pmc_comm_destroy(void* comm);
int pmc_comm_adapt_update(void* comm, const double* const* parts, int32_t n_parts, int32_t D, double* total_out, double* h_sums,
                          double* adapt_state, int32_t adapt_mode, double c_sigma, double c_mu, double cap, double n_total,
                          const pmc_done_t* done, double timeout_s, void* stream);
int pmc_pipeline_set_comm(void* pipeline, void* comm);

/* The pipelined step of a walker set stepped as row ranges ("lanes": the device works on the proposals of lane k+1 and
 * the accept of lane k-1 while the host evaluates the likelihood of lane k), as one object: everything of a loop iteration
 * of pocomc/mcmc.py:74-156 that is not a black box is enqueued by these calls; the host language keeps the prior /
 * likelihood callbacks and the stop rule (mcmc.py:159-180, from the sums in the last lane's h_sums).
 *   lanes[k]: the lanes' pmc_step_t (kept alive and unchanged by the caller; host_direct, h_done, done_ticket, rng_* set;
 *   ONE adapt_state shared by all lanes, initialised by the caller with {sigma, (1 - sigma^2)^0.5, mu}); offsets[k]: the
 *   lane's first global walker index (Philox key).
 * pmc_pipeline_start enqueues the pre-steps (pmc_step_pre) of step `first_step` for every lane.  Then, per step:
 *   pmc_pipeline_next(p, -1, ...)   returns when lane 0's x', finite mask and logp' are in host memory;
 *   pmc_pipeline_next(p, k, ...)    the host has written lane k's logl' (and a host prior's logp'): enqueues its accept
 *                                   (pmc_step_post); k < last: returns when lane k+1's x' is there; k == last: that accept
 *                                   adds the lanes' sums and applies the adaptation (adapt_mode, c_sigma, c_mu, cap, n_total
 *                                   as in pmc_step_t.adapt_*), the pre-steps of step + 1 follow (more != 0), and the call
 *                                   returns when the sums are in the last lane's h_sums.
 * Same launches in the same order as one pmc_step_pre / pmc_step_post per lane from the host language. */
void* pmc_pipeline_create(const pmc_step_t* const* lanes, int32_t n_lanes, uint64_t seed, const uint64_t* offsets,
                          void* prefetcher, double wait_timeout_s, void* stream);
void pmc_pipeline_destroy(void* pipeline);
int pmc_pipeline_start(void* pipeline, double nu, int64_t first_step);
int pmc_pipeline_next(void* pipeline, int32_t lane_done, double beta, double nu, int32_t adapt_mode, double c_sigma,
                      double c_mu, double cap, double n_total, int32_t more);
/* out f64 [6] <- { seconds spent waiting for x', waiting for the sums, enqueuing accepts, enqueuing pre-steps, steps, 0 } */
int pmc_pipeline_stats(void* pipeline, double* out, int32_t reset);
/* hipEvent helpers for the host language (live kernel timing inside bench.py). */
void* pmc_event_create(void);
int pmc_event_record(void* ev, void* stream);
float pmc_event_elapsed_ms(void* a, void* b);
void pmc_event_destroy(void* ev);

/* --------------------------------------------------------- particle math */

/* Particles.compute_logw_and_logz, particles.py:215-231, un-normalised part:
 * logw[t*N+k] = logl[t][k]*beta_final - (logsumexp_i(logl[t][k]*beta[i]-logz[i]) - log T). */
int pmc_logw(const double* logl, const double* beta, const double* logz, double beta_final,
             double* logw, int32_t T, int64_t N, void* stream);

/* stats f64 [4] <- { max(logw), sum exp(logw-max), sum exp(2(logw-max)), sum 1-(1-w)^k }  with
 * w = exp(logw-max)/sum: everything effective_sample_size / unique_sample_size /
 * increment_logz need (tools.py:56-133).  k <= 0 skips the last entry.
 * workspace: pmc_reduce_workspace_bytes(P). */
int64_t pmc_reduce_workspace_bytes(int64_t P);
int pmc_logw_stats(const double* logw, int64_t P, int64_t k, double* stats, void* workspace, void* stream);

/* trim_weights, tools.py:10-53: the weight threshold the reference's downward percentile scan stops at.
 * w f64 [P] (normalised; not modified); result f64 [2] <- { threshold, index of the accepted percentile bin };
 * the caller keeps the samples with w >= threshold (tools.py:38-39).  One radix sort + two scans
 * instead of up to `bins` np.percentile passes.  workspace: pmc_trim_workspace_bytes(P). */
int64_t pmc_trim_workspace_bytes(int64_t P);
int pmc_trim_threshold(const double* w, int64_t P, double ess, int32_t bins, double* result,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ---- SMC bookkeeping on a pool that stays in HBM (csrc/pool.hip) ------------------------------------------------ */

/* Importance weights of Sampler._reweight, sampler.py:779-781: w = exp(logw - max) / sum.
 * stats f64 [>= 2] on the device: { max, sum exp(logw - max) } as left by pmc_logw_stats. */
int pmc_weights_from_logw(const double* logw, int64_t P, const double* stats, double* w, void* stream);

/* out f64 [1] (device) <- sum of a f64 [n], added in a fixed order (np.sum of a weight vector: tools.py:34, :168). */
int pmc_sum_f64(const double* a, int64_t n, double* out, void* stream);

/* trim_weights, tools.py:38-41, after pmc_trim_threshold: keep the samples with w >= *threshold in their order,
 * idx_out i64 [<= P] <- their indices, w_out f64 [<= P] <- their renormalised weights, count i64 [1] <- how many
 * (threshold, count: device memory).  workspace: pmc_trim_select_workspace_bytes(P). */
int64_t pmc_trim_select_workspace_bytes(int64_t P);
int pmc_trim_select(const double* w, int64_t P, const double* threshold, int64_t* idx_out, double* w_out,
                    int64_t* count, void* workspace, int64_t workspace_bytes, void* stream);

/* First and second moments of the rows x[idx[r]], r < n (idx NULL: rows 0..n-1), with weights w (NULL: ones):
 * v f64 [2] <- { V1 = sum w, V2 = sum w^2 }, mean f64 [D] <- sum w x / V1, S f64 [D][D] <- sum w (x - mean)(x - mean)^T.
 * Everything np.average / np.cov(aweights=w) / np.var of Geometry.fit (geometry.py:44-49) and the start values of
 * fit_mvstud (student.py:46-47) need.  Rows float64 (x) or float32 (x32: theta, the float32 output of the flow).
 * Ordered reductions.  workspace: pmc_moments_workspace_bytes(D). */
int64_t pmc_moments_workspace_bytes(int32_t D);
int pmc_moments(const double* x, const float* x32, const int64_t* idx, const double* w, int64_t n, int32_t D,
                double* mean, double* S, double* v, void* workspace, int64_t workspace_bytes, void* stream);

/* np.median(data, 1) of student.py:45 over the rows x[idx[r]] (idx NULL: rows 0..n-1): one segmented radix sort;
 * for an even n the mean of the two middle elements in the input's precision (np.mean of a float32 pair is a
 * float32).  med64 f64 [D] for float64 rows, med32 f32 [D] for float32 rows.
 * workspace: pmc_column_medians_workspace_bytes(n, D, is_f32). */
int64_t pmc_column_medians_workspace_bytes(int64_t n, int32_t D, int32_t is_f32);
int pmc_column_medians(const double* x, const float* x32, const int64_t* idx, int64_t n, int32_t D, double* med64,
                       float* med32, void* workspace, int64_t workspace_bytes, void* stream);

/* The full affine map of Reparameterize(diagonal=False), scaler.py:288-292 / :308-313, on rows f64 [n][D]:
 * mode 0: out = mu + M in (M = L: _inverse_affine), mode 1: out = M (in - mu) (M = L^-1: _forward_affine).  in != out. */
int pmc_affine_rows(const double* M, const double* mu, const double* in, double* out, int64_t n, int32_t D,
                    int32_t mode, void* stream);

/* Bootstrap of the evidence estimate, Sampler._compute_evidence, sampler.py:905-911:
 * out[b] = logsumexp(logw[choice(n, n)]) - log(n) for b < B, the draws from Philox keyed by (seed, b, draw).
 * stats f64 [>= 1] on the device: stats[0] = max(logw) (pmc_logw_stats). */
int pmc_bootstrap_logz(const double* logw, int64_t n, const double* stats, int64_t B, uint64_t seed, double* out,
                       void* stream);
/* The same with the draws given: draws i64 [B][n] on the device, values in [0, n) -- np.random.choice(n, n) of
 * sampler.py:908 as the reference drew it (parity tests replay the recorded stream). */
int pmc_bootstrap_logz_replay(const double* logw, int64_t n, const double* stats, int64_t B, const int64_t* draws,
                              double* out, void* stream);

/* Sampler._resample gather, sampler.py:707-713: dst[i] = src[idx[i]] for the five arrays. */
int pmc_gather(const int64_t* idx, int64_t n_out, int32_t D, const double* u, const double* x,
               const double* logdetj, const double* logl, const double* logp, double* u_out,
               double* x_out, double* logdetj_out, double* logl_out, double* logp_out, void* stream);

/* Resampling indices from normalised weights w f64 [P]:
 * multinomial (np.random.choice(p=w), sampler.py:703: cdf.searchsorted(uniform, 'right'))
 * with uniforms f64 [n_out]; systematic (tools.py:136-186) with one offset in [0,1).
 * cdf: f64 [P] scratch. */
int pmc_resample_multinomial(const double* w, int64_t P, const double* uniforms, int64_t n_out,
                             double* cdf, int64_t* idx, void* stream);
int pmc_resample_systematic(const double* w, int64_t P, double offset, int64_t n_out, double* cdf,
                            int64_t* idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POCOMC_AMD_H */
