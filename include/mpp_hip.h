/*
 * mpp_hip.h -- C ABI of libmppgpu.so, the MI355X (gfx950) implementation of the
 * MPP / RJMCMC sampling path of Ayana-Inria/MPP_CNN_RS_object_detection.
 *
 * The reference has no FFI: its boundary is a set of Python call signatures.
 * Each entry point below names the reference interface it stands behind
 * (paths relative to the reference repository root).  The Python binding that
 * mirrors those signatures is mpp_cnn_rs_object_detection_amd/{hip_api,point_set,sampler}.py;
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - every call returns 0 on success, <0 on error; mpp_last_error(ctx) has the text;
 *  - the caller owns every host buffer; the library owns device memory for the
 *    lifetime of the ctx; pointers passed with on_device=1 are borrowed, never freed;
 *  - a ctx is bound to one GPU and one HIP stream and is not thread-safe; distinct
 *    ctxs are independent; no global state, no callbacks, no Python objects;
 *  - a ctx holds n_tiles independent tiles of identical H x W ("tiles" are the
 *    reference's 256-px patches, models/mpp/mpp_model.py:231-248); one workgroup
 *    samples one tile.
 *  - "slot": points of a tile are kept in dense slots 0..n-1 (birth appends, death
 *    moves the last slot into the hole, move/transform rewrites in place).
 */
#ifndef MPP_HIP_H
#define MPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

#define MPP_MAX_UNIT 8
#define MPP_MAX_PAIR 2
#define MPP_NCLASS 32
#define MPP_NKERNEL 10

/* unit energies: models/mpp/energies/data_energies.py, prior_energies.py */
enum {
  MPP_U_POSITION = 0,    /* p[0]=thr: -2*(det[x,y]-thr)                       data_energies.py:12-24   */
  MPP_U_SHAPE_REMAP = 1, /* p[0..2]=coef p[3..5]=icpt: mean_k(-2*sigmoid(coef_k*P_k+icpt_k)+1)
                            data_energies.py:27-45 + energy_setup_legacy.py:142-147                     */
  MPP_U_MARK_NEG = 2,    /* p[0]=k: -P_k[x,y,class_k]            energy_setup_no_calibration.py:68-80  */
  MPP_U_MARK_REMAP = 3,  /* p[0]=k p[1]=coef p[2]=icpt           (calib_marks=True)                     */
  MPP_U_AREA = 4,        /* p[0]=min p[1]=max: max(0,min-A,A-max)             prior_energies.py:53-67  */
  MPP_U_RATIO_PRIOR = 5, /* p[0]=target: |target-ratio|                       prior_energies.py:70-78  */
  MPP_U_CONST = 6,       /* p[0]=c (the unit energy of the reference's unit tests)                      */
  /* the classic image energies of the contrast setup (energy_setups/energy_setup_contrast.py:29-105): they read the
   * picture given with mpp_set_image, not a score map.  Chains with one of them run the extended kernel (spec_waves 1 or 8). */
  MPP_U_CONTRAST = 7,    /* p = {measure, dilation, gap, erode, thresh, fac, default_value}; measure: 0 lafarge, 1 craciun,
                            2 craciun2, 3 mean, 4 t-test, 5 debug: fac*measure(fill pixels, rim pixels) summed over the
                            channels - thresh                                  energies/classics.py:100-196 */
  MPP_U_GRADIENT = 8     /* p = {thresh, eps}: -|mean(grad . outline normal)| - thresh; the image holds np.gradient of
                            the picture, [H][W][C][2]                          energies/classics.py:199-235 */
};
/* pair energies: prior_energies.py */
enum {
  MPP_P_OVERLAP = 0,     /* area(P1^P2)/(min(A1,A2)+1e-6), reduce max         prior_energies.py:11-24  */
  MPP_P_ALIGN = 1,       /* 1-|cos(a1-a2)|-rewarding; p[0]=rewarding          prior_energies.py:27-50  */
  MPP_P_DIST_LE = 2,     /* [d<=max_dist]  (test/test_energy_graph.py:26-37)                            */
  MPP_P_DIST_LT = 3      /* [d< max_dist]  (test/test_interacting_points_set.py:30-42)                  */
};
enum { MPP_REDUCE_MAX = 0, MPP_REDUCE_MIN = 1 };
/* E(x) = sum_u F(lin0 + sum_k coef_k*g_k*v_k(u)); g_k=[v_gate<=gate_thr] if gated else 1.
 * F=identity: combination/hierarchical.py:13-48 and the plain sum of energy_graph.py:130-131;
 * F=2*sigmoid-1: combination/logistic.py:14-26. */
enum { MPP_C_LINEAR = 0, MPP_C_LOGISTIC = 1 };

typedef struct { int32_t kind, gated; double coef; double p[8]; } mpp_unit_term;
typedef struct { int32_t kind, gated, reduce, _pad; double coef, max_dist; double p[2]; } mpp_pair_term;

/* what EnergySetup.make_energies + an EnergyCombinationModel amount to
 * (energies/energy_setups/ and energies/combination/) */
typedef struct {
  int32_t n_unit, n_pair, combinator, gate_term;
  double gate_thr, lin0;
  mpp_unit_term unit[MPP_MAX_UNIT];
  mpp_pair_term pair[MPP_MAX_PAIR];
} mpp_model;

/* the three ValueMappings (models/shape_net/mappings.py:9-74): lower bin edges and ranges */
typedef struct {
  int32_t cyclic[3], _pad;
  double vmin[3], vmax[3];
  double edges[3][MPP_NCLASS];
} mpp_mappings;

/* kernel mixture, rjmcmc_sampler/kernels/make_kernels.py:50-177 (same order as its list) */
enum { MPP_K_UBIRTH = 0, MPP_K_UDEATH, MPP_K_DBIRTH, MPP_K_DDEATH, MPP_K_GTRANS, MPP_K_DTRANS,
       MPP_K_GTRANSF, MPP_K_DTRANSF,
       MPP_K_SPLIT, MPP_K_MERGE };   /* kernels/split_and_merge_kernels.py:40-178; p = 0 unless use_split_merge */
typedef struct {
  double p_kernel[MPP_NKERNEL];
  double sigma_trans;      /* 2   */
  double sigma_transform;  /* 0.1 */
  int32_t max_delta;       /* 8   */
  int32_t _pad;
  double split_radius;     /* 16  make_kernels.py:148 (SplitSampler.pos_radius = merge_radius) */
  double split_sigma;      /* 0.1 make_kernels.py:150 (x mark range) */
} mpp_kernels;

/* one fully specified proposal (tape replay); custom_types/perturbation.py:7-12 flattened */
typedef struct {
  int32_t kernel;
  int32_t target;          /* slot removed / moved, -1 if none.  Split: the split point, (aux0, aux1) = position
                            * delta, (as, ar, aa) = mark deltas.  Merge: target = p0, param_id = p1 (-1: none) */
  int32_t ax, ay;          /* proposed point */
  double as, ar, aa;
  double aux0, aux1;       /* raw normal deltas of the Gaussian kernels (Perturbation.data['delta']) */
  int32_t param_id, new_class;
  double u_accept;         /* the uniform of rjmcmc.py:113 */
} mpp_proposal;

/* per-step record; custom_types/rjmcmc.py:5-14 */
typedef struct {
  double dE, fwd, bwd, log_alpha, T;
  int32_t accepted, n_after;
} mpp_step_out;

typedef struct mpp_ctx mpp_ctx;

/* lifetime ------------------------------------------------------------------ */
int mpp_create(int device_id, mpp_ctx **out);
int mpp_destroy(mpp_ctx *ctx);
const char *mpp_last_error(mpp_ctx *ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int mpp_set_stream(mpp_ctx *ctx, void *hip_stream);
int mpp_synchronize(mpp_ctx *ctx);
/* "spec_waves": proposals evaluated speculatively per round, one wave each (1 = strictly one at a
 * time); "spec_lanes" (0 = off): lane mode, 4 waves of which `v` lanes each evaluate one step on
 * their own, 4*v steps per round (overrides spec_waves); "deep" (default 128; 0 = off; a power of two in 8..256): deep
 * rounds -- every LANE of the chain's spec_waves waves evaluates one step, up to `v` (at most 64 * spec_waves) steps per
 * round, the steps sorted by kernel type across the workgroup, neighbour energies evaluated by the whole wave for all its
 * steps at once (csrc/mpp_deep.hip); used for chains drawn from Philox under the shipped energy setups (overlap / max +
 * alignment / min pair terms, no split / merge, no classic image energy), others run one wave per step as before;
 * "handover" (default 1): a chain of 8 waves starts with one wave per step and is handed to the deep rounds once about 6 of
 * 8 steps commit per round (a hot chain changes its state every few steps: short rounds suit it better); same chain;
 * "handover_at" (default 1280): that number of steps committed per round, x 256;
 * "handover_tiles" (default 64): only launches of at most this many chains start that way (the launch that hands over
 * ends when its last chain has cooled down);
 * "deep_fixed" (tests): a fixed number of steps per round instead of the adaptive depth; "deep_gain" (8..64, default 12):
 * the adaptive depth in eighths of the smoothed number of steps the last rounds committed; read-only "deep_stat0".."deep_stat3":
 * rounds, steps evaluated, rounds with a second pass, steps committed by the last mpp_run.  The chain is identical for every setting. "point_capacity": slots per tile (before mpp_set_maps), "cell_capacity" (points per
 * 32-px cell, at most 64), "auto_grow" (default 1): a chain that would exceed either capacity stops BEFORE that step and
 * mpp_run / mpp_replay double the capacity (while the chain still fits the 160 KB of LDS) and continue it -- the reference's
 * point set has no capacity (point_set/point_set.py:45-188); with 0 the call fails with -11 / -12 and the chain can be
 * continued by hand from the written-back state ("grow_events", read-only, counts the re-launches), "replicas" (before mpp_set_maps): v independent chains per tile, chain t on the maps of
 * tile t % n_tiles; "remap_table" (-1 auto, default; 0 never; 1 always): chains of a model with the
 * MPP_U_SHAPE_REMAP term read the remapped mark probabilities from [H][W][32] float64 tables built once per mpp_set_maps (as
 * the reference does, energy_setup_legacy.py:142-147) instead of evaluating three sigmoids per proposal -- the same values bit
 * for bit; auto: while the tables fit 2 GB (a few tiles sampled long; for hundreds of tiles building them costs more than they save); reading it back tells whether they are in use; "force_accept": apply every proposal without the Metropolis test (the kernel random
 * walks of models/mpp/perturbation_sampler.py:152-169); "scratch_grid_min_points" (default 256; 0 = never): configurations of
 * at least that many points get a candidate grid (PointsSet.get_potential_neighbors, point_set.py:111-145) for
 * mpp_total_energy / mpp_delta_batch / mpp_delta_vectors / mpp_papangelou instead of a scan of all points.  Read-only: "n_chains", "lds_bytes", and the spatial-hash
 * geometry "grid_nx", "grid_ny", "grid_res" (point_set/point_set.py:58-61) */
int mpp_set_option(mpp_ctx *ctx, const char *name, int64_t value);
int64_t mpp_get_option(mpp_ctx *ctx, const char *name);

/* score maps: ImageWMaps.detection_map / param_dist_maps (custom_types/image_w_maps.py:11-22).
 * det: [n_tiles][H][W] float32; m0..m2: [n_tiles][H][W][32] float32 (size, ratio, angle). */
int mpp_set_maps(mpp_ctx *ctx, int n_tiles, int H, int W, const float *det, const float *m0,
                 const float *m1, const float *m2, int on_device);
/* the picture the classic image energies read (ImageWMaps.image, custom_types/image_w_maps.py:11-22, as prepared by
 * ContrastEnergy.__post_init__ / GradientEnergy.__post_init__, classics.py:113-149, :207-215): img [n_tiles][H][W][C]
 * float32, after mpp_set_maps (same n_tiles, H, W); C = 1 or 3 for MPP_U_CONTRAST, 2 or 6 (np.gradient, [..][C/2][2]) for
 * MPP_U_GRADIENT.  on_device: borrowed.  Dropped by the next mpp_set_maps. */
int mpp_set_image(mpp_ctx *ctx, int n_tiles, int C, const float *img, int on_device);
int mpp_set_model(mpp_ctx *ctx, const mpp_model *model, const mpp_mappings *mappings);
/* make_kernels(image_data, intensity, rng): intensity[n_tiles] = max(1,len(init)) (sample_rjmcmc.py:68) */
int mpp_set_kernels(mpp_ctx *ctx, const mpp_kernels *kernels, const double *intensity);

/* EPointsSet (point_set/energy_point_set.py:18-166) ---------------------------- */
int mpp_set_points(mpp_ctx *ctx, int tile, int n, const int32_t *xy, const double *marks);
int mpp_get_points(mpp_ctx *ctx, int tile, int cap, int32_t *n, int32_t *xy, double *marks);
int mpp_count(mpp_ctx *ctx, int tile, int32_t *n);
/* all tiles at once (the per-tile results `Pool.map` hands back, mpp_model.py:250-262): n[n_tiles]; if xy and marks are
 * not NULL, xy [n_tiles][cap][2] and marks [n_tiles][cap][3] receive the first min(n[t], cap) points of every tile */
int mpp_get_points_all(mpp_ctx *ctx, int cap, int32_t *n, int32_t *xy, double *marks);
/* The configurations of tiles 0..n-1 of the ctx packed into ONE fixed-capacity record buffer ON THE DEVICE: the send
 * buffer of the all-gather (RCCL over xGMI) that stands in for the result list `Pool.map` returns in
 * mpp_model.py:250-262.  out_dev [capacity+1][7] float64 (device pointer of this ctx's GPU): row 0 = (count, 0...),
 * row 1+k = (tile_ids[t], x + anchors[t][0], y + anchors[t][1], size, ratio, angle, 0) -- image coordinates as
 * merge_patches forms them (data_loaders.py:133-138), tiles in order, points in slot order; the rest is zeroed.
 * tile_ids [n], anchors [n][2]: host arrays.  count (may be NULL) receives the number of records; -4 if > capacity. */
int mpp_pack_detections(mpp_ctx *ctx, int n, const int32_t *tile_ids, const int32_t *anchors, int capacity,
                        double *out_dev, int32_t *count);
/* total_energy(): combined energy and, optionally, [n][n_unit+n_pair] per-point vectors */
int mpp_total_energy(mpp_ctx *ctx, int tile, double *energy, double *vectors_or_null);
/* energy_delta(Perturbation) for a batch of perturbations with list removals/additions:
 * case i removes slots rem[rem_off[i]..rem_off[i+1]) and adds rectangles add_off[i]..add_off[i+1] */
int mpp_delta_batch(mpp_ctx *ctx, int tile, int n_cases, const int32_t *rem_off, const int32_t *rem,
                    const int32_t *add_off, const int32_t *add_xy, const double *add_marks, double *dE);
/* The same perturbations, but the per-point energy VECTORS (n_unit unit terms, then n_pair reductions) before and
 * after each one instead of the combined dE: what EnergyComputeTorch.compute feeds to the torch weight model in
 * train_energy_combination/train_ordering_criterion.py:27-40,101-118 (via energy_graph.py:139-225).
 * Row i*stride + j of before/after ([n_cases*stride][n_unit+n_pair]) and mask ([n_cases*stride]): j < n is
 * existing slot j, j >= n the (j-n)-th rectangle added by case i; stride >= n + additions of every case.
 * mask: 0 untouched, 1 neighbour of a change (both rows valid), 2 removed (before only), 3 added (after only). */
int mpp_delta_vectors(mpp_ctx *ctx, int tile, int n_cases, const int32_t *rem_off, const int32_t *rem,
                      const int32_t *add_off, const int32_t *add_xy, const double *add_marks, int stride,
                      double *before, double *after, unsigned char *mask);
/* papangelou(u, remove_u_from_point_set=True, return_energy_delta=True) of every point */
int mpp_papangelou(mpp_ctx *ctx, int tile, double *dE);
/* merge_patches(..., method='distance', distance) of models/mpp/data_loaders.py:122-161 and the two scorings around it
 * (mpp_model.py:296-304), for EVERY tile of the ctx at once: a tile holds the aggregated detections of one image (mpp_set_points)
 * on that image's score maps.  On the device: the Papangelou intensity of every point in its tile's configuration; the walk
 * of data_loaders.py:140-159 -- in index order every not-yet-removed point keeps, among the not-yet-removed points within
 * `distance` of it, only the one with the best intensity (scores equal to 1e-9 tie, the first wins) --; the removals in the
 * order EPointsSet.remove leaves (the last point takes the hole); the Papangelou values of the survivors.  The tiles'
 * configurations ARE the survivors afterwards.  n_out [n_tiles], xy [n_tiles][cap][2], marks [n_tiles][cap][3], dE
 * [n_tiles][cap] = E(with u) - E(without u) (score = exp(-dE)), n_removed [n_tiles] (may be NULL).  -4: a tile holds more
 * points than the device walk takes (8192): merge that image on the host. */
int mpp_merge_score(mpp_ctx *ctx, double distance, int cap, int32_t *n_out, int32_t *xy, double *marks, double *dE,
                    int32_t *n_removed);
/* naive_detection (sample_rjmcmc.py:23-35): threshold + greedy distance-NMS, sets every tile's points */
int mpp_naive_init(mpp_ctx *ctx, double threshold, double nms_distance);

/* RJMCMC (rjmcmc_sampler/rjmcmc.py:52-187, sample_rjmcmc.py:38-102) ------------- */
int mpp_set_schedule(mpp_ctx *ctx, double T0, double alpha, double T_target);
/* replay n given proposals on one tile; out may be NULL */
int mpp_replay(mpp_ctx *ctx, int tile, int n, const mpp_proposal *tape, mpp_step_out *out);
/* n_steps of the chain on every tile at once; proposals from Philox4x32-10(key=seed,
 * ctr=(step, block, chain = chain0+tile)).  trace_tile>=0 records that tile's steps. */
int mpp_run(mpp_ctx *ctx, int64_t n_steps, uint64_t seed, uint32_t chain0, int trace_tile,
            mpp_step_out *out_or_null, mpp_proposal *props_or_null);
/* Per-chain Philox keys: chain t of the ctx draws its proposals from Philox4x32-10(key = seeds[t], ctr = (step, block,
 * chains[t])) instead of (the launch's seed, chain0 + t).  This is what lets the tiles of SEVERAL images share one launch --
 * the reference samples image after image (mpp_model.py:220-262), one kernel launch per image would leave most of the GPU
 * idle -- while every tile runs exactly the chain it would run in a launch of its own image.  n = number of chains of the
 * ctx; seeds == NULL or chains == NULL: back to mpp_run's arguments.  Reset by mpp_set_maps. */
int mpp_set_chain_keys(mpp_ctx *ctx, int n, const uint64_t *seeds, const uint32_t *chains);
int mpp_step_index(mpp_ctx *ctx, int tile, int64_t *step);
/* time of the last mpp_run / mpp_replay kernel in ms (HIP events on the ctx's stream) */
int mpp_last_kernel_ms(mpp_ctx *ctx, double *ms);

/* score-map epilogue of the two U-Nets (position_net/pos_net_model.py:186-200,338-346,
 * torch_div.py:8-43; shape_net/shape_net_model.py:139-141): all device pointers.
 * pos_out: [3][ldh][ldw] (vec0, vec1, mask logit; the network's padded output, of which the top-left
 * H x W region is used) -> det [H][W];
 * logits: [32][ldh][ldw] of one mark head -> marks [H][W][32], softmax over the 32 classes. */
int mpp_posnet_epilogue(mpp_ctx *ctx, int H, int W, int ldh, int ldw, const float *pos_out, double div_w,
                        double div_b, float *det);
int mpp_shapenet_epilogue(mpp_ctx *ctx, int H, int W, int ldh, int ldw, const float *logits, float *marks);
/* epilogue of one convolution of a DoubleConv block (model_parts/unet/unet_parts.py:12-31), in place on the
 * convolution's output x [planes][hw] (NCHW, plane p = channel p % C; float32 or bfloat16, device pointers):
 * x <- max(0, x * scale[c] + shift[c]) with scale = gamma / sqrt(var + eps) and
 * shift = beta + (conv_bias - mean) * scale, i.e. bias + BatchNorm(eval) + ReLU in one pass. */
int mpp_affine_relu(mpp_ctx *ctx, void *x, int planes, int C, int64_t hw, int elem_bytes, const float *scale,
                    const float *shift);

/* The two epilogues above on channels-last network outputs: pos_out [ldh][ldw][3], logits [ldh][ldw][32], elements
 * float32 (elem_bytes 4) or bfloat16 (2), device pointers; same arithmetic and outputs (det [H][W], marks [H][W][32]
 * float32).  A pixel's 32 logits are contiguous here, so the softmax is one coalesced pass with no transpose. */
int mpp_posnet_epilogue_nhwc(mpp_ctx *ctx, int H, int W, int ldh, int ldw, const void *pos_out, int elem_bytes, double div_w,
                             double div_b, float *det);
int mpp_shapenet_epilogue_nhwc(mpp_ctx *ctx, int H, int W, int ldh, int ldw, const void *logits, int elem_bytes, float *marks);

/* Channels-last (NHWC) glue between two convolutions of the U-Nets, ONE pass over the activation (device pointers,
 * float32 or bfloat16 elements; in_bytes / out_bytes = 4 or 2):
 *   y [H+2*pad][W+2*pad][C0+C1] <- reflect_pad( f( maxpool2x2( cat(x0 [..][C0], x1 [..][C1]) ) ) )
 * with f(v) = max(0, v*scale[c] + shift[c]) (bias + BatchNorm(eval) + ReLU folded as in mpp_affine_relu; identity when
 * scale == shift == NULL), pool != 0: the sources are [2H][2W] and are max-pooled 2x2 (after f), C1 == 0: no concat,
 * pad = 0 or 1 (reflect, the padding_mode of every 3x3 convolution).  Replaces F.pad(mode="reflect") + BatchNorm2d +
 * ReLU (model_parts/unet/unet_parts.py:12-31), MaxPool2d(2) (:34-45) and torch.cat([skip, up]) (:48-67).
 * y == x0 is allowed when pad == pool == C1 == 0 (in-place epilogue). */
int mpp_nhwc_glue(mpp_ctx *ctx, const void *x0, const void *x1, void *y, int H, int W, int C0, int C1, int pad, int pool,
                  int in_bytes, int out_bytes, const float *scale, const float *shift);

/* One Conv2d(C_in -> 32, kernel 3, padding 1, padding_mode='reflect') of a DoubleConv (model_parts/unet/unet_parts.py:12-31)
 * on the matrix cores (v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate), for the U-Nets' full-resolution level where
 * the library's N = 32 kernels are slowest: x0 [H][W][32] float32 channels-last; x1 = NULL, or the second half of the
 * concatenation cat([skip, up]) of `Up` (unet_parts.py:48-67) as a second [H][W][32] source (C_in = 64, no concatenated
 * copy); wp [C_in / 32][3*3][32 in][32 out] = weight[o][32 s + i][kh][kw] repacked; in_scale / in_shift [32] (or both NULL):
 * x0 <- max(0, x0 * in_scale + in_shift) at the load, i.e. the BatchNorm + ReLU of the layer that produced x0;
 * y [H][W][32] <- (relu ? max(0, .) : .)(conv * out_scale + out_shift) with this layer's folded bias + BatchNorm (or both
 * NULL).  Reflect padding is index arithmetic (no padded copy).  All device pointers, the ctx's stream. */
int mpp_conv3x3_c32(mpp_ctx *ctx, const float *x0, const float *x1, int H, int W, const float *wp, const float *in_scale,
                    const float *in_shift, const float *out_scale, const float *out_shift, int relu, float *y);

/* The stem of a U-Net: Conv2d(3 -> 32, 3x3, padding_mode='reflect') + folded bias / BatchNorm + ReLU
 * (model_parts/unet/unet_parts.py:12-31, first convolution of the first DoubleConv): x [H][W][3] float32 channels-last ->
 * y [H][W][32] = max(0, conv * scale + shift); wp [9 taps][3 in][32 out].  Device pointers, the ctx's stream. */
int mpp_conv3x3_stem(mpp_ctx *ctx, const float *x, int H, int W, const float *wp, const float *scale, const float *shift, float *y);

/* ShapeNet's three 1x1 heads, their biases and the softmax in one pass (model_parts/shape_net.py:12-46,
 * shape_net_model.py's inference softmax): h [ldh][ldw][32] float32 channels-last (the backbone's output) ->
 * marks_* [H][W][32] = softmax_c(sum_i w[k][c][i] * h[i] + b[k][c]), k = size, ratio, angle; w [3][32 classes][32 in],
 * b [3][32].  Replaces three library convolutions + bias adds + mpp_shapenet_epilogue_nhwc (same values up to float32
 * summation order).  All device pointers, 16-byte aligned; the ctx's stream. */
int mpp_shapenet_heads(mpp_ctx *ctx, int H, int W, int ldh, int ldw, const float *h, const float *w, const float *b,
                       float *marks_size, float *marks_ratio, float *marks_angle);

/* IoU matrix of convex quadrilaterals for the DOTA task-1 evaluation: a [n][8], b [m][8] (x1 y1 .. x4 y4, either
 * orientation) -> out [n][m] = |A_i n B_j| / (|A_i| + |B_j| - |A_i n B_j|), or -1 where the axis-aligned extents
 * (inclusive-pixel +1 convention) do not overlap.  Stands in for `polyiou.iou_poly` and the hbb pre-filter of
 * DOTA_devkit dota_evaluation_task1.voc_eval as called from metrics/dota_eval.py:37-47 (devkit not vendored,
 * README.md:22-30).  on_device != 0: a, b, out are device pointers. */
int mpp_quad_iou(mpp_ctx *ctx, int n, const double *a, int m, const double *b, double *out, int on_device);

void mpp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int mpp_abi_version(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
