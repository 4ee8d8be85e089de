/*
 * mpp_oracle.h -- CPU restatement of the reference's MPP / RJMCMC sampling path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (libmppgpu.so) never links or calls it.
 *
 * Parity status: PINNED for the energy graph / dE / proposal densities / accept
 * rule by (i) the reference's own known answers (test/test_energy_graph.py,
 * test/test_interacting_points_set.py, test/test_points_set.py) and (ii) tapes
 * and energy cases recorded from the reference itself (tests/golden/*.npz,
 * generator tests/golden/make_golden.py).  UNPINNED at one boundary: the
 * rectangle-intersection area lives in shapely/GEOS, which is not in the
 * container; the recorded values used a Sutherland-Hodgman stand-in, so the
 * overlap term is pinned by analytic known answers instead (tests/test_geometry.py).
 * UNPINNED likewise: the rasteriser of the classic image energies (ORC_U_CONTRAST / ORC_U_GRADIENT) lives in
 * scikit-image 0.18.1, absent too; it is restated from that version's published algorithm, and everything above the
 * primitive is pinned by values, masks, outlines and a chain the reference produced over the same restatement
 * (tests/golden/classics_golden.npz, tape_contrast_96.npz).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#ifndef MPP_ORACLE_H
#define MPP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_UNIT 8
#define ORC_MAX_PAIR 2
#define ORC_NCLASS 32

/* unit term kinds */
enum {
  ORC_U_POSITION = 0,   /* p[0]=threshold : -2*(det[x,y]-thr)         data_energies.py:12-24        */
  ORC_U_SHAPE_REMAP = 1,/* p[0..2]=coef p[3..5]=icpt : mean_k(-2*sigmoid(coef_k*P_k+icpt_k)+1)
                           data_energies.py:27-45, energy_setup_legacy.py:142-147                     */
  ORC_U_MARK_NEG = 2,   /* p[0]=k : -P_k[x,y,class_k]            energy_setup_no_calibration.py:71   */
  ORC_U_MARK_REMAP = 3, /* p[0]=k p[1]=coef p[2]=icpt            (calib_marks=True)                   */
  ORC_U_AREA = 4,       /* p[0]=min p[1]=max : max(0,min-A,A-max) prior_energies.py:53-67             */
  ORC_U_RATIO_PRIOR = 5,/* p[0]=target : |target-ratio|           prior_energies.py:70-78             */
  ORC_U_CONST = 6,      /* p[0]=c  (the reference tests' TestUnitEnergy)                              */
  ORC_U_CONTRAST = 7,   /* p = {measure, dilation, gap, erode, thresh, fac, default}; measure: 0 lafarge 1 craciun
                           2 craciun2 3 mean 4 t-test 5 debug      energies/classics.py:100-196 (orc_set_image) */
  ORC_U_GRADIENT = 8    /* p = {thresh, eps}; the image holds np.gradient of the picture  classics.py:199-235 */
};
/* pair term kinds */
enum {
  ORC_P_OVERLAP = 0,    /* area(P1^P2)/(min(A1,A2)+1e-6)          prior_energies.py:11-24             */
  ORC_P_ALIGN = 1,      /* 1-|cos(a1-a2)|-rewarding ; p[0]=rewarding   prior_energies.py:27-50        */
  ORC_P_DIST_LE = 2,    /* 1 if d<=max_dist else 0  (test_energy_graph.py:26-37)                      */
  ORC_P_DIST_LT = 3     /* 1 if d< max_dist else 0  (test_interacting_points_set.py:30-42)            */
};
enum { ORC_REDUCE_MAX = 0, ORC_REDUCE_MIN = 1 };
/* combinators: E = sum_u F(lin0 + sum_k coef_k*g_k*v_k), g_k = gate if term is gated else 1,
 * gate = [v_{gate_term} <= gate_thr]  (hierarchical.py:21-32, :41-48; logistic.py:20-26) */
enum { ORC_C_LINEAR = 0, ORC_C_LOGISTIC = 1 };

typedef struct {
  int32_t kind, gated;
  double coef;
  double p[8];
} orc_unit_term;

typedef struct {
  int32_t kind, gated, reduce, _pad;
  double coef, max_dist;
  double p[2];
} orc_pair_term;

typedef struct {
  int32_t n_unit, n_pair, combinator, gate_term;
  double gate_thr, lin0;
  orc_unit_term unit[ORC_MAX_UNIT];
  orc_pair_term pair[ORC_MAX_PAIR];
} orc_model;

/* proposal kernels, make_kernels.py:50-177; order = the reference's kernel list */
enum {
  ORC_K_UBIRTH = 0, ORC_K_UDEATH, ORC_K_DBIRTH, ORC_K_DDEATH,
  ORC_K_GTRANS, ORC_K_DTRANS, ORC_K_GTRANSF, ORC_K_DTRANSF,
  ORC_K_SPLIT, ORC_K_MERGE,           /* split_and_merge_kernels.py:40-178, only with use_split_merge */
  ORC_NKERNEL
};

typedef struct {
  double p_kernel[ORC_NKERNEL];
  double intensity;          /* max(1,len(init)), sample_rjmcmc.py:68 */
  double sigma_trans;        /* 2     make_kernels.py:118 */
  double sigma_transform;    /* 0.1   make_kernels.py:130 */
  int32_t max_delta;         /* 8     make_kernels.py:124 */
  int32_t cyclic[3];
  double vmin[3], vmax[3];
  double edges[3][ORC_NCLASS];
  double split_radius;       /* 16    make_kernels.py:148 (pos_radius = merge_radius) */
  double split_sigma;        /* 0.1   make_kernels.py:150 (x mark range) */
} orc_kernels;

/* one tape record: a fully specified proposal (replay) */
typedef struct {
  int32_t kernel;
  int32_t target;            /* slot of the removed/moved point, -1 if none */
                             /* split: target = the split point, (aux0, aux1) = position delta, (as, ar, aa) =
                              * mark deltas.  merge: target = p0, param_id = p1 (-1: p0 has no neighbour) */
  int32_t ax, ay;            /* proposed point (birth / moved / transformed) */
  double as, ar, aa;
  double aux0, aux1;         /* raw normal deltas of the Gaussian kernels */
  int32_t param_id, new_class;
  double u_accept;
} orc_proposal;

typedef struct {
  double dE, fwd, bwd, log_alpha, T;
  int32_t accepted, n_after;
} orc_step_out;

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(int H, int W, const float *det, const float *m0, const float *m1, const float *m2,
                    const orc_model *model, const orc_kernels *kernels);
void orc_destroy(orc_ctx *c);
/* the image the classic energies read, float32 [H][W][C] (copied): C = 1 or 3 for ORC_U_CONTRAST, np.gradient of the
 * picture laid out [H][W][C/2][2] for ORC_U_GRADIENT */
int orc_set_image(orc_ctx *c, int C, const float *img);
int orc_contrast_masks(orc_ctx *c, const double rect[5], int dilation, int gap, int erode, int cap, int32_t *fill_rc,
                       int32_t *n_fill, int32_t *rim_rc, int32_t *n_rim);
int orc_outline(orc_ctx *c, const double rect[5], double eps, int cap, int32_t *rc, double *normals);
double orc_unit_value(orc_ctx *c, const orc_unit_term *t, const double rect[5]);
int orc_set_points(orc_ctx *c, int n, const int32_t *xy, const double *marks);
int orc_get_points(orc_ctx *c, int cap, int32_t *xy, double *marks);
int orc_count(orc_ctx *c);
/* per-point energy vectors [n][n_unit+n_pair] and combined energy (energy_graph.py:108-137) */
double orc_total_energy(orc_ctx *c, double *vectors_or_null);
/* dE of removing slots rem[] and adding rectangles (energy_graph.py:139-225) */
double orc_delta(orc_ctx *c, int n_rem, const int32_t *rem, int n_add, const int32_t *add_xy,
                 const double *add_marks);
/* Papangelou energy delta of every point, removed from the set (energy_point_set.py:102-116) */
void orc_papangelou(orc_ctx *c, double *out_dE);
void orc_set_temperature(orc_ctx *c, double T, double alpha, double T_target);
/* replay n tape records sequentially (rjmcmc.py:83-164 with the proposal given) */
int orc_replay(orc_ctx *c, int n, const orc_proposal *tape, orc_step_out *out);
/* native chain: proposals drawn from Philox4x32-10(key=seed, ctr=(step, block, chain)) */
int orc_run(orc_ctx *c, int64_t n_steps, uint64_t seed, uint32_t chain, orc_step_out *out_or_null,
            orc_proposal *props_or_null);
/* follow `tape` (its proposals, the oracle's own decisions) while recording the proposals the oracle would draw itself */
int orc_follow(orc_ctx *c, int n, uint64_t seed, uint32_t chain, const orc_proposal *tape, orc_step_out *out_or_null,
               orc_proposal *native_or_null);
int64_t orc_step_index(orc_ctx *c);
void orc_set_step_index(orc_ctx *c, int64_t step);
double orc_temperature(orc_ctx *c);
/* replay with every accept decision imposed (test resync after a tie within the dE tolerance) */
int orc_replay_forced(orc_ctx *c, int n, const orc_proposal *tape, const int32_t *accept, orc_step_out *out_or_null);
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_overlap(const double r1[5], const double r2[5]);
/* greedy distance NMS init (sample_rjmcmc.py:23-35, utils/nms.py:68-110); returns count */
int orc_naive_detection(orc_ctx *c, double threshold, double nms_dist, int cap, int32_t *xy, double *marks);

#ifdef __cplusplus
}
#endif
#endif
