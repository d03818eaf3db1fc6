/*
 * kde_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).  See kde_oracle.c.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef KDE_ORACLE_H
#define KDE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  OKDE_OK = 0,
  OKDE_ERR_ARG = -1,
  OKDE_ERR_ALLOC = -2,
  OKDE_ERR_RANDU = -3, /* randU too short (Julia BoundsError) */
  OKDE_ERR_RANDN = -4  /* randN too short */
};

/* The six flat arrays of a BallTreeDensity that gibbs1 reads (reference 1-based node ids;
 * src/BallTreeDensity01.jl:11-24, src/BallTree01.jl:10-28). */
typedef struct okde_tree {
  int64_t npts;
  int64_t ndim;
  const double *means;         /* [ndim*2*npts] */
  const double *bandwidth;     /* [ndim*2*npts] variances */
  const double *weights;       /* [2*npts] */
  const int64_t *left_child;   /* [2*npts] */
  const int64_t *right_child;  /* [2*npts], NO_CHILD = -1 */
  const int64_t *permutation;  /* [2*npts] */
} okde_tree;

/* kde!(points, ks, weights): all output arrays are caller-allocated
 * (centers/ranges/means/bandwidth: D*2N; weights and index arrays: 2N; bandwidthMin/Max: D*N). */
int okde_make_density(int64_t D, int64_t N, const double *points, const double *ks, int64_t nks,
                      const double *weights_in, double *centers, double *ranges, double *weights,
                      int64_t *left_child, int64_t *right_child, int64_t *lowest_leaf,
                      int64_t *highest_leaf, int64_t *permutation, double *means, double *bandwidth,
                      double *bandwidthMin, double *bandwidthMax);

int okde_nlevels(int64_t maxNp);
int64_t okde_randu_per_sample(int Ndens, int Nlevels, int Niter);
int64_t okde_randn_per_sample(int ndims, int Nlevels);

/* gibbs1: pts [ndims*Np] column-major, ind [Ndens*Np] column-major (value = permutation + 1).
 * partialDimMask [Ndens*ndims] (1 = active) or NULL.  labels optional [Np][Ndens][Nlevels] or NULL. */
int okde_gibbs1(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy,
                int ndims, const uint8_t *partialDimMask, int32_t *labels);

/* gibbs1 with the operator tuples addop / diffop / getMu / getLambda (src/MSGibbs01.jl:650-653) given as a per-dimension
 * ENUM: manifold[d] = 0 Euclidean (the reference's defaults), 1 = circular (2 pi) -- wrap to [-pi, pi), tangent-space mean at
 * the first contributing kernel; this repo's stated semantic, applied at the reference's hook points (:290, :183-184 /
 * :210-213, :456).  manifold == NULL: identical to okde_gibbs1. */
int okde_gibbs1_manifold(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                         const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy,
                         int ndims, const uint8_t *partialDimMask, const uint8_t *manifold, int32_t *labels);
int okde_gibbs1_range(int Ndens, const okde_tree *trees, int64_t s_begin, int64_t s_end, int Niter,
                      double *pts, int64_t *ind, const double *randU, int64_t nU, const double *randN,
                      int64_t nN, int addEntropy, int ndims, const uint8_t *partialDimMask,
                      int32_t *labels);

int okde_gibbs1_omp(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts,
                    int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                    int addEntropy, int ndims, const uint8_t *partialDimMask, int nthreads);

/* Number of times makeFasterSampleIndex! took its underflow branch (pT < 1e-99, :311-315) since the last reset
 * (process-wide; reset != 0 clears it). */
int64_t okde_fallback_count(int reset);

/* Direct evaluation (evalDirect, FORCE_EVAL_DIRECT = true): p[q] for pos (D x Nq, column-major); with
 * loo != 0 the density's own points in original order, leave-one-out (pos ignored). */
int okde_eval_direct(const okde_tree *bd, const double *pos, int64_t Nq, int loo, double *p);
/* kde!(points) automatic LOOCV bandwidth: bw_out[D] standard deviations. */
int okde_auto_bandwidth(int64_t D, int64_t N, const double *points, double *bw_out, int *nevals_total);

#ifdef __cplusplus
}
#endif
#endif
