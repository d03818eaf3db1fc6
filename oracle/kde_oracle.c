/*
 * kde_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, fp64 restatement of the algorithm of the reference
 * JuliaRobotics/KernelDensityEstimate.jl (v0.5.13) for the multiscale-Gibbs product hot path and
 * the tree/density layout it reads.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libkdehip.so) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - tree / density layout (okde_make_density): PINNED against the reference's own golden files
 *     test/testdata/test1DResult.txt, test2DResult.txt, test2DvarResult.txt, and the 100-point
 *     structures test1Dlcv100Result.txt, test2Dlcv100Result.txt, test2Dvarlcv100Result.txt
 *     (tests/golden/).
 *   - Gibbs arithmetic (okde_gibbs1): PARITY UNPINNED by any golden vector -- the reference holds
 *     none for this path and Julia is not available here to run it.  It is constrained only by the
 *     reference's statistical acceptance tests (test/runtests.jl:167-201, test/testPartialProd.jl)
 *     and closed-form invariants, all replayed in tests/.  oracle/julia_crosscheck.jl is the (un-run)
 *     script that pins it against the real reference wherever Julia is available.
 *   - direct evaluation + LOOCV bandwidth (okde_eval_direct, okde_auto_bandwidth): PINNED by the
 *     reference's golden test1Dlcv100Result.txt (kde!(x) at the reference's 1e-4, UnitTest1Dlcv01).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Node ids are the reference's 1-based ids; array element [id-1] stores node `id`.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, so a*b+c is never fused, as in Julia).
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include "kde_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define NO_CHILD (-1) /* src/BallTree01.jl:5 */

/* ------------------------------------------------------------------------------------------------
 * Tree + density construction
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
  int64_t D, N;
  double *centers, *ranges, *weights;
  int64_t *left, *right, *lo, *hi, *perm;
  double *means, *bw;
  int64_t next;
} okde_build;

/* src/BallTree01.jl:83 */
static int valid_index(const okde_build *t, int64_t i) { return (0 < i) && (i <= 2 * t->N); }

/* swapBall! src/BallTree01.jl:109-138 followed by swapDensity! src/BallTreeDensity01.jl:112-139
 * (uniform-bandwidth case: bandwidthMin/Max are not swapped, :128). */
static void swap_leaves(okde_build *t, int64_t i, int64_t j) {
  if (i == j) return;
  double tw = t->weights[i - 1]; t->weights[i - 1] = t->weights[j - 1]; t->weights[j - 1] = tw;
  int64_t tp = t->perm[i - 1]; t->perm[i - 1] = t->perm[j - 1]; t->perm[j - 1] = tp;
  const int64_t D = t->D;
  for (int64_t k = 0; k < D; ++k) {
    double c = t->centers[(i - 1) * D + k];
    t->centers[(i - 1) * D + k] = t->centers[(j - 1) * D + k];
    t->centers[(j - 1) * D + k] = c;
  }
  for (int64_t k = 0; k < D; ++k) {
    double m = t->means[(i - 1) * D + k];
    t->means[(i - 1) * D + k] = t->means[(j - 1) * D + k];
    t->means[(j - 1) * D + k] = m;
    double b = t->bw[(i - 1) * D + k];
    t->bw[(i - 1) * D + k] = t->bw[(j - 1) * D + k];
    t->bw[(j - 1) * D + k] = b;
  }
}

/* most_spread_coord src/BallTree01.jl:142-173.  NB the reference's point ranges end at
 * dims*(high-1), i.e. the LAST leaf `high` is excluded from both loops, while w = 1/(high-low). */
static int64_t most_spread_coord(const okde_build *t, int64_t low, int64_t high) {
  const int64_t D = t->D;
  double max_variance = 0.0;
  int64_t max_dim = 1;
  const double w = 1.0 / (double)(high - low);
  for (int64_t dim = 1; dim <= D; ++dim) {
    double mean = 0.0;
    for (int64_t p = low; p < high; ++p) mean = mean + w * t->centers[(p - 1) * D + (dim - 1)];
    double variance = 0.0;
    for (int64_t p = low; p < high; ++p) {
      double d = t->centers[(p - 1) * D + (dim - 1)] - mean;
      variance += d * d;
    }
    if (variance > max_variance) { max_variance = variance; max_dim = dim; }
  }
  return max_dim;
}

/* select! src/BallTree01.jl:223-242 (single forward scan quick-select around the middle leaf). */
static void select_leaves(okde_build *t, int64_t dim, int64_t position, int64_t low, int64_t high) {
  const int64_t D = t->D;
  while (low < high) {
    int64_t r = (low + high) / 2;
    swap_leaves(t, r, low);
    int64_t m = low;
    for (int64_t i = low; i <= high; ++i) {
      if (t->centers[(dim - 1) + D * (i - 1)] - t->centers[(dim - 1) + D * (low - 1)] < 0.0) {
        m += 1;
        swap_leaves(t, m, i);
      }
    }
    swap_leaves(t, low, m);
    if (m <= position) low = m + 1;
    if (m >= position) high = m - 1;
  }
}

/* calcStatsBall! src/BallTree01.jl:282-336 (+ getMiniMaxi :249-278) then
 * calcStatsDensity! src/BallTreeDensity01.jl:141-187 (uniform-bandwidth branch). */
static void calc_stats(okde_build *t, int64_t root) {
  const int64_t D = t->D;
  int64_t L = t->left[root - 1], R = t->right[root - 1];
  if (!valid_index(t, L) || !valid_index(t, R)) return;
  for (int64_t d = 0; d < D; ++d) {
    double a = t->centers[(L - 1) * D + d] + t->ranges[(L - 1) * D + d];
    double b = t->centers[(R - 1) * D + d] + t->ranges[(R - 1) * D + d];
    double maxi = (a > b) ? a : b;
    double c = t->centers[(L - 1) * D + d] - t->ranges[(L - 1) * D + d];
    double c2 = t->centers[(R - 1) * D + d] - t->ranges[(R - 1) * D + d];
    double mini = (c < c2) ? c : c2;
    double halfspan = (maxi - mini) / 2.0;
    t->ranges[(root - 1) * D + d] = halfspan;
    t->centers[(root - 1) * D + d] = mini + halfspan;
  }
  if (L != R) t->weights[root - 1] = t->weights[L - 1] + t->weights[R - 1];
  else t->weights[root - 1] = t->weights[L - 1];

  /* density part */
  double wtL = t->weights[L - 1], wtR = t->weights[R - 1];
  double wtT = wtL + wtR + DBL_EPSILON; /* eps(Float64) src/BallTreeDensity01.jl:161 */
  wtL /= wtT;
  wtR /= wtT;
  for (int64_t k = 0; k < D; ++k) {
    double mL = t->means[(L - 1) * D + k], mR = t->means[(R - 1) * D + k];
    double m = wtL * mL + wtR * mR;
    t->means[(root - 1) * D + k] = m;
    t->bw[(root - 1) * D + k] =
        wtL * (t->bw[(L - 1) * D + k] + mL * mL) + wtR * (t->bw[(R - 1) * D + k] + mR * mR) - m * m;
  }
}

/* buildBall! src/BallTree01.jl:342-411 */
static void build_ball(okde_build *t, int64_t low, int64_t high, int64_t root) {
  if (low == high) { /* N = 1 special case :351-362 */
    t->lo[root - 1] = low;
    t->hi[root - 1] = high;
    t->left[root - 1] = low;
    t->right[root - 1] = high;
    calc_stats(t, root);
    t->right[root - 1] = NO_CHILD;
    return;
  }
  int64_t coord = most_spread_coord(t, low, high);
  int64_t split = (low + high) / 2;
  select_leaves(t, coord, split, low, high);
  int64_t left, right;
  if (split <= low) left = low; else { left = t->next; t->next += 1; }
  if (split + 1 >= high) right = high; else { right = t->next; t->next += 1; }
  t->lo[root - 1] = low;
  t->hi[root - 1] = high;
  t->left[root - 1] = left;
  t->right[root - 1] = right;
  if (left != low) build_ball(t, low, split, left);
  if (right != high) build_ball(t, split + 1, high, right);
  calc_stats(t, root);
}

/* kde!(points, ks, weights) src/KDE01.jl:34-57 -> makeBallTreeDensity src/BallTreeDensity01.jl:192-231
 * -> makeBallTree src/BallTree01.jl:437-463 -> buildTree! :415-434. */
int okde_make_density(int64_t D, int64_t N, const double *points, const double *ks, int64_t nks,
                      const double *weights_in, double *centers, double *ranges, double *weights,
                      int64_t *left_child, int64_t *right_child, int64_t *lowest_leaf,
                      int64_t *highest_leaf, int64_t *permutation, double *means, double *bandwidth,
                      double *bandwidthMin, double *bandwidthMax) {
  if (D < 1 || N < 1) return -1;
  if (nks != 1 && nks != D) return -2;
  okde_build t;
  t.D = D; t.N = N;
  t.centers = centers; t.ranges = ranges; t.weights = weights;
  t.left = left_child; t.right = right_child; t.lo = lowest_leaf; t.hi = highest_leaf;
  t.perm = permutation; t.means = means; t.bw = bandwidth;

  /* makeBallTree: zeros / ones initialisation :447-456 */
  memset(centers, 0, sizeof(double) * (size_t)(2 * N * D));
  memset(ranges, 0, sizeof(double) * (size_t)(2 * N * D));
  memset(weights, 0, sizeof(double) * (size_t)(2 * N));
  memset(means, 0, sizeof(double) * (size_t)(2 * N * D));
  memset(bandwidth, 0, sizeof(double) * (size_t)(2 * N * D));
  for (int64_t i = 0; i < 2 * N; ++i) {
    left_child[i] = 1; right_child[i] = 1; lowest_leaf[i] = 1; highest_leaf[i] = 1; permutation[i] = 0;
  }
  /* weights ./ sum(weights) src/KDE01.jl:46 ; NULL = ones(Np) :67 */
  double wsum = 0.0;
  for (int64_t i = 0; i < N; ++i) wsum += weights_in ? weights_in[i] : 1.0;
  for (int64_t i = 0; i < N; ++i) weights[N + i] = (weights_in ? weights_in[i] : 1.0) / wsum;
  memcpy(centers + N * D, points, sizeof(double) * (size_t)(N * D));
  memcpy(means + N * D, points, sizeof(double) * (size_t)(N * D));
  /* ks -> ks.^2, length-1 ks repeated src/KDE01.jl:41-45 ; repeat over points BallTreeDensity01.jl:213-215 */
  for (int64_t i = 0; i < N; ++i)
    for (int64_t k = 0; k < D; ++k) {
      double s = (nks == 1) ? ks[0] : ks[k];
      double v = s * s;
      bandwidth[(N + i) * D + k] = v;
      bandwidthMin[i * D + k] = v;
      bandwidthMax[i * D + k] = v;
    }
  /* buildTree! src/BallTree01.jl:415-434 */
  for (int64_t j = 1; j <= N; ++j) {
    int64_t i = N + j;
    lowest_leaf[i - 1] = i;
    highest_leaf[i - 1] = i;
    left_child[i - 1] = i;
    right_child[i - 1] = NO_CHILD;
    permutation[i - 1] = j;
  }
  t.next = 2;
  build_ball(&t, N + 1, 2 * N, 1);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * gibbs1 (src/MSGibbs01.jl:527-629) and helpers
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
  int Ndens, Ndim, Nlevels;
  const okde_tree *trees;
  const uint8_t *mask; /* [Ndens*Ndim], 1 = active; NULL = all active */
  /* per-dimension manifold of the operator tuples addop / diffop / getMu / getLambda (src/MSGibbs01.jl:650-653): NULL or
   * 0 = Euclidean (the reference's defaults (+,), (-,), getEuclidMu, getEuclidLambda); 1 = CIRCULAR(2 pi).  The reference
   * repo defines no circular operators (its callers bring them): the semantic below is THIS repo's, stated in
   * include/kdehip.h "manifolds" -- the hook points are the reference's. */
  const uint8_t *manifold;
  double *particles, *variance; /* [Ndim x Ndens] column-major as the reference (:3-4) */
  double *p;
  int64_t *ind;
  double *Malmost, *Calmost;
  double *calclambdas, *calcmu;
  int64_t *levelList, *levelListNew; /* [Ndens x maxNp], row j contiguous here */
  int64_t *dNpts;
  int64_t maxNp;
  const double *randU, *randN;
  int64_t nU, nN;
  int64_t ruptr, rnptr; /* reference cursors (:23-24): 1-based, randU read BEFORE increment */
  int err;
  int32_t *labels; /* optional [Np][Ndens][Nlevels] = permutation of the label kept at each level */
} okde_glb;

static inline int mask_at(const okde_glb *g, int j, int d) { return g->mask ? g->mask[j * g->Ndim + d] : 1; }
static inline double t_mean(const okde_tree *t, int64_t i, int k) { return t->means[(i - 1) * t->ndim + k]; }
static inline double t_bw(const okde_tree *t, int64_t i, int k) { return t->bandwidth[(i - 1) * t->ndim + k]; }
static inline int t_valid(const okde_tree *t, int64_t i) { return (0 < i) && (i <= 2 * t->npts); }

/* updateGlbParticlesVariance! src/MSGibbs01.jl:89-115 */
static void update_particle(okde_glb *g, int j) {
  for (int d = 0; d < g->Ndim; ++d) {
    if (!mask_at(g, j, d)) {
      g->particles[d + g->Ndim * j] = 0.0;
      g->variance[d + g->Ndim * j] = 0.0;
    } else {
      g->particles[d + g->Ndim * j] = t_mean(&g->trees[j], g->ind[j], d);
      g->variance[d + g->Ndim * j] = t_bw(&g->trees[j], g->ind[j], d);
    }
  }
}

/* calcIndices! :123-130 */
static void calc_indices(okde_glb *g) { for (int j = 0; j < g->Ndens; ++j) update_particle(g, j); }

/* ---- the enumerated CIRCULAR(2 pi) operators (no reference counterpart: see okde_glb.manifold) ----
 * wrap(t) = t - 2 pi floor((t + pi) / (2 pi)) in [-pi, pi); diffop(a, b) = wrap(a - b); addop(a, b) = wrap(a + b);
 * getLambda = sum (as Euclidean); getMu(mus, lambdas, scale) = addop(ref, scale * sum_j lambda_j diffop(mu_j, ref)) with
 * ref = the mu of the FIRST contributing density (lambda_j > 0): the information-weighted mean in the tangent space at
 * ref, mapped back -- it equals the Euclidean formula whenever no difference wraps. */
static const double OKDE_TWO_PI = 6.283185307179586476925286766559, OKDE_PI = 3.141592653589793238462643383279;
static double circ_wrap(double t) { return t - OKDE_TWO_PI * floor((t + OKDE_PI) / OKDE_TWO_PI); }
static int is_circ(const okde_glb *g, int dim) { return g->manifold && g->manifold[dim] == 1; }

/* gaussianProductMeanCov! :176-216 with getEuclidLambda :141 and getEuclidMu :152-161 (or the circular pair above: the
 * hooks getLambda / getMu of :183-184, applied at :210-213).
 * skip is a 0-based density index, or -1 for "none". */
static void gaussian_product(okde_glb *g, int dim, double *destMu, double *destCov, int skip) {
  *destMu = 0.0;
  *destCov = 0.0;
  int any = 0;
  for (int j = 0; j < g->Ndens; ++j) if (j != skip && mask_at(g, j, dim)) any = 1;
  if (!any) return;
  for (int j = 0; j < g->Ndens; ++j) {
    if (j != skip && mask_at(g, j, dim)) {
      g->calclambdas[j] = 1.0 / g->variance[dim + g->Ndim * j];
      g->calcmu[j] = g->particles[dim + g->Ndim * j];
    } else {
      g->calclambdas[j] = 0.0;
      g->calcmu[j] = 0.0;
    }
  }
  double lam = 0.0;
  for (int j = 0; j < g->Ndens; ++j) lam += g->calclambdas[j];
  double cov = 1.0 / lam;
  *destCov = cov;
  if (is_circ(g, dim)) {
    int first = -1;
    for (int j = 0; j < g->Ndens && first < 0; ++j) if (g->calclambdas[j] > 0.0) first = j;
    const double ref = first >= 0 ? g->calcmu[first] : 0.0;
    double acc = 0.0;
    for (int j = 0; j < g->Ndens; ++j) acc += g->calclambdas[j] * circ_wrap(g->calcmu[j] - ref);
    *destMu = circ_wrap(ref + cov * acc);
    return;
  }
  double lambdamu = 0.0;
  for (int j = 0; j < g->Ndens; ++j) lambdamu += g->calcmu[j] * g->calclambdas[j];
  *destMu = cov * lambdamu;
}

/* How often the underflow branch of makeFasterSampleIndex! (:311-315) was taken since the last reset: lets the
 * tests compare the fallback behaviour of the GPU path's precisions with the reference arithmetic. */
static int64_t g_fallbacks = 0;
int64_t okde_fallback_count(int reset) {
  int64_t v = __atomic_load_n(&g_fallbacks, __ATOMIC_RELAXED);
  if (reset) __atomic_store_n(&g_fallbacks, 0, __ATOMIC_RELAXED);
  return v;
}

/* makeFasterSampleIndex! :250-328.  muValue has Ndim entries (caller applies the offset). */
static void make_sample_index(okde_glb *g, int j, const double *muValue, const double *covValue,
                              int doCalmost) {
  const okde_tree *t = &g->trees[j];
  const int D = g->Ndim;
  const int64_t n = g->dNpts[j];
  const int64_t *list = g->levelList + (size_t)j * g->maxNp;
  double pT = 0.0;
  int64_t zz = list[0];
  uint8_t dimmask[64];
  for (int d = 0; d < D; ++d) {
    dimmask[d] = 0;
    for (int k = 0; k < g->Ndens; ++k) if (k != j && mask_at(g, k, d)) dimmask[d] = 1;
  }
  for (int64_t z = 0; z < n; ++z) {
    double acc = 0.0;
    for (int i = 0; i < D; ++i) {
      if (!mask_at(g, j, i) || !dimmask[i]) continue;
      double tmpC = t_bw(t, zz, i);
      if (doCalmost) tmpC += covValue[i];
      double tmpM = t_mean(t, zz, i) - muValue[i];           /* diffop[i] :290 */
      if (is_circ(g, i)) tmpM = circ_wrap(tmpM);
      double distr = (tmpM * tmpM) / tmpC;
      if (!isnan(distr)) {
        acc += distr;
        acc += log(tmpC);
      }
    }
    double pz = exp(-0.5 * acc) * t->weights[zz - 1];
    if (isnan(pz)) pz = 0.0;
    g->p[z] = pz;
    pT += pz;
    if (z + 1 < n) zz = list[z + 1];
  }
  if (pT < 1e-99) { /* :311-315, zz is the last node of the level */
    __atomic_fetch_add(&g_fallbacks, 1, __ATOMIC_RELAXED); /* test instrumentation, not in the reference */
    double w = t->weights[zz - 1];
    pT = 0.0;
    for (int64_t z = 0; z < n; ++z) { g->p[z] = w; pT += w; }
  }
  for (int64_t z = 0; z < n; ++z) g->p[z] /= pT;
  for (int64_t z = 1; z < n; ++z) g->p[z] += g->p[z - 1];
}

/* selectLabelOnLevel :330-351 */
static void select_label(okde_glb *g, int j) {
  const int64_t dNp = g->dNpts[j];
  const int64_t *list = g->levelList + (size_t)j * g->maxNp;
  int64_t z = 1;
  int64_t zz = list[0];
  while (z <= dNp - 1) {
    if (g->ruptr < 1 || g->ruptr > g->nU) { g->err = OKDE_ERR_RANDU; break; } /* Julia BoundsError */
    if (g->randU[g->ruptr - 1] <= g->p[z - 1]) break;
    z += 1;
    if (z <= dNp) zz = list[z - 1];
  }
  g->ind[j] = zz;
  g->ruptr += 1;
}

/* sampleIndices! :364-385 */
static void sample_indices(okde_glb *g, const double *X) {
  for (int j = 0; j < g->Ndens; ++j) {
    make_sample_index(g, j, X, NULL, 0);
    select_label(g, j);
  }
  calc_indices(g);
}

/* sampleIndex :404-429 */
static void sample_index(okde_glb *g, int j) {
  for (int i = 0; i < g->Ndim; ++i) gaussian_product(g, i, &g->Malmost[i], &g->Calmost[i], j);
  make_sample_index(g, j, g->Malmost, g->Calmost, 1);
  select_label(g, j);
  update_particle(g, j);
}

/* samplePoint! :440-463 */
static void sample_point(okde_glb *g, double *X, int addEntropy) {
  for (int d = 0; d < g->Ndim; ++d) {
    double mn, vn;
    gaussian_product(g, d, &mn, &vn, -1);
    g->rnptr += 1;
    if (addEntropy) {
      if (g->rnptr < 1 || g->rnptr > g->nN) { g->err = OKDE_ERR_RANDN; X[d] = mn; continue; }
      X[d] = mn + sqrt(vn) * g->randN[g->rnptr - 1];         /* addop[dim] :456 */
      if (is_circ(g, d)) X[d] = circ_wrap(X[d]);
    } else {
      X[d] = mn;
    }
  }
}

/* levelInit! :467-475 and initIndices! :477-497 */
static void level_init(okde_glb *g) {
  for (int j = 0; j < g->Ndens; ++j) {
    g->dNpts[j] = 1;
    g->levelList[(size_t)j * g->maxNp] = 1; /* root() src/BallTree01.jl:64 */
  }
}
static void init_indices(okde_glb *g) {
  for (int j = 0; j < g->Ndens; ++j) {
    const int64_t dNp = g->dNpts[j];
    const int64_t *list = g->levelList + (size_t)j * g->maxNp;
    for (int64_t z = 0; z < dNp; ++z) g->p[z] = g->trees[j].weights[list[z] - 1];
    for (int64_t z = 1; z < dNp; ++z) g->p[z] += g->p[z - 1];
    select_label(g, j);
  }
}

/* levelDown! :500-523 */
static void level_down(okde_glb *g) {
  for (int j = 0; j < g->Ndens; ++j) {
    const okde_tree *t = &g->trees[j];
    int64_t *cur = g->levelList + (size_t)j * g->maxNp;
    int64_t *nxt = g->levelListNew + (size_t)j * g->maxNp;
    int64_t z = 0;
    for (int64_t y = 0; y < g->dNpts[j]; ++y) {
      int64_t node = cur[y];
      int64_t L = t->left_child[node - 1], R = t->right_child[node - 1];
      if (t_valid(t, L)) nxt[z++] = L;
      if (t_valid(t, R)) nxt[z++] = R;
      if (g->ind[j] == node) g->ind[j] = nxt[z - 1];
    }
    g->dNpts[j] = z;
  }
  int64_t *tmp = g->levelList; g->levelList = g->levelListNew; g->levelListNew = tmp;
}

int okde_nlevels(int64_t maxNp) {
  /* src/MSGibbs01.jl:568 : floor(Int, log(maxNp)/log(2) + 1) */
  return (int)floor(log((double)maxNp) / log(2.0) + 1.0);
}

static int glb_alloc(okde_glb *g, int Ndens, const okde_tree *trees, int ndims, const uint8_t *mask) {
  memset(g, 0, sizeof(*g));
  g->Ndens = Ndens; g->trees = trees; g->Ndim = ndims; g->mask = mask;
  int64_t maxNp = 0;
  for (int j = 0; j < Ndens; ++j) if (trees[j].npts > maxNp) maxNp = trees[j].npts;
  g->maxNp = maxNp;
  g->Nlevels = okde_nlevels(maxNp);
  g->particles = (double *)calloc((size_t)ndims * Ndens, sizeof(double));
  g->variance = (double *)calloc((size_t)ndims * Ndens, sizeof(double));
  g->p = (double *)calloc((size_t)maxNp, sizeof(double));
  g->ind = (int64_t *)calloc((size_t)Ndens, sizeof(int64_t));
  g->Malmost = (double *)calloc((size_t)ndims, sizeof(double));
  g->Calmost = (double *)calloc((size_t)ndims, sizeof(double));
  g->calclambdas = (double *)calloc((size_t)Ndens, sizeof(double));
  g->calcmu = (double *)calloc((size_t)Ndens, sizeof(double));
  g->levelList = (int64_t *)calloc((size_t)Ndens * maxNp, sizeof(int64_t));
  g->levelListNew = (int64_t *)calloc((size_t)Ndens * maxNp, sizeof(int64_t));
  g->dNpts = (int64_t *)calloc((size_t)Ndens, sizeof(int64_t));
  if (!g->particles || !g->variance || !g->p || !g->ind || !g->Malmost || !g->Calmost ||
      !g->calclambdas || !g->calcmu || !g->levelList || !g->levelListNew || !g->dNpts)
    return OKDE_ERR_ALLOC;
  for (int j = 0; j < Ndens; ++j) g->ind[j] = 1;
  return 0;
}
static void glb_free(okde_glb *g) {
  free(g->particles); free(g->variance); free(g->p); free(g->ind); free(g->Malmost); free(g->Calmost);
  free(g->calclambdas); free(g->calcmu); free(g->levelList); free(g->levelListNew); free(g->dNpts);
}

/* One output sample: body of the `for s in 1:Np` loop, src/MSGibbs01.jl:581-626. */
static void gibbs_one_sample(okde_glb *g, int64_t s, int Niter, double *pts, int64_t *ind_out,
                             int addEntropy) {
  double *X = pts + (size_t)s * g->Ndim; /* frm = (s-1)*Ndim :584 */
  level_init(g);
  init_indices(g);
  calc_indices(g);
  for (int l = 0; l < g->Nlevels; ++l) {
    sample_point(g, X, 1);
    level_down(g);
    sample_indices(g, X);
    for (int i = 0; i < Niter; ++i)
      for (int j = 0; j < g->Ndens; ++j) sample_index(g, j);
    if (g->labels)
      for (int j = 0; j < g->Ndens; ++j)
        g->labels[((size_t)s * g->Ndens + j) * g->Nlevels + l] =
            (int32_t)g->trees[j].permutation[g->ind[j] - 1];
  }
  for (int j = 0; j < g->Ndens; ++j) /* :612-616 */
    ind_out[(size_t)s * g->Ndens + j] = g->trees[j].permutation[g->ind[j] - 1] + 1;
  sample_point(g, X, addEntropy); /* :625 */
}

int64_t okde_randu_per_sample(int Ndens, int Nlevels, int Niter) {
  return (int64_t)Ndens * (1 + (int64_t)Nlevels * (Niter + 1));
}
int64_t okde_randn_per_sample(int ndims, int Nlevels) { return (int64_t)ndims * (Nlevels + 1); }

/* gibbs1 src/MSGibbs01.jl:527-629.  Samples [s_begin, s_end) of Np are produced; because the
 * reference's cursors advance by a data-independent amount per sample, sample s starts at
 * ruptr = s*K, rnptr = s*R -- identical to running the whole loop from s = 0. */
int okde_gibbs1_range(int Ndens, const okde_tree *trees, int64_t s_begin, int64_t s_end, int Niter,
                      double *pts, int64_t *ind, const double *randU, int64_t nU, const double *randN,
                      int64_t nN, int addEntropy, int ndims, const uint8_t *partialDimMask,
                      int32_t *labels) {
  if (Ndens < 1 || ndims < 1 || ndims > 64) return OKDE_ERR_ARG;
  for (int j = 0; j < Ndens; ++j)
    if (trees[j].ndim != ndims || trees[j].npts < 1) return OKDE_ERR_ARG;
  okde_glb g;
  int rc = glb_alloc(&g, Ndens, trees, ndims, partialDimMask);
  if (rc) { glb_free(&g); return rc; }
  g.randU = randU; g.nU = nU; g.randN = randN; g.nN = nN; g.labels = labels;
  const int64_t K = okde_randu_per_sample(Ndens, g.Nlevels, Niter);
  const int64_t R = okde_randn_per_sample(ndims, g.Nlevels);
  for (int64_t s = s_begin; s < s_end && !g.err; ++s) {
    g.ruptr = s * K;
    g.rnptr = s * R;
    gibbs_one_sample(&g, s, Niter, pts, ind, addEntropy);
  }
  rc = g.err;
  glb_free(&g);
  return rc;
}

/* gibbs1 with a per-dimension manifold (okde_glb.manifold: ndims bytes, 0 Euclidean / 1 circular; NULL = all Euclidean) */
int okde_gibbs1_manifold(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                         const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy,
                         int ndims, const uint8_t *partialDimMask, const uint8_t *manifold, int32_t *labels) {
  if (Ndens < 1 || ndims < 1 || ndims > 64) return OKDE_ERR_ARG;
  for (int j = 0; j < Ndens; ++j)
    if (trees[j].ndim != ndims || trees[j].npts < 1) return OKDE_ERR_ARG;
  if (manifold)
    for (int d = 0; d < ndims; ++d) if (manifold[d] > 1) return OKDE_ERR_ARG;
  okde_glb g;
  int rc = glb_alloc(&g, Ndens, trees, ndims, partialDimMask);
  if (rc) { glb_free(&g); return rc; }
  g.manifold = manifold;
  g.randU = randU; g.nU = nU; g.randN = randN; g.nN = nN; g.labels = labels;
  const int64_t K = okde_randu_per_sample(Ndens, g.Nlevels, Niter);
  const int64_t R = okde_randn_per_sample(ndims, g.Nlevels);
  for (int64_t s = 0; s < Np && !g.err; ++s) {
    g.ruptr = s * K;
    g.rnptr = s * R;
    gibbs_one_sample(&g, s, Niter, pts, ind, addEntropy);
  }
  rc = g.err;
  glb_free(&g);
  return rc;
}

int okde_gibbs1(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts, int64_t *ind,
                const double *randU, int64_t nU, const double *randN, int64_t nN, int addEntropy,
                int ndims, const uint8_t *partialDimMask, int32_t *labels) {
  return okde_gibbs1_range(Ndens, trees, 0, Np, Niter, pts, ind, randU, nU, randN, nN, addEntropy,
                           ndims, partialDimMask, labels);
}

/* Same work split over OpenMP threads (one private scratch per thread); used only to report an
 * all-cores CPU baseline.  Results are identical to okde_gibbs1 (samples are independent). */
int okde_gibbs1_omp(int Ndens, const okde_tree *trees, int64_t Np, int Niter, double *pts,
                    int64_t *ind, const double *randU, int64_t nU, const double *randN, int64_t nN,
                    int addEntropy, int ndims, const uint8_t *partialDimMask, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  int rc_all = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int t = 0; t < nthreads; ++t) {
    int64_t b = Np * t / nthreads, e = Np * (t + 1) / nthreads;
    int rc = okde_gibbs1_range(Ndens, trees, b, e, Niter, pts, ind, randU, nU, randN, nN, addEntropy,
                               ndims, partialDimMask, NULL);
    if (rc) {
#pragma omp critical
      rc_all = rc;
    }
  }
  return rc_all;
}

/* ------------------------------------------------------------------------------------------------
 * Direct KDE evaluation and LOOCV bandwidth selection
 * ---------------------------------------------------------------------------------------------- */

/* One kernel value as evalDirect computes it (src/DualTree01.jl:130-162 -> maxDistKer! ->
 * distGauss! :14-47 with minmaxFnc = bwMin, leaf ranges = 0, uniform bandwidth):
 * exp(-0.5 * sum_k |x_k - c_k|^2 / bw_k), bw = bandwidthMin[1..D] = the first leaf's variances. */
static double direct_kernel(const double *x, const double *c, const double *bw, int D) {
  double acc = 0.0;
  for (int k = 0; k < D; ++k) {
    double t = fabs(x[k] - c[k]);
    acc += (t * t) / bw[k];
  }
  return exp(-0.5 * acc);
}

/* evaluate(bd, locations, p, maxErr) with FORCE_EVAL_DIRECT = true (src/DualTree01.jl:303-346):
 * p[q] = sum_i w_i K(x_q, c_i) / norm, norm = (2 pi)^(D/2) * prod_k sqrt(bw_k) (:325-330).
 * loo != 0: the query points ARE the density's own points in ORIGINAL order (pos may be NULL);
 * the self term is skipped (:141) and the result divided by (1 - w_q) (:335). */
int okde_eval_direct(const okde_tree *bd, const double *pos, int64_t Nq, int loo, double *p) {
  const int D = (int)bd->ndim;
  const int64_t N = bd->npts;
  if (D < 1 || D > 64 || N < 1) return OKDE_ERR_ARG;
  const double *bw = bd->bandwidth + (size_t)N * D; /* bandwidthMin[1..D], src/BallTreeDensity01.jl:98,214 */
  double norm = pow(2.0 * M_PI, D / 2.0);
  for (int k = 0; k < D; ++k) norm *= sqrt(bw[k]);
  if (loo) {
    for (int64_t j = N + 1; j <= 2 * N; ++j) { /* locations in tree (leaf) order */
      const double *x = bd->means + (size_t)(j - 1) * D;
      double s = 0.0;
      for (int64_t i = N + 1; i <= 2 * N; ++i) {
        if (i == j) continue;
        s += direct_kernel(x, bd->means + (size_t)(i - 1) * D, bw, D) * bd->weights[i - 1];
      }
      p[bd->permutation[j - 1] - 1] = 0.5 * (s + s) / norm / (1.0 - bd->weights[j - 1]);
    }
    return 0;
  }
  for (int64_t q = 0; q < Nq; ++q) {
    const double *x = pos + (size_t)q * D;
    double s = 0.0;
    for (int64_t i = N + 1; i <= 2 * N; ++i)
      s += direct_kernel(x, bd->means + (size_t)(i - 1) * D, bw, D) * bd->weights[i - 1];
    p[q] = 0.5 * (s + s) / norm;
  }
  return 0;
}

/* A private, mutable copy of a 1-D (or D-dim) density used by the bandwidth search. */
typedef struct {
  int64_t D, N;
  double *centers, *ranges, *weights, *means, *bw, *bwMin, *bwMax;
  int64_t *left, *right, *lo, *hi, *perm;
} okde_owned;

static void owned_free(okde_owned *o) {
  free(o->centers); free(o->ranges); free(o->weights); free(o->means); free(o->bw); free(o->bwMin);
  free(o->bwMax); free(o->left); free(o->right); free(o->lo); free(o->hi); free(o->perm);
}
static int owned_make(okde_owned *o, int64_t D, int64_t N, const double *pts, const double *ks, int64_t nks,
                      const double *w) {
  memset(o, 0, sizeof(*o));
  o->D = D; o->N = N;
  size_t nd = (size_t)(2 * N * D), n2 = (size_t)(2 * N);
  o->centers = (double *)malloc(nd * sizeof(double)); o->ranges = (double *)malloc(nd * sizeof(double));
  o->means = (double *)malloc(nd * sizeof(double)); o->bw = (double *)malloc(nd * sizeof(double));
  o->weights = (double *)malloc(n2 * sizeof(double));
  o->bwMin = (double *)malloc((size_t)(N * D) * sizeof(double)); o->bwMax = (double *)malloc((size_t)(N * D) * sizeof(double));
  o->left = (int64_t *)malloc(n2 * sizeof(int64_t)); o->right = (int64_t *)malloc(n2 * sizeof(int64_t));
  o->lo = (int64_t *)malloc(n2 * sizeof(int64_t)); o->hi = (int64_t *)malloc(n2 * sizeof(int64_t));
  o->perm = (int64_t *)malloc(n2 * sizeof(int64_t));
  if (!o->centers || !o->ranges || !o->means || !o->bw || !o->weights || !o->bwMin || !o->bwMax || !o->left ||
      !o->right || !o->lo || !o->hi || !o->perm)
    return OKDE_ERR_ALLOC;
  return okde_make_density(D, N, pts, ks, nks, w, o->centers, o->ranges, o->weights, o->left, o->right, o->lo,
                           o->hi, o->perm, o->means, o->bw, o->bwMin, o->bwMax);
}
static okde_tree owned_view(const okde_owned *o) {
  okde_tree t;
  t.npts = o->N; t.ndim = o->D; t.means = o->means; t.bandwidth = o->bw; t.weights = o->weights;
  t.left_child = o->left; t.right_child = o->right; t.permutation = o->perm;
  return t;
}

/* entropy(bd) = -evalAvgLogL(bd, bd) (src/DualTree01.jl:450-474,505-508) with the leave-one-out
 * evaluation above; weights in original order (getWeights, src/KDE01.jl:127-136). */
static double loo_entropy(const okde_owned *o, double *scratch_p) {
  okde_tree t = owned_view(o);
  okde_eval_direct(&t, NULL, 0, 1, scratch_p);
  const int64_t N = o->N;
  /* W[perm] = weights[leaf]; any zero likelihood carrying weight -> -Inf */
  double ll = 0.0;
  int bad = 0;
  for (int64_t j = N + 1; j <= 2 * N; ++j) {
    int64_t q = o->perm[j - 1] - 1;
    if (scratch_p[q] == 0.0 && o->weights[j - 1] != 0.0) bad = 1;
  }
  if (bad) return INFINITY; /* H = -(-Inf) */
  /* (log.(L)')*W in original order */
  double *W = (double *)malloc((size_t)N * sizeof(double));
  for (int64_t j = N + 1; j <= 2 * N; ++j) W[o->perm[j - 1] - 1] = o->weights[j - 1];
  for (int64_t q = 0; q < N; ++q) {
    double L = scratch_p[q];
    if (L == 0.0) L = 1.0;
    ll += log(L) * W[q];
  }
  free(W);
  return -ll;
}

/* nLOO_LL (src/CrossValidation.jl:15-24): bandwidth *= alpha^2 (whole array, updateBandwidth! :5-12),
 * H = entropy, bandwidth /= alpha^2 -- the rounding drift of (b*a)/a is the reference's. */
static double nloo_ll(double alpha, okde_owned *o, double *scratch_p) {
  alpha = alpha * alpha;
  size_t nd = (size_t)(2 * o->N * o->D);
  for (size_t i = 0; i < nd; ++i) o->bw[i] = o->bw[i] * alpha;
  double H = loo_entropy(o, scratch_p);
  for (size_t i = 0; i < nd; ++i) o->bw[i] = o->bw[i] / alpha;
  return H;
}

/* golden (src/CrossValidation.jl:44-98) */
static double golden_search(okde_owned *o, double ax, double bx, double cx, double tol, double *scratch_p,
                            int *nevals) {
  const double C = (3.0 - sqrt(5.0)) / 2.0, R = 1.0 - C;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + C * (cx - bx); }
  else { x1 = bx - C * (bx - ax); x2 = bx; }
  double f1 = nloo_ll(x1, o, scratch_p), f2 = nloo_ll(x2, o, scratch_p);
  int k = 2;
  while (fabs(x3 - x0) > tol * (fabs(x1) + fabs(x2))) {
    if (f2 < f1) {
      x0 = x1; x1 = x2; x2 = R * x1 + C * x3; f1 = f2; f2 = nloo_ll(x2, o, scratch_p);
    } else {
      x3 = x2; x2 = x1; x1 = R * x2 + C * x0; f2 = f1; f1 = nloo_ll(x1, o, scratch_p);
    }
    ++k;
  }
  if (nevals) *nevals = k;
  return (f1 < f2) ? x1 : x2;
}

/* ksize (src/CrossValidation.jl:110-120) of a 1-D marginal with unit bandwidth, incl. neighborMinMax
 * (:100-108).  Returns the selected standard deviation. */
static int ksize_1d(int64_t N, const double *x, const double *w, double *ks_out, int *nevals) {
  okde_owned m;
  const double one = 1.0;
  int rc = owned_make(&m, 1, N, x, &one, 1, w); /* marginal(p,[i]) of p = kde!(points,[1.0]) */
  if (rc) { owned_free(&m); return rc; }
  /* neighborMinMax: ranges of nodes 1..N (first half), D = 1 */
  double maxm = sqrt((2.0 * m.ranges[0]) * (2.0 * m.ranges[0]));
  double minm = INFINITY;
  for (int64_t i = 0; i < N - 1; ++i) {
    double v = sqrt((2.0 * m.ranges[i]) * (2.0 * m.ranges[i]));
    if (v < minm) minm = v;
  }
  if (N - 1 < 1) minm = maxm; /* degenerate: a single point has no internal node below the root */
  if (minm < 1e-6) minm = 1e-6;
  /* getPoints / getWeights in original order */
  double *xo = (double *)malloc((size_t)N * sizeof(double)), *wo = (double *)malloc((size_t)N * sizeof(double));
  for (int64_t j = N + 1; j <= 2 * N; ++j) {
    xo[m.perm[j - 1] - 1] = m.centers[j - 1];
    wo[m.perm[j - 1] - 1] = m.weights[j - 1];
  }
  owned_free(&m);
  okde_owned p;
  const double mid = (minm + maxm) / 2.0;
  rc = owned_make(&p, 1, N, xo, &mid, 1, wo);
  if (rc) { owned_free(&p); free(xo); free(wo); return rc; }
  double *scratch = (double *)malloc((size_t)N * sizeof(double));
  double ks = golden_search(&p, 2.0 * minm / (minm + maxm), 1.0, 2.0 * maxm / (minm + maxm), 1e-2, scratch, nevals);
  ks = ks * (minm + maxm) / 2.0;
  /* npd = kde!(..., [ks], ...); getBW(npd)[1] = sqrt(ks^2) (src/KDE01.jl:45,118) */
  *ks_out = sqrt(ks * ks);
  free(scratch); free(xo); free(wo);
  owned_free(&p);
  return 0;
}

/* kde!(points) automatic bandwidth (src/KDE01.jl:3-27): per dimension, ksize of the 1-D marginal. */
int okde_auto_bandwidth(int64_t D, int64_t N, const double *points, double *bw_out, int *nevals_total) {
  if (D < 1 || N < 2) return OKDE_ERR_ARG;
  double *x = (double *)malloc((size_t)N * sizeof(double));
  if (!x) return OKDE_ERR_ALLOC;
  int total = 0;
  for (int64_t d = 0; d < D; ++d) {
    for (int64_t i = 0; i < N; ++i) x[i] = points[i * D + d];
    int ne = 0;
    int rc = ksize_1d(N, x, NULL, &bw_out[d], &ne);
    if (rc) { free(x); return rc; }
    total += ne;
  }
  if (nevals_total) *nevals_total = total;
  free(x);
  return 0;
}
