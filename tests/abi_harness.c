/* abi_harness.c -- the C view of include/kdehip.h, compiled as C11 (gcc -std=c11 -pedantic -Wall -Werror), not
 * C++: what a Julia `ccall` (or any other C FFI) binds.  Layout assertions at compile time; at run time the
 * hand-checkable trace of SURVEY.md Appendix A goes through kdehip_make_density (host) and, when a device is
 * present, through kdehip_gibbs1 and kdehip_gibbs1_trace; without a device every compute entry must fail with
 * KDEHIP_ERR_NO_DEVICE (no CPU fallback).  Built and run by tests/test_abi_harness.py.
 * Exit code 0 = all checks passed; the last line printed says which mode ran ("device" or "no-device"). */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "kdehip.h"

_Static_assert(sizeof(kdehip_density) == 64, "kdehip_density is eight 8-byte fields");
_Static_assert(offsetof(kdehip_density, npts) == 0, "npts");
_Static_assert(offsetof(kdehip_density, ndim) == 8, "ndim");
_Static_assert(offsetof(kdehip_density, means) == 16, "means");
_Static_assert(offsetof(kdehip_density, bandwidth) == 24, "bandwidth");
_Static_assert(offsetof(kdehip_density, weights) == 32, "weights");
_Static_assert(offsetof(kdehip_density, left_child) == 40, "left_child");
_Static_assert(offsetof(kdehip_density, right_child) == 48, "right_child");
_Static_assert(offsetof(kdehip_density, permutation) == 56, "permutation");
_Static_assert(sizeof(kdehip_product_info_t) == 48, "kdehip_product_info_t");
_Static_assert(offsetof(kdehip_product_info_t, nodes_per_sweep) == 16, "nodes_per_sweep");
_Static_assert(offsetof(kdehip_product_info_t, fast_math_path) == 40, "fast_math_path");
_Static_assert(KDEHIP_OK == 0 && KDEHIP_ERR_NO_DEVICE == -4 && KDEHIP_ERR_UNSUPPORTED == -7, "error codes");

#define N 3
#define NODES (2 * N)

typedef struct {
  double centers[NODES], ranges[NODES], weights[NODES], means[NODES], bandwidth[NODES], bwmin[N], bwmax[N];
  int64_t left[NODES], right[NODES], lo[NODES], hi[NODES], perm[NODES];
} dens1d;

static int fails = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      ++fails;                                            \
      printf("FAIL %s:%d: ", __FILE__, __LINE__);         \
      printf(__VA_ARGS__);                                \
      printf("\n");                                       \
    }                                                     \
  } while (0)

static void build(const double *pts, double ks, dens1d *d, kdehip_density *c) {
  const int rc = kdehip_make_density(1, N, pts, &ks, 1, NULL, d->centers, d->ranges, d->weights, d->left, d->right,
                                     d->lo, d->hi, d->perm, d->means, d->bandwidth, d->bwmin, d->bwmax);
  CHECK(rc == KDEHIP_OK, "kdehip_make_density rc=%d (%s)", rc, kdehip_last_error());
  c->npts = N; c->ndim = 1;
  c->means = d->means; c->bandwidth = d->bandwidth; c->weights = d->weights;
  c->left_child = d->left; c->right_child = d->right; c->permutation = d->perm;
}

int main(void) {
  /* A = kde!([0,1,3],[0.5]), B = kde!([0.2,2,2.5],[0.4]) */
  const double a_pts[N] = {0.0, 1.0, 3.0}, b_pts[N] = {0.2, 2.0, 2.5};
  dens1d A, B;
  kdehip_density trees[2];
  build(a_pts, 0.5, &A, &trees[0]);
  build(b_pts, 0.4, &B, &trees[1]);
  /* tree arrays of the appendix (1-based node order 1..6; slot 3 is the unused slot N) */
  const double a_means[NODES] = {4.0 / 3.0, 0.5, 0.0, 0.0, 1.0, 3.0};
  const double a_bw[NODES] = {1.8055555555555556, 0.5, 0.0, 0.25, 0.25, 0.25};
  const double b_means[NODES] = {1.5666666666666667, 1.1, 0.0, 0.2, 2.0, 2.5};
  const double b_bw[NODES] = {1.1355555555555557, 0.97, 0.0, 0.16, 0.16, 0.16};
  const int64_t left[NODES] = {2, 4, 1, 4, 5, 6}, right[NODES] = {6, 5, 1, -1, -1, -1}, perm[NODES] = {0, 0, 0, 1, 2, 3};
  for (int i = 0; i < NODES; ++i) {
    CHECK(fabs(A.means[i] - a_means[i]) < 1e-12 && fabs(A.bandwidth[i] - a_bw[i]) < 1e-12, "A node %d", i + 1);
    CHECK(fabs(B.means[i] - b_means[i]) < 1e-12 && fabs(B.bandwidth[i] - b_bw[i]) < 1e-12, "B node %d", i + 1);
    CHECK(A.left[i] == left[i] && A.right[i] == right[i] && A.perm[i] == perm[i], "A topology node %d", i + 1);
    CHECK(B.left[i] == left[i] && B.right[i] == right[i] && B.perm[i] == perm[i], "B topology node %d", i + 1);
  }

  /* streams of the appendix: Nout = 2, Niter = 1 => L = 2, K = 10, R = 3 */
  enum { NP = 2, NITER = 1, NU = 24, NN = 6, L = 2 };
  double randU[NU];
  for (int i = 0; i < NU; ++i) randU[i] = (double)((37 * i + 11) % 100) / 100.0 + 0.005;
  const double randN[NN] = {-1.25, 0.5, -0.5, 1.25, 0.25, -0.75};
  double pts[NP];
  int64_t ind[2 * NP];
  int32_t labels[NP * 2 * L];

  const int ndev = kdehip_device_count();
  CHECK(kdehip_version() == KDEHIP_VERSION, "version %d", kdehip_version());
  if (ndev < 1) {
    /* no device: loud failure, never a CPU result */
    const int rc = kdehip_gibbs1(2, trees, NP, NITER, pts, ind, randU, NU, randN, NN, 1, 1, NULL, 0);
    CHECK(rc == KDEHIP_ERR_NO_DEVICE, "gibbs1 without a device returned %d", rc);
    CHECK(strlen(kdehip_last_error()) > 0, "empty error message");
    kdehip_product *plan = NULL;
    CHECK(kdehip_product_create(&plan, 2, trees, 1, NULL, 64, 0) == KDEHIP_ERR_NO_DEVICE && plan == NULL, "plan without a device");
    printf("%s: no-device\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
  }

  /* addEntropy = false: precision-weighted means of the selected leaves */
  int rc = kdehip_gibbs1(2, trees, NP, NITER, pts, ind, randU, NU, randN, NN, 0, 1, NULL, 0);
  CHECK(rc == KDEHIP_OK, "gibbs1 rc=%d (%s)", rc, kdehip_last_error());
  CHECK(ind[0] == 3 && ind[1] == 2 && ind[2] == 3 && ind[3] == 3, "indices %lld %lld %lld %lld", (long long)ind[0],
        (long long)ind[1], (long long)ind[2], (long long)ind[3]);
  CHECK(fabs(pts[0] - 0.5121951219512195) < 1e-14 && fabs(pts[1] - 1.6097560975609757) < 1e-14, "points %.17g %.17g", pts[0], pts[1]);
  /* addEntropy = true + label trace */
  memset(labels, 0xFF, sizeof labels);
  rc = kdehip_gibbs1_trace(2, trees, NP, NITER, pts, ind, randU, NU, randN, NN, 1, 1, NULL, 0, labels);
  CHECK(rc == KDEHIP_OK, "gibbs1_trace rc=%d (%s)", rc, kdehip_last_error());
  CHECK(fabs(pts[0] - 0.3560213600626134) < 1e-14 && fabs(pts[1] - 1.3754954547280667) < 1e-14, "points %.17g %.17g", pts[0], pts[1]);
  /* the trace's last level holds permutation[ind_j] = indices - 1 (src/MSGibbs01.jl:612-616) */
  for (int s = 0; s < NP; ++s)
    for (int j = 0; j < 2; ++j)
      CHECK(labels[(s * 2 + j) * L + (L - 1)] == (int32_t)(ind[s * 2 + j] - 1), "trace s=%d j=%d: %d", s, j, labels[(s * 2 + j) * L + (L - 1)]);
  /* short random streams are the reference's BoundsError */
  rc = kdehip_gibbs1(2, trees, NP, NITER, pts, ind, randU, 5, randN, NN, 1, 1, NULL, 0);
  CHECK(rc == KDEHIP_ERR_RAND_SHORT, "short randU rc=%d", rc);
  /* resident plan through C */
  kdehip_product *plan = NULL;
  rc = kdehip_product_create(&plan, 2, trees, 1, NULL, 64, 0);
  CHECK(rc == KDEHIP_OK && plan != NULL, "plan rc=%d", rc);
  if (plan) {
    kdehip_product_info_t info;
    CHECK(kdehip_product_info(plan, &info) == KDEHIP_OK && info.ndens == 2 && info.ndims == 1 && info.nlevels == L && info.precision == 64, "plan info");
    CHECK(kdehip_product_randu_per_sample(plan, NITER) == 10 && kdehip_product_randn_per_sample(plan) == 3, "K, R");
    double p2[NP];
    int64_t i2[2 * NP];
    rc = kdehip_product_sample_philox_host(plan, NP, NITER, 42u, 0, 1, p2, i2, NULL);
    CHECK(rc == KDEHIP_OK, "philox run rc=%d (%s)", rc, kdehip_last_error());
    /* the same draws through the host twin of the device RNG and the streams entry */
    double u[NP * 10], n[NP * 3];
    kdehip_philox_fill_uniform(42u, 0, NP, 10, u);
    kdehip_philox_fill_normal(42u, 0, NP, 3, n);
    rc = kdehip_gibbs1(2, trees, NP, NITER, pts, ind, u, NP * 10, n, NP * 3, 1, 1, NULL, 0);
    CHECK(rc == KDEHIP_OK, "gibbs1 on the twin streams rc=%d", rc);
    for (int s = 0; s < NP; ++s) {
      CHECK(pts[s] == p2[s], "philox vs streams point %d: %.17g %.17g", s, pts[s], p2[s]);
      CHECK(ind[2 * s] == i2[2 * s] && ind[2 * s + 1] == i2[2 * s + 1], "philox vs streams labels %d", s);
    }
    kdehip_product_destroy(plan);
  }
  printf("%s: device\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
