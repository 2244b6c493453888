// TEST-ONLY host build of the per-point math the kernels use (dc_eig3.h, dc_pointmath.h).
// Lets the CPU test-suite pin the eigen-solver and the covariance / loss / backward-coefficient
// arithmetic against LAPACK and the oracle without a GPU.  Never loaded by depth_correction_amd.
#include "dc_common.h"
#include "dc_eig3.h"
#include "dc_pointmath.h"

extern "C" {

// cov: [n,6] (xx xy xz yy yz zz) -> lam [n,3], vec [n,9] (vec[i, k*3 + c] = component c of eigenvector k)
void dc_host_eig3(const double* cov, long n, double* lam, double* vec) {
  for (long i = 0; i < n; ++i) {
    const double* c = cov + i * 6;
    double V[3][3];
    dc::eig3_sym<double>(c[0], c[1], c[2], c[3], c[4], c[5], lam + i * 3, V);
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j) vec[i * 9 + k * 3 + j] = V[k][j];
  }
}

// round 4's full solver (what features_fwd_tile_kernel calls)
void dc_host_eig3_v2(const double* cov, long n, double* lam, double* vec) {
  for (long i = 0; i < n; ++i) {
    const double* c = cov + i * 6;
    double V[3][3];
    dc::eig3_sym_v2(c[0], c[1], c[2], c[3], c[4], c[5], lam + i * 3, V);
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j) vec[i * 9 + k * 3 + j] = V[k][j];
  }
}

// Hot-path variant: smallest eigenpair + trace only.
void dc_host_eig3_smallest(const double* cov, long n, double* lam0, double* v0, double* tr) {
  for (long i = 0; i < n; ++i) {
    const double* c = cov + i * 6;
    dc::eig3_smallest(c[0], c[1], c[2], c[3], c[4], c[5], lam0 + i, v0 + i * 3, tr + i);
  }
}

// round 2's solver (isolate-then-deflate with three cross products), still the A-B baseline of the step kernel
void dc_host_eig3_smallest_r2(const double* cov, long n, double* lam0, double* v0, double* tr) {
  for (long i = 0; i < n; ++i) {
    const double* c = cov + i * 6;
    dc::eig3_smallest_r2(c[0], c[1], c[2], c[3], c[4], c[5], lam0 + i, v0 + i * 3, tr + i);
  }
}

// the slimmer solver of the one-pass step kernel (trace-1 core: adjugate eigenvector, one reciprocal)
void dc_host_eig3_smallest_v2(const double* cov, long n, double* lam0, double* v0, double* tr) {
  for (long i = 0; i < n; ++i) {
    const double* c = cov + i * 6;
    dc::eig3_smallest_v2(c[0], c[1], c[2], c[3], c[4], c[5], lam0 + i, v0 + i * 3, tr + i);
  }
}

// One neighbourhood per row of nbr [n,k]; points [np,3].  Outputs per centre: mean[3], cov6[6], lam[3], v0[3],
// loss, c1, c2 (loss / backward coefficients for an unmasked point with zero offset).
void dc_host_neighbourhoods(const double* points, const int* nbr, long n, int k, double scale, int loss_kind,
                            int normalization, int sqrt_, double* mean, double* cov6, double* lam, double* v0,
                            double* loss, double* c1, double* c2) {
  dc::LossParams lp{loss_kind, normalization, sqrt_};
  for (long i = 0; i < n; ++i) {
    const double* xi = points + i * 3;
    dc::CovAcc acc;
    dc::cov_init(acc);
    for (int q = 0; q < k; ++q) {
      const int j = nbr[i * k + q];
      if (j < 0) continue;
      const double* xj = points + (long)j * 3;
      dc::cov_add(acc, xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2], 1.0);
    }
    double moff[3], cm[3], C[6], D, omega, V[3][3], l[3];
    dc::cov_finish(acc, scale, moff, cm, C, &D, &omega);
    dc::eig3_sym<double>(C[0], C[1], C[2], C[3], C[4], C[5], l, V);
    for (int a = 0; a < 3; ++a) { mean[i * 3 + a] = xi[a] + moff[a]; lam[i * 3 + a] = l[a]; v0[i * 3 + a] = V[0][a]; }
    for (int a = 0; a < 6; ++a) cov6[i * 6 + a] = C[a];
    loss[i] = dc::loss_and_coeffs(lp, l[0], l[0] + l[1] + l[2], D, 0.0, true, c1 + i, c2 + i);
  }
}

double dc_host_model_depth(int kind, int n_terms, const double* w, const double* e, double depth, double inc, int in_mask) {
  dc::ModelParams mp;
  mp.kind = kind; mp.n_terms = n_terms;
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) { mp.w[k] = k < n_terms ? w[k] : 0.0; mp.e[k] = k < n_terms ? e[k] : 0.0; }
  return dc::model_depth(mp, depth, inc, in_mask != 0);
}

void dc_host_normal_inc(const double* dir, const double* v0, double* normal, double* inc) {
  dc::normal_and_incidence(dir, v0, normal, inc);
}

}  // extern "C"
