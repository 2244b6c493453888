// Symmetric 3x3 eigen-decomposition for per-neighbourhood covariance matrices.
//
// Replaces torch.linalg.eigh (LAPACK syevd) at reference depth_cloud.py:376-399.  Requirements taken
// from the reference's only known-answer test (loss.py:714-735: eigenvalues atol 1e-6, eigenvectors
// atol 1e-5 up to sign) and from BASELINE.json (eigenvalues within 1e-5 relative): the absolute error
// must stay O(eps * |C|) for EVERY eigenvalue, including clustered small ones (collinear lidar rings
// give l0 ~ l1 << l2), where the plain trigonometric closed form loses half the digits.
//
// Method: scale by max|c_ij|; trigonometric roots of the characteristic cubic pick the ISOLATED
// eigenvalue (largest if det(B) >= 0, else smallest), whose value and eigenvector (largest cross
// product of two rows of C - l*I) are well conditioned; the remaining pair comes from the exact 2x2
// problem in the orthogonal complement, so a cluster is resolved at its own scale.  No iteration, no
// data-dependent loop: one wavefront's lanes stay converged except for two short selects.
#pragma once
#include "dc_common.h"
#include <math.h>

namespace dc {

template <typename R> DC_HD R rsqrt_(R x) { return R(1) / sqrt(x); }

template <typename R>
DC_HD void cross3(const R* a, const R* b, R* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// Unit eigenvector of the symmetric matrix for a well separated eigenvalue `lam`.
template <typename R>
DC_HD void eigvec_isolated(R a00, R a01, R a02, R a11, R a12, R a22, R lam, R* v) {
  R r0[3] = {a00 - lam, a01, a02};
  R r1[3] = {a01, a11 - lam, a12};
  R r2[3] = {a02, a12, a22 - lam};
  R c0[3], c1[3], c2[3];
  cross3(r0, r1, c0);
  cross3(r0, r2, c1);
  cross3(r1, r2, c2);
  R d0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  R d1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  R d2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
  R dm = d0;
  R s0 = c0[0], s1 = c0[1], s2 = c0[2];
  if (d1 > dm) { dm = d1; s0 = c1[0]; s1 = c1[1]; s2 = c1[2]; }
  if (d2 > dm) { dm = d2; s0 = c2[0]; s1 = c2[1]; s2 = c2[2]; }
  if (dm > R(0)) {
    R inv = rsqrt_(dm);
    v[0] = s0 * inv; v[1] = s1 * inv; v[2] = s2 * inv;
  } else {
    v[0] = R(1); v[1] = R(0); v[2] = R(0);
  }
}

// lam[0] <= lam[1] <= lam[2]; V[k][0..2] = unit eigenvector of lam[k] (sign arbitrary).
template <typename R>
DC_HD void eig3_sym(R a00, R a01, R a02, R a11, R a12, R a22, R* lam, R (*V)[3]) {
  R m = fmax(fmax(fabs(a00), fabs(a11)), fabs(a22));
  m = fmax(m, fmax(fabs(a01), fmax(fabs(a02), fabs(a12))));
  if (!(m > R(0)) || !(m < R(INFINITY))) {   // zero matrix, or NaN/inf input: propagate like LAPACK would not throw
    R z = (m == R(0)) ? R(0) : R(NAN);
    lam[0] = lam[1] = lam[2] = z;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? R(1) : R(0);
    return;
  }
  const R inv_m = R(1) / m;
  a00 *= inv_m; a01 *= inv_m; a02 *= inv_m; a11 *= inv_m; a12 *= inv_m; a22 *= inv_m;

  const R off = a01 * a01 + a02 * a02 + a12 * a12;
  const R q = (a00 + a11 + a22) * R(1.0 / 3.0);
  const R b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const R p2 = (b00 * b00 + b11 * b11 + b22 * b22 + R(2) * off) * R(1.0 / 6.0);
  R iso[3], u[3], w[3];
  R l_iso, l_lo, l_hi, ca, cb;
  bool iso_is_max;
  if (!(p2 > R(0))) {
    // multiple of the identity
    lam[0] = lam[1] = lam[2] = q * m;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? R(1) : R(0);
    return;
  }
  {
    const R p = sqrt(p2);
    const R c00 = b11 * b22 - a12 * a12;
    const R c01 = a01 * b22 - a12 * a02;
    const R c02 = a01 * a12 - b11 * a02;
    const R det = (b00 * c00 - a01 * c01 + a02 * c02) / (p * p2);
    R half = fmin(fmax(det * R(0.5), R(-1)), R(1));
    const R ang = acos(half) * R(1.0 / 3.0);
    iso_is_max = half >= R(0);
    // beta2 = 2cos(ang) (largest), beta0 = 2cos(ang + 2pi/3) (smallest)
    const R beta = iso_is_max ? R(2) * cos(ang) : R(2) * cos(ang + R(2.0943951023931954923));
    l_iso = q + p * beta;
  }
  eigvec_isolated(a00, a01, a02, a11, a12, a22, l_iso, iso);
  // orthonormal basis (u, w) of the complement of iso
  if (fabs(iso[0]) > fabs(iso[1])) {
    R inv = rsqrt_(iso[0] * iso[0] + iso[2] * iso[2]);
    u[0] = -iso[2] * inv; u[1] = R(0); u[2] = iso[0] * inv;
  } else {
    R inv = rsqrt_(iso[1] * iso[1] + iso[2] * iso[2]);
    u[0] = R(0); u[1] = iso[2] * inv; u[2] = -iso[1] * inv;
  }
  cross3(iso, u, w);
  // A u, A w, A iso
  const R au0 = a00 * u[0] + a01 * u[1] + a02 * u[2];
  const R au1 = a01 * u[0] + a11 * u[1] + a12 * u[2];
  const R au2 = a02 * u[0] + a12 * u[1] + a22 * u[2];
  const R aw0 = a00 * w[0] + a01 * w[1] + a02 * w[2];
  const R aw1 = a01 * w[0] + a11 * w[1] + a12 * w[2];
  const R aw2 = a02 * w[0] + a12 * w[1] + a22 * w[2];
  const R ai0 = a00 * iso[0] + a01 * iso[1] + a02 * iso[2];
  const R ai1 = a01 * iso[0] + a11 * iso[1] + a12 * iso[2];
  const R ai2 = a02 * iso[0] + a12 * iso[1] + a22 * iso[2];
  l_iso = iso[0] * ai0 + iso[1] * ai1 + iso[2] * ai2;          // Rayleigh quotient
  const R m00 = u[0] * au0 + u[1] * au1 + u[2] * au2;
  const R m01 = u[0] * aw0 + u[1] * aw1 + u[2] * aw2;
  const R m11 = w[0] * aw0 + w[1] * aw1 + w[2] * aw2;
  // 2x2 symmetric eigenproblem
  const R h = (m00 - m11) * R(0.5);
  const R mean = (m00 + m11) * R(0.5);
  const R rad = sqrt(h * h + m01 * m01);
  l_hi = mean + rad;
  l_lo = mean - rad;
  // eigenvector of l_hi in (u, w) coordinates: the better conditioned of the two row null vectors
  {
    R x1 = m01, y1 = l_hi - m00;      // from row 0: (m00 - l) x + m01 y = 0
    R x2 = l_hi - m11, y2 = m01;      // from row 1
    R n1 = x1 * x1 + y1 * y1, n2 = x2 * x2 + y2 * y2;
    R x = n1 >= n2 ? x1 : x2, y = n1 >= n2 ? y1 : y2, n = n1 >= n2 ? n1 : n2;
    if (n > R(0)) { R inv = rsqrt_(n); ca = x * inv; cb = y * inv; } else { ca = R(1); cb = R(0); }
  }
  R vhi[3] = {ca * u[0] + cb * w[0], ca * u[1] + cb * w[1], ca * u[2] + cb * w[2]};
  R vlo[3] = {-cb * u[0] + ca * w[0], -cb * u[1] + ca * w[1], -cb * u[2] + ca * w[2]};
  if (iso_is_max) {
    lam[0] = l_lo; lam[1] = l_hi; lam[2] = l_iso;
    for (int j = 0; j < 3; ++j) { V[0][j] = vlo[j]; V[1][j] = vhi[j]; V[2][j] = iso[j]; }
  } else {
    lam[0] = l_iso; lam[1] = l_lo; lam[2] = l_hi;
    for (int j = 0; j < 3; ++j) { V[0][j] = iso[j]; V[1][j] = vlo[j]; V[2][j] = vhi[j]; }
  }
  // round-off can leave neighbours out of order by an ulp; restore ascending order
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < 2; ++k) {
      if (lam[k] > lam[k + 1]) {
        R t = lam[k]; lam[k] = lam[k + 1]; lam[k + 1] = t;
        for (int j = 0; j < 3; ++j) { R s = V[k][j]; V[k][j] = V[k + 1][j]; V[k + 1][j] = s; }
      }
    }
  }
  lam[0] *= m; lam[1] *= m; lam[2] *= m;
}

}  // namespace dc
