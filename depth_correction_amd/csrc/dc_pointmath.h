// Per-point / per-neighbourhood arithmetic of the map-consistency path, as inline functions that the
// HIP kernels call (and that csrc/dc_hostcheck.cpp compiles for the host so CPU tests can pin them).
//
// Reference semantics restated here (paths relative to the reference's src/depth_correction/):
//   model bias / corrected depth     model.py:243-261 (ScaledPolynomial), :181-199 (Polynomial)
//   rigid transform                  depth_cloud.py:135-152
//   points                           depth_cloud.py:122-124
//   weighted mean                    depth_cloud.py:291-295
//   neighbour weights                depth_cloud.py:356-364
//   weighted covariance              utils.py:109-149  (Bessel correction, clamp(W-1, 1e-6))
//   normals, incidence angle         depth_cloud.py:401-424
//   min-eigenvalue / trace loss      loss.py:250-289, 330-363
//   backward                         closed form of the reference's autograd graph
#pragma once
#include "dc_common.h"
#include "dc_eig3.h"

namespace dc {

// ---- K1: depth correction model -------------------------------------------------------------
struct ModelParams {
  int kind;                       // DC_MODEL_*
  int n_terms;                    // P
  double w[DC_MAX_MODEL_TERMS];
  double e[DC_MAX_MODEL_TERMS];
};

// gamma^e with the cheap path for the small integral exponents the reference's models use ([2,4]).
DC_HD double pow_term(double g, double e) {
  int ei = (int)e;
  if ((double)ei == e && ei >= 0 && ei <= 8) {
    double r = 1.0, b = g;
    for (int bit = 0; bit < 4; ++bit) {
      if (ei & (1 << bit)) r *= b;
      b *= b;
    }
    return r;
  }
  return pow(g, e);
}

DC_HD double model_bias(const ModelParams& mp, double inc) {
  double b = 0.0;
  // fixed trip count + predicate: keeps w[] / e[] in registers (a dynamic bound would index them through scratch)
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k)
    if (k < mp.n_terms) b += pow_term(inc, mp.e[k]) * mp.w[k];
  return b;
}

// d' = f(d, inc); points outside the local mask keep their depth (model.py:76-78, 256-260).
// Kinds beyond the polynomials (block-uniform switch): Linear model.py:113-146, InvCos :289-313, ScaledInvCos :316-349.
DC_HD double model_depth(const ModelParams& mp, double depth, double inc, bool in_mask) {
  if (mp.kind == DC_MODEL_NONE || !in_mask) return depth;
  if (mp.kind == DC_MODEL_LINEAR) return mp.w[0] * depth + mp.w[1] * inc + mp.w[2];
  if (mp.kind == DC_MODEL_INVCOS) return depth - mp.w[0] / cos(inc);
  if (mp.kind == DC_MODEL_SCALED_INVCOS) return depth * (1.0 - mp.w[0] / fabs(cos(inc)));
  double b = model_bias(mp, inc);
  return mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? depth * (1.0 - b) : depth - b;
}

// dd'/dw_k of the non-polynomial kinds
DC_HD double model_dw_other(const ModelParams& mp, int k, double depth, double inc) {
  if (mp.kind == DC_MODEL_LINEAR) return k == 0 ? depth : (k == 1 ? inc : 1.0);
  if (mp.kind == DC_MODEL_INVCOS) return -1.0 / cos(inc);
  return -depth / fabs(cos(inc));                          // DC_MODEL_SCALED_INVCOS
}

// ---- K2/K3: rigid transform + point --------------------------------------------------------
// T = row-major 3x4 [R | t].
DC_HD void rot3(const double* T, const double* v, double* o) {
  o[0] = T[0] * v[0] + T[1] * v[1] + T[2] * v[2];
  o[1] = T[4] * v[0] + T[5] * v[1] + T[6] * v[2];
  o[2] = T[8] * v[0] + T[9] * v[1] + T[10] * v[2];
}

// ---- K7-K9: one-pass weighted mean / covariance about an anchor -----------------------------
// The anchor (the centre point itself) removes the large common offset exactly, so the one-pass
// second-moment form is as accurate as the reference's two-pass form.
struct CovAcc {
  double W;        // sum of validity weights
  double Wm;       // sum of mean weights (== W unless explicit weights are given)
  double s[3];     // sum w  * d
  double sm[3];    // sum wm * d
  double S[6];     // sum w * d d^T : xx xy xz yy yz zz
};

DC_HD void cov_init(CovAcc& a) {
  a.W = a.Wm = 0.0;
  for (int i = 0; i < 3; ++i) a.s[i] = a.sm[i] = 0.0;
  for (int i = 0; i < 6; ++i) a.S[i] = 0.0;
}

// validity weights only (the hot path): mean weights == validity, filled in by cov_same_weights()
// (explicit fma: the same rounding wherever this is inlined -- the gather and the LDS-staged kernels must agree bit for bit)
DC_HD void cov_add_d(CovAcc& a, double dx, double dy, double dz) {
  a.s[0] += dx; a.s[1] += dy; a.s[2] += dz;
  a.S[0] = fma(dx, dx, a.S[0]); a.S[1] = fma(dx, dy, a.S[1]); a.S[2] = fma(dx, dz, a.S[2]);
  a.S[3] = fma(dy, dy, a.S[3]); a.S[4] = fma(dy, dz, a.S[4]); a.S[5] = fma(dz, dz, a.S[5]);
}
DC_HD void cov_add1(CovAcc& a, double dx, double dy, double dz) {
  a.W += 1.0;
  cov_add_d(a, dx, dy, dz);
}
DC_HD void cov_same_weights(CovAcc& a) {
  a.Wm = a.W;
  a.sm[0] = a.s[0]; a.sm[1] = a.s[1]; a.sm[2] = a.s[2];
}

DC_HD void cov_add(CovAcc& a, double dx, double dy, double dz, double wm) {
  a.W += 1.0;
  a.Wm += wm;
  a.s[0] += dx; a.s[1] += dy; a.s[2] += dz;
  a.sm[0] += wm * dx; a.sm[1] += wm * dy; a.sm[2] += wm * dz;
  a.S[0] += dx * dx; a.S[1] += dx * dy; a.S[2] += dx * dz;
  a.S[3] += dy * dy; a.S[4] += dy * dz; a.S[5] += dz * dz;
}

// mean_off: weighted mean minus anchor (depth_cloud.py:291-295 uses the cloud's current weights);
// C: covariance with the weights of update_weights (validity x optional Gaussian of the distance
// between the centre and the mean, depth_cloud.py:356-364) and utils.covs' normalisation.
// cmean_off: the mean covs() itself subtracts (validity-weighted), returned for the backward.
// unit2: square of the length unit of the accumulated differences (q32 grid steps; 1 for metres), folded into the
// normalisation factor so that C comes out in m^2 without six extra multiplications.
DC_HD void cov_finish(const CovAcc& a, double scale, double* mean_off, double* cmean_off, double* C, double* D_out,
                      double* omega_out, double unit2 = 1.0) {
  // 0/0 -> NaN exactly like the reference when a neighbourhood has no valid member.
  const double invWm = recip_(a.Wm);        // Wm = 0 -> NaN either way (0 * inf), like the reference's 0/0
  for (int i = 0; i < 3; ++i) mean_off[i] = a.Wm > 0.0 ? a.sm[i] * invWm : a.sm[i] / a.Wm;
  double omega = 1.0;
  if (scale > 0.0) {
    // dist = |x_i - mean_i| and x_i is the anchor
    double d2 = mean_off[0] * mean_off[0] + mean_off[1] * mean_off[1] + mean_off[2] * mean_off[2];
    omega = exp(-d2 / (scale * scale));
  }
  const double Wc = omega * a.W;
  double D = Wc - 1.0;
  D = D < 1e-6 ? 1e-6 : D;
  const double invW = (a.W == a.Wm) ? invWm : recip_(a.W);
  const double c0 = a.s[0] * invW, c1 = a.s[1] * invW, c2 = a.s[2] * invW;
  cmean_off[0] = c0; cmean_off[1] = c1; cmean_off[2] = c2;
  const double f = omega * recip_(D) * unit2;       // D >= 1e-6
  C[0] = (a.S[0] - a.s[0] * c0) * f;
  C[1] = (a.S[1] - a.s[0] * c1) * f;
  C[2] = (a.S[2] - a.s[0] * c2) * f;
  C[3] = (a.S[3] - a.s[1] * c1) * f;
  C[4] = (a.S[4] - a.s[1] * c2) * f;
  C[5] = (a.S[5] - a.s[2] * c2) * f;
  *D_out = D;
  *omega_out = omega;
}

// ---- K11/K12: oriented normal + incidence angle ----------------------------------------------
DC_HD void normal_and_incidence(const double* dir, const double* v0, double* normal, double* inc) {
  double c = dir[0] * v0[0] + dir[1] * v0[1] + dir[2] * v0[2];
  double sgn = (c > 0.0) ? 1.0 : ((c < 0.0) ? -1.0 : 0.0);     // torch.sign: sign(0) = 0
  normal[0] = -sgn * v0[0]; normal[1] = -sgn * v0[1]; normal[2] = -sgn * v0[2];
  double a = fabs(c);
  if (c != c) { *inc = c; return; }
  // |dir . n| can exceed 1 by an ulp for unit vectors; the reference's arccos would return NaN there.
  *inc = acos(a > 1.0 ? 1.0 : a);
}

// ---- K14/K15 + backward coefficients ------------------------------------------------------------
struct LossParams {
  int kind;            // DC_LOSS_*
  int normalization;   // min-eigenvalue / total variance (loss.py:253-254)
  int sqrt_;           // loss.py:286-287
  int raw_pointwise;   // the optional per-point output holds the loss BEFORE relu / sqrt (what loss.py:256-277 gates on)
  int skip;            // 0: every masked point counts; 1: NaN losses are dropped (skip_nans); 2: non-finite ones (only_finite) -- loss.py:125-137
};
// the pointwise loss `l` of a masked point is dropped from the reduction: no term in the sum, the count or the gradients
DC_HD bool loss_dropped(const LossParams& lp, double l) {
  return lp.skip != 0 && (l != l || (lp.skip == 2 && !(fabs(l) < (double)INFINITY)));
}
DC_HD LossParams make_loss_params(int loss_kind, int normalization, int sqrt_) {
  return LossParams{loss_kind & 0xFF, normalization, sqrt_, (loss_kind & DC_LOSS_RAW_POINTWISE) != 0,
                    (loss_kind & DC_LOSS_ONLY_FINITE) ? 2 : ((loss_kind & DC_LOSS_SKIP_NANS) ? 1 : 0)};
}

// Pointwise loss l_i (after offset, relu, sqrt) and the coefficients of
//   dL/dx_j += c1 * (v0 . d) v0 - c2 * d,  d = x_j - cmean_i,
// for a unit upstream weight (mask and 1/M are applied by the caller through `a`).
DC_HD double loss_and_coeffs(const LossParams& lp, double lam0, double tr, double D, double offset, bool in_mask,
                             double* c1, double* c2, double* raw_out = nullptr) {
  double raw, g_vv = 0.0, g_eye = 0.0;    // G = g_vv * v0 v0^T + g_eye * I
  if (lp.kind == DC_LOSS_MIN_EIGVAL) {
    if (lp.normalization) {
      const double tc = tr < 1e-6 ? 1e-6 : tr;
      const double inv = recip_(tc);          // tc >= 1e-6
      raw = lam0 * inv;
      g_vv = inv;
      g_eye = (tr > 1e-6) ? -raw * inv : 0.0;
    } else {
      raw = lam0;
      g_vv = 1.0;
    }
  } else {
    raw = tr;
    g_eye = 1.0;
  }
  double l = raw - offset;
  if (raw_out) *raw_out = l;
  double a = (in_mask && l > 0.0) ? 1.0 : 0.0;     // relu'; NaN compares false
  l = l > 0.0 ? l : (l != l ? l : 0.0);           // relu keeps NaN
  if (lp.sqrt_) {
    double s = sqrt(l);
    a = (l > 0.0) ? a * 0.5 / s : 0.0;
    l = s;
  }
  const double f = (D == 9.0) ? a * (2.0 / 9.0) : 2.0 * a * recip_(D);     // K = 10 valid neighbours: a constant
  *c1 = f * g_vv;
  *c2 = -f * g_eye;
  return l;
}

}  // namespace dc
