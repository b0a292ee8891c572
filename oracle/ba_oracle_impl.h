/*
 * oracle/ba_oracle_impl.h -- CPU restatement of the reference's dense bundle adjustment.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and there only as the
 * checker / timed CPU baseline.  PARITY UNPINNED: the reference ships no fixtures or tests for
 * this path and neither its CUDA extension nor its lietorch-based Python twin can run in this
 * pipeline, so this restatement is pinned only by independent derivations
 * (tests/test_oracle_*.py), not by reference outputs.
 *
 * The file is included twice by ba_oracle.c, once with REAL=double (suffix _f64, "truth": every
 * quantity in double) and once with REAL=float (suffix _f32: fp32 wherever the reference's device
 * code is fp32, fp64 for the host-side solve, like the reference).
 *
 * Every function cites the lines of /root/reference/src/droid_kernels.cu it follows ("dk:").
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef REAL FN(real);
#define real FN(real)

/* ------------------------------------------------------------------ SE3 helpers, dk:58-175 */

/* dk:58-68 actSO3: rotate X by unit quaternion q=(x,y,z,w). */
static void FN(actSO3)(const real *q, const real *X, real *Y) {
  real uv[3];
  uv[0] = (real)(2.0 * (q[1] * X[2] - q[2] * X[1]));
  uv[1] = (real)(2.0 * (q[2] * X[0] - q[0] * X[2]));
  uv[2] = (real)(2.0 * (q[0] * X[1] - q[1] * X[0]));
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

/* dk:70-77 actSE3 on a homogeneous point (X,Y,Z,disp). */
static void FN(actSE3)(const real *t, const real *q, const real *X, real *Y) {
  FN(actSO3)(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0];
  Y[1] += X[3] * t[1];
  Y[2] += X[3] * t[2];
}

/* dk:79-94 adjSE3: Y = Adj(T)^T X for a 6-vector X (tau, phi). */
static void FN(adjSE3)(const real *t, const real *q, const real *X, real *Y) {
  real qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  FN(actSO3)(qinv, &X[0], &Y[0]);
  FN(actSO3)(qinv, &X[3], &Y[3]);
  real u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  FN(actSO3)(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}

/* dk:96-107 relSE3: Tij = Tj * Ti^-1. */
static void FN(relSE3)(const real *ti, const real *qi, const real *tj, const real *qj, real *tij,
                       real *qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  FN(actSO3)(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}

/* dk:110-132 expSO3 with the small-angle branch at theta^2 < 1e-8. */
static void FN(expSO3)(const real *phi, real *q) {
  real theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  real theta_p4 = theta_sq * theta_sq;
  real theta = (real)sqrt((double)theta_sq);
  real imag, re;
  if (theta_sq < 1e-8) {
    imag = (real)(0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4);
    re = (real)(1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4);
  } else {
    imag = (real)(SIN((real)(0.5 * theta)) / theta);
    re = (real)COS((real)(0.5 * theta));
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = re;
}

/* dk:134-145 */
static void FN(crossInplace)(const real *a, real *b) {
  real x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0];
  b[1] = x[1];
  b[2] = x[2];
}

/* dk:147-175 expSE3 with the V-matrix branch at theta > 1e-4. */
static void FN(expSE3)(const real *xi, real *t, real *q) {
  FN(expSO3)(xi + 3, q);
  real tau[3] = {xi[0], xi[1], xi[2]};
  real phi[3] = {xi[3], xi[4], xi[5]};
  real theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  real theta = (real)sqrt((double)theta_sq);
  t[0] = tau[0];
  t[1] = tau[1];
  t[2] = tau[2];
  if (theta > 1e-4) {
    real a = (1 - (real)COS(theta)) / theta_sq;
    FN(crossInplace)(phi, tau);
    t[0] += a * tau[0];
    t[1] += a * tau[1];
    t[2] += a * tau[2];
    real b = (theta - (real)SIN(theta)) / (theta * theta_sq);
    FN(crossInplace)(phi, tau);
    t[0] += b * tau[0];
    t[1] += b * tau[1];
    t[2] += b * tau[2];
  }
}

/* dk:877-895 retrSE3: T1 = exp(xi) * T (left multiplication, no renormalisation). */
static void FN(retrSE3)(const real *xi, const real *t, const real *q, real *t1, real *q1) {
  real dt[3] = {0, 0, 0};
  real dq[4] = {0, 0, 0, 1};
  FN(expSE3)(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  FN(actSO3)(dq, t, t1);
  t1[0] += dt[0];
  t1[1] += dt[1];
  t1[2] += dt[2];
}

/* ------------------------------------------------------------------ linearisation, dk:176-424
 * One edge e=(ix -> jx).  Outputs (all for this edge):
 *   Hs4[4][36]  Hii,Hij,Hji,Hjj row-major 6x6        (dk:400-423)
 *   vs2[2][6]   vi, vj                                (dk:383-398)
 *   Eii,Eij [6][HW], Cii[HW], bz[HW]                  (dk:320-321,340-341,353-354,373-374)
 * Block sums are accumulated in `real` (the reference: per-thread fp32 partials + tree reduce;
 * the order of an fp32 sum is not reproducible there either). */
static void FN(linearize_edge)(const real *target, const real *weight, /* [2][HW] each */
                               const real *poses, const real *disps, const real *intr, int ix,
                               int jx, int ht, int wd, real *Hs4, real *vs2, real *Eii, real *Eij,
                               real *Cii, real *bz) {
  const int HW = ht * wd;
  const real fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  real tij[3], qij[4];
  if (ix == jx) { /* dk:219-229 stereo pair: fixed baseline */
    tij[0] = (real)-0.1;
    tij[1] = 0;
    tij[2] = 0;
    qij[0] = 0;
    qij[1] = 0;
    qij[2] = 0;
    qij[3] = 1;
  } else {
    FN(relSE3)(&poses[7 * ix], &poses[7 * ix + 3], &poses[7 * jx], &poses[7 * jx + 3], tij, qij);
  }
  real hij[78];
  real vi[6], vj[6];
  for (int l = 0; l < 78; l++) hij[l] = 0;
  for (int n = 0; n < 6; n++) vi[n] = vj[n] = 0;
  const real *disp_i = &disps[(size_t)ix * HW];

  for (int k = 0; k < HW; k++) {
    const int i = k / wd, j = k % wd;
    const real u = (real)j, v = (real)i;
    real Xi[4], Xj[4], Jx[12], Jz;
    real *Ji = &Jx[0], *Jj = &Jx[6];
    Xi[0] = (u - cx) / fx;
    Xi[1] = (v - cy) / fy;
    Xi[2] = 1;
    Xi[3] = disp_i[k];
    FN(actSE3)(tij, qij, Xi, Xj);
    const real x = Xj[0], y = Xj[1], h = Xj[3];
    const real d = (Xj[2] < MIN_DEPTH) ? (real)0.0 : (real)(1.0 / Xj[2]);
    const real d2 = d * d;
    real wu = (Xj[2] < MIN_DEPTH) ? (real)0.0 : (real)(.001 * weight[k]);
    real wv = (Xj[2] < MIN_DEPTH) ? (real)0.0 : (real)(.001 * weight[HW + k]);
    const real ru = target[k] - (fx * d * x + cx);
    const real rv = target[HW + k] - (fy * d * y + cy);

    /* x - coordinate, dk:312-341 */
    Jj[0] = fx * (h * d);
    Jj[1] = fx * 0;
    Jj[2] = fx * (-x * h * d2);
    Jj[3] = fx * (-x * y * d2);
    Jj[4] = fx * (1 + x * x * d2);
    Jj[5] = fx * (-y * d);
    Jz = fx * (tij[0] * d - tij[2] * (x * d2));
    Cii[k] = wu * Jz * Jz;
    bz[k] = wu * ru * Jz;
    if (ix == jx) wu = 0;
    FN(adjSE3)(tij, qij, Jj, Ji);
    for (int n = 0; n < 6; n++) Ji[n] *= -1;
    int l = 0;
    for (int n = 0; n < 12; n++)
      for (int m = 0; m <= n; m++) hij[l++] += wu * Jx[n] * Jx[m];
    for (int n = 0; n < 6; n++) {
      vi[n] += wu * ru * Ji[n];
      vj[n] += wu * ru * Jj[n];
      Eii[n * HW + k] = wu * Jz * Ji[n];
      Eij[n * HW + k] = wu * Jz * Jj[n];
    }

    /* y - coordinate, dk:345-375 */
    Jj[0] = fy * 0;
    Jj[1] = fy * (h * d);
    Jj[2] = fy * (-y * h * d2);
    Jj[3] = fy * (-1 - y * y * d2);
    Jj[4] = fy * (x * y * d2);
    Jj[5] = fy * (x * d);
    Jz = fy * (tij[1] * d - tij[2] * (y * d2));
    Cii[k] += wv * Jz * Jz;
    bz[k] += wv * rv * Jz;
    if (ix == jx) wv = 0;
    FN(adjSE3)(tij, qij, Jj, Ji);
    for (int n = 0; n < 6; n++) Ji[n] *= -1;
    l = 0;
    for (int n = 0; n < 12; n++)
      for (int m = 0; m <= n; m++) hij[l++] += wv * Jx[n] * Jx[m];
    for (int n = 0; n < 6; n++) {
      vi[n] += wv * rv * Ji[n];
      vj[n] += wv * rv * Jj[n];
      Eii[n * HW + k] += wv * Jz * Ji[n];
      Eij[n * HW + k] += wv * Jz * Jj[n];
    }
  }

  for (int n = 0; n < 6; n++) {
    vs2[n] = vi[n];
    vs2[6 + n] = vj[n];
  }
  real *Hii = Hs4, *Hij = Hs4 + 36, *Hji = Hs4 + 72, *Hjj = Hs4 + 108;
  int l = 0;
  for (int n = 0; n < 12; n++) {
    for (int m = 0; m <= n; m++) {
      const real s = hij[l++];
      if (n < 6 && m < 6) { /* dk:407-410 */
        Hii[n * 6 + m] = s;
        Hii[m * 6 + n] = s;
      } else if (n >= 6 && m < 6) { /* dk:411-414 */
        Hij[m * 6 + (n - 6)] = s;
        Hji[(n - 6) * 6 + m] = s;
      } else { /* dk:415-418 */
        Hjj[(n - 6) * 6 + (m - 6)] = s;
        Hjj[(m - 6) * 6 + (n - 6)] = s;
      }
    }
  }
}

/* ------------------------------------------------------------------ driver, dk:1314-1434
 *
 * State (poses [nbuf][7], disps [nbuf][HW]) is updated in place, as in the reference.
 * The host-side sparse solve (dk:1117-1219, Eigen SimplicialLLT<double>) is restated as a dense
 * fp64 Cholesky: the same factorisation of the same matrix.
 *
 * Sharding hooks (not in the reference; used by the world_size-2 tests): own0/own1 restrict the
 * window frames that create depth slots to [max(t0,own0), min(t1,own1)); with own0=0, own1=INT_MAX
 * this is exactly the reference.  `mode`: 0 = full iteration loop; 1 = build only (one
 * linearisation, writes the damping-free dense system to sys_H/sys_b and keeps the depth-side
 * state in `keep`); see ba_oracle.c for the phase API built on top of this.
 *
 * Returns 0, or a negative code for contract violations the reference would hit as UB / a torch
 * shape error (eta rows != M, index out of range).
 */
typedef struct {
  int P, M, E, HW;
  int64_t *kx;      /* [M] sorted unique depth frames */
  int *kk;          /* [Pw+E] slot of each expanded entry */
  int *exp_pose;    /* [Pw+E] pose index jj_exp - t0 of each expanded entry */
  int *exp_src;     /* [Pw+E] frame ii_exp */
  int n_exp, Pw;    /* Pw = number of (owned) window frames in the expanded list */
  real *Q, *w;      /* [M][HW] */
  real *Erows;      /* [(Pw+E)][6][HW]: Ei for window frames, then Eij for edges (dk:1402-1403) */
} FN(depth_state);

static void FN(free_depth_state)(FN(depth_state) * s) {
  free(s->kx);
  free(s->kk);
  free(s->exp_pose);
  free(s->exp_src);
  free(s->Q);
  free(s->w);
  free(s->Erows);
  memset(s, 0, sizeof(*s));
}

/* One linearisation + assembly: everything of one `for itr` pass up to (not including) the solve.
 * Hd [n*n] (n = 6P) and bd [n] receive A - S and its right-hand side in fp64 WITHOUT damping. */
static int FN(build_system)(const real *poses, const real *disps, const real *intr,
                            const real *disps_sens, const real *targets, const real *weights,
                            const real *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                            int E, int nbuf, int ht, int wd, int t0, int t1, int own0, int own1,
                            int motion_only, double *Hd, double *bd, FN(depth_state) * st,
                            real *dbg_Hs, real *dbg_vs) {
  const int HW = ht * wd;
  const int P = t1 - t0;
  const int n = 6 * P;
  if (P <= 0) return -1;
  for (int e = 0; e < E; e++)
    if (ii[e] < 0 || ii[e] >= nbuf || jj[e] < 0 || jj[e] >= nbuf) return -2;
  if (t0 < 0 || t1 > nbuf) return -2;

  memset(Hd, 0, sizeof(double) * (size_t)n * n);
  memset(bd, 0, sizeof(double) * n);

  /* dk:1350-1355 work buffers */
  real *Hs = (real *)calloc((size_t)E * 144 + 1, sizeof(real));
  real *vs = (real *)calloc((size_t)E * 12 + 1, sizeof(real));
  real *Eii = (real *)calloc((size_t)E * 6 * HW + 1, sizeof(real));
  real *Eij = (real *)calloc((size_t)E * 6 * HW + 1, sizeof(real));
  real *Cii = (real *)calloc((size_t)E * HW + 1, sizeof(real));
  real *wi = (real *)calloc((size_t)E * HW + 1, sizeof(real));

  /* dk:1359-1372 */
#pragma omp parallel for schedule(dynamic, 4)
  for (int e = 0; e < E; e++) {
    FN(linearize_edge)(&targets[(size_t)e * 2 * HW], &weights[(size_t)e * 2 * HW], poses, disps,
                       intr, (int)ii[e], (int)jj[e], ht, wd, &Hs[(size_t)e * 144],
                       &vs[(size_t)e * 12], &Eii[(size_t)e * 6 * HW], &Eij[(size_t)e * 6 * HW],
                       &Cii[(size_t)e * HW], &wi[(size_t)e * HW]);
  }
  if (dbg_Hs) memcpy(dbg_Hs, Hs, sizeof(real) * (size_t)E * 144);
  if (dbg_vs) memcpy(dbg_vs, vs, sizeof(real) * (size_t)E * 12);

  /* dk:1376-1383 + SparseBlock::update_lhs/update_rhs dk:1131-1173: fp32 blocks widened to fp64,
   * scatter-added; block rows/cols < 0 are dropped (dk:1146, dk:1167).  Indices >= P are undefined
   * behaviour in the reference (no upper-bound check); they are dropped here. */
  for (int e = 0; e < E; e++) {
    const int pi = (int)ii[e] - t0, pj = (int)jj[e] - t0;
    const int rows[4] = {pi, pi, pj, pj};
    const int cols[4] = {pi, pj, pi, pj};
    for (int b = 0; b < 4; b++) {
      const int r = rows[b], c = cols[b];
      if (r < 0 || c < 0 || r >= P || c >= P) continue;
      for (int k = 0; k < 6; k++)
        for (int l = 0; l < 6; l++)
          Hd[(size_t)(6 * r + k) * n + 6 * c + l] += (double)Hs[(size_t)e * 144 + b * 36 + k * 6 + l];
    }
    if (pi >= 0 && pi < P)
      for (int k = 0; k < 6; k++) bd[6 * pi + k] += (double)vs[(size_t)e * 12 + k];
    if (pj >= 0 && pj < P)
      for (int k = 0; k < 6; k++) bd[6 * pj + k] += (double)vs[(size_t)e * 12 + 6 + k];
  }

  if (motion_only) {
    free(Hs); free(vs); free(Eii); free(Eij); free(Cii); free(wi);
    st->P = P; st->M = 0; st->E = E; st->HW = HW;
    return 0;
  }

  /* dk:1336-1344: ts, ii_exp = cat(ts, ii), jj_exp = cat(ts, jj), (kx, kk_exp) = unique(ii_exp).
   * With the sharding hook, ts only spans the owned part of the window. */
  const int w0 = t0 > own0 ? t0 : own0;
  const int w1 = t1 < own1 ? t1 : own1;
  const int Pw = w1 > w0 ? w1 - w0 : 0;
  const int n_exp = Pw + E;
  int64_t *ii_exp = (int64_t *)malloc(sizeof(int64_t) * (n_exp + 1));
  int64_t *jj_exp = (int64_t *)malloc(sizeof(int64_t) * (n_exp + 1));
  for (int p = 0; p < Pw; p++) ii_exp[p] = jj_exp[p] = w0 + p;
  for (int e = 0; e < E; e++) {
    ii_exp[Pw + e] = ii[e];
    jj_exp[Pw + e] = jj[e];
  }
  char *present = (char *)calloc(nbuf, 1);
  for (int a = 0; a < n_exp; a++) present[ii_exp[a]] = 1;
  int M = 0;
  for (int f = 0; f < nbuf; f++) M += present[f];
  int64_t *kx = (int64_t *)malloc(sizeof(int64_t) * (M + 1));
  int *slot_of = (int *)malloc(sizeof(int) * nbuf);
  M = 0;
  for (int f = 0; f < nbuf; f++) {
    slot_of[f] = -1;
    if (present[f]) {
      slot_of[f] = M;
      kx[M++] = f;
    }
  }
  free(present);
  if (eta_rows != M) { /* (1-m)*eta.view(-1,HW) would not broadcast, dk:1398 */
    free(ii_exp); free(jj_exp); free(kx); free(slot_of);
    free(Hs); free(vs); free(Eii); free(Eij); free(Cii); free(wi);
    return -3;
  }
  int *kk = (int *)malloc(sizeof(int) * (n_exp + 1));
  for (int a = 0; a < n_exp; a++) kk[a] = slot_of[ii_exp[a]];

  /* dk:1396-1400 depth system: C, w, Q over the M depth slots (accum_cuda dk:948-998 = sum of the
   * rows whose key equals kx[m]). */
  real *C = (real *)calloc((size_t)M * HW + 1, sizeof(real));
  real *w = (real *)calloc((size_t)M * HW + 1, sizeof(real));
  for (int e = 0; e < E; e++) {
    const int m = slot_of[ii[e]];
    for (int k = 0; k < HW; k++) {
      C[(size_t)m * HW + k] += Cii[(size_t)e * HW + k];
      w[(size_t)m * HW + k] += wi[(size_t)e * HW + k];
    }
  }
  const real alpha = (real)0.05;
  real *Q = (real *)malloc(sizeof(real) * ((size_t)M * HW + 1));
  for (int m = 0; m < M; m++) {
    const size_t f = (size_t)kx[m];
    for (int k = 0; k < HW; k++) {
      const real ms = disps_sens[f * HW + k] > 0 ? (real)1 : (real)0;
      const size_t o = (size_t)m * HW + k;
      C[o] = C[o] + ms * alpha + (1 - ms) * eta[o];
      w[o] = w[o] - ms * alpha * (disps[f * HW + k] - disps_sens[f * HW + k]);
      Q[o] = (real)(1.0 / C[o]);
    }
  }

  /* dk:1402-1403: Ei = accum(Eii, ii, ts); E = cat(Ei, Eij). */
  real *Er = (real *)calloc((size_t)n_exp * 6 * HW + 1, sizeof(real));
  for (int e = 0; e < E; e++) {
    const int p = (int)ii[e] - w0;
    if (p >= 0 && p < Pw) {
      real *dst = &Er[(size_t)p * 6 * HW];
      const real *src = &Eii[(size_t)e * 6 * HW];
      for (int k = 0; k < 6 * HW; k++) dst[k] += src[k];
    }
    memcpy(&Er[(size_t)(Pw + e) * 6 * HW], &Eij[(size_t)e * 6 * HW], sizeof(real) * 6 * HW);
  }

  /* schur_block dk:1222-1311.  graph[t] lists (depth slot, row) of every expanded entry whose
   * pose jj_exp is in the window (dk:1244-1253; the reference's inclusive `j <= t1` would index
   * graph[P] out of range and is never reached by callers, so the bound is exclusive here).  The
   * quadruple loop dk:1257-1272 pairs entries of poses i and j that share a depth slot; each pair
   * is one EEt6x6 block (dk:1001-1056) accumulated into S(i,j).  Enumerated per depth slot here,
   * which visits exactly the same (entry a, entry b) pairs. */
  {
    int *cnt = (int *)calloc(M + 1, sizeof(int));
    for (int a = 0; a < n_exp; a++) {
      const int p = (int)jj_exp[a] - t0;
      if (p >= 0 && p < P) cnt[kk[a] + 1]++;
    }
    for (int m = 0; m < M; m++) cnt[m + 1] += cnt[m];
    int *lst = (int *)malloc(sizeof(int) * (cnt[M] + 1));
    int *fill = (int *)calloc(M + 1, sizeof(int));
    for (int a = 0; a < n_exp; a++) {
      const int p = (int)jj_exp[a] - t0;
      if (p >= 0 && p < P) lst[cnt[kk[a]] + fill[kk[a]]++] = a;
    }
    free(fill);
#pragma omp parallel
    {
      real *qe = (real *)malloc(sizeof(real) * 6 * HW);
#pragma omp for schedule(dynamic, 1)
      for (int m = 0; m < M; m++) {
        const real *Qm = &Q[(size_t)m * HW];
        const real *wm = &w[(size_t)m * HW];
        for (int xa = cnt[m]; xa < cnt[m + 1]; xa++) {
          const int a = lst[xa];
          const int pa = (int)jj_exp[a] - t0;
          const real *Ea = &Er[(size_t)a * 6 * HW];
          for (int c = 0; c < 6; c++)
            for (int k = 0; k < HW; k++) qe[c * HW + k] = Ea[c * HW + k] * Qm[k]; /* dk:1030 */
          for (int xb = cnt[m]; xb < cnt[m + 1]; xb++) {
            const int b = lst[xb];
            const int pb = (int)jj_exp[b] - t0;
            const real *Eb = &Er[(size_t)b * 6 * HW];
            real dS[36];
            for (int c = 0; c < 6; c++)
              for (int d = 0; d < 6; d++) {
                real s = 0;
                const real *x = &qe[c * HW], *y = &Eb[d * HW];
                for (int k = 0; k < HW; k++) s += x[k] * y[k]; /* dk:1035-1039 */
                dS[c * 6 + d] = s;
              }
            for (int c = 0; c < 6; c++)
              for (int d = 0; d < 6; d++) {
                double *dst = &Hd[(size_t)(6 * pa + c) * n + 6 * pb + d];
                const double val = (double)dS[c * 6 + d];
#pragma omp atomic
                *dst -= val; /* (A - S), dk:1175-1177, 1406 */
              }
          }
          /* Ev6x1 dk:1059-1093 + update_rhs(v, jj_exp - t0) dk:1308 */
          for (int c = 0; c < 6; c++) {
            real s = 0;
            for (int k = 0; k < HW; k++) s += (Qm[k] * wm[k]) * Ea[c * HW + k];
            double *dst = &bd[6 * pa + c];
            const double val = (double)s;
#pragma omp atomic
            *dst -= val;
          }
        }
      }
      free(qe);
    }
    free(cnt);
    free(lst);
  }

  st->P = P; st->M = M; st->E = E; st->HW = HW;
  st->kx = kx; st->kk = kk; st->n_exp = n_exp; st->Pw = Pw;
  st->exp_pose = (int *)malloc(sizeof(int) * (n_exp + 1));
  st->exp_src = (int *)malloc(sizeof(int) * (n_exp + 1));
  for (int a = 0; a < n_exp; a++) {
    st->exp_pose[a] = (int)jj_exp[a] - t0;
    st->exp_src[a] = (int)ii_exp[a];
  }
  st->Q = Q; st->w = w; st->Erows = Er;
  free(C); free(ii_exp); free(jj_exp); free(slot_of);
  free(Hs); free(vs); free(Eii); free(Eij); free(Cii); free(wi);
  return 0;
}

/* SparseBlock::solve dk:1192-1213: L = A; L.diag += ep + lm*L.diag; LL^T; x = L^-T L^-1 b in fp64;
 * factorisation failure => dx = 0 (dk:1207-1210).  dx is rounded to `real` like the reference's
 * .to(kFloat32).  Hd is destroyed. */
static int FN(solve_system)(double *Hd, const double *bd, int P, double lm, double ep, real *dx) {
  const int n = 6 * P;
  for (int i = 0; i < n; i++) Hd[(size_t)i * n + i] += ep + lm * Hd[(size_t)i * n + i];
  int ok = 1;
  for (int j = 0; j < n && ok; j++) { /* lower Cholesky, row-major, in place */
    double *Lj = &Hd[(size_t)j * n];
    double d = Lj[j];
    for (int k = 0; k < j; k++) d -= Lj[k] * Lj[k];
    if (!(d > 0.0)) {
      ok = 0;
      break;
    }
    d = sqrt(d);
    Lj[j] = d;
#pragma omp parallel for schedule(static) if (n - j > 256)
    for (int i = j + 1; i < n; i++) {
      double *Li = &Hd[(size_t)i * n];
      double s = Li[j];
      for (int k = 0; k < j; k++) s -= Li[k] * Lj[k];
      Li[j] = s / d;
    }
  }
  if (!ok) {
    for (int i = 0; i < n; i++) dx[i] = 0;
    return 1;
  }
  double *y = (double *)malloc(sizeof(double) * n);
  for (int i = 0; i < n; i++) {
    double s = bd[i];
    for (int k = 0; k < i; k++) s -= Hd[(size_t)i * n + k] * y[k];
    y[i] = s / Hd[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= Hd[(size_t)k * n + i] * y[k];
    y[i] = s / Hd[(size_t)i * n + i];
  }
  for (int i = 0; i < n; i++) dx[i] = (real)y[i];
  free(y);
  return 0;
}

/* dk:1408-1417: EvT6x1 (with its `p <= 0 || p >= P` early return, dk:1105-1106, which drops the
 * first window pose from the depth back-substitution), accum over depth slots, dz = Q*(w - sum). */
static void FN(back_substitute)(const FN(depth_state) * st, const real *dx, real *dz) {
  const int HW = st->HW, M = st->M;
  real *acc = (real *)calloc((size_t)M * HW + 1, sizeof(real));
  for (int a = 0; a < st->n_exp; a++) {
    const int p = st->exp_pose[a];
    if (p <= 0 || p >= st->P) continue;
    const real *Ea = &st->Erows[(size_t)a * 6 * HW];
    real *dst = &acc[(size_t)st->kk[a] * HW];
    for (int k = 0; k < HW; k++) {
      real dw = 0;
      for (int c = 0; c < 6; c++) dw += Ea[c * HW + k] * dx[6 * p + c];
      dst[k] += dw;
    }
  }
  for (size_t o = 0; o < (size_t)M * HW; o++) dz[o] = st->Q[o] * (st->w[o] - acc[o]);
  free(acc);
}

/* pose_retr_kernel dk:898-931 */
static void FN(retract_poses)(real *poses, const real *dx, int t0, int t1) {
  for (int k = t0; k < t1; k++) {
    real t1v[3], q1[4];
    FN(retrSE3)(&dx[6 * (k - t0)], &poses[7 * k], &poses[7 * k + 3], t1v, q1);
    for (int c = 0; c < 3; c++) poses[7 * k + c] = t1v[c];
    for (int c = 0; c < 4; c++) poses[7 * k + 3 + c] = q1[c];
  }
}

/* disp_retr_kernel dk:933-946 */
static void FN(retract_disps)(real *disps, const real *dz, const int64_t *kx, int M, int HW) {
  for (int m = 0; m < M; m++)
    for (int k = 0; k < HW; k++) disps[(size_t)kx[m] * HW + k] += dz[(size_t)m * HW + k];
}

/* The exported whole-call entry point: ba_cuda dk:1314-1434. */
int FN(droid_oracle_ba)(real *poses, real *disps, const real *intr, const real *disps_sens,
                        const real *targets, const real *weights, const real *eta, int eta_rows,
                        const int64_t *ii, const int64_t *jj, int E, int nbuf, int ht, int wd,
                        int t0, int t1, int iterations, double lm, double ep, int motion_only,
                        real *dx_out, real *dz_out, int64_t *kx_out, int *M_out,
                        double *dbg_H, double *dbg_b, real *dbg_Hs, real *dbg_vs) {
  const int P = t1 - t0;
  if (P <= 0) return -1;
  const int n = 6 * P;
  const int HW = ht * wd;
  double *Hd = (double *)malloc(sizeof(double) * (size_t)n * n);
  double *bd = (double *)malloc(sizeof(double) * n);
  real *dx = (real *)calloc(n, sizeof(real));
  int rc = 0;
  for (int itr = 0; itr < iterations; itr++) {
    FN(depth_state) st;
    memset(&st, 0, sizeof(st));
    rc = FN(build_system)(poses, disps, intr, disps_sens, targets, weights, eta, eta_rows, ii, jj,
                          E, nbuf, ht, wd, t0, t1, 0, 2147483647, motion_only, Hd, bd, &st,
                          itr == 0 ? dbg_Hs : NULL, itr == 0 ? dbg_vs : NULL);
    if (rc) break;
    if (itr == 0 && dbg_H) memcpy(dbg_H, Hd, sizeof(double) * (size_t)n * n);
    if (itr == 0 && dbg_b) memcpy(dbg_b, bd, sizeof(double) * n);
    FN(solve_system)(Hd, bd, P, lm, ep, dx);
    /* storage formats of the reference: dx, dz, poses and disps are float32 tensors (dk:1202-1212 `dx` is
     * converted to kFloat32, dk:1417 dz = Q * (w - ...) on float tensors, dk:898-946 the retraction kernels
     * write float).  With g_storage_f32 the fp64 restatement rounds exactly these four to float, so that a
     * multi-iteration call sees the state the reference's next iteration would see. */
    if (g_storage_f32)
      for (int k = 0; k < n; k++) dx[k] = (real)(float)dx[k];
    if (!motion_only) {
      real *dz = (real *)malloc(sizeof(real) * ((size_t)st.M * HW + 1));
      FN(back_substitute)(&st, dx, dz);
      if (g_storage_f32)
        for (size_t k = 0; k < (size_t)st.M * HW; k++) dz[k] = (real)(float)dz[k];
      FN(retract_poses)(poses, dx, t0, t1);
      FN(retract_disps)(disps, dz, st.kx, st.M, HW);
      if (g_storage_f32) {
        for (int k = 7 * t0; k < 7 * t1; k++) poses[k] = (real)(float)poses[k];
        for (int q = 0; q < st.M; q++)
          for (int k = 0; k < HW; k++) disps[(size_t)st.kx[q] * HW + k] = (real)(float)disps[(size_t)st.kx[q] * HW + k];
      }
      if (dz_out) memcpy(dz_out, dz, sizeof(real) * (size_t)st.M * HW);
      if (kx_out) memcpy(kx_out, st.kx, sizeof(int64_t) * st.M);
      if (M_out) *M_out = st.M;
      free(dz);
    } else {
      FN(retract_poses)(poses, dx, t0, t1);
      if (g_storage_f32)
        for (int k = 7 * t0; k < 7 * t1; k++) poses[k] = (real)(float)poses[k];
      if (M_out) *M_out = 0;
    }
    FN(free_depth_state)(&st);
  }
  if (dx_out) memcpy(dx_out, dx, sizeof(real) * n);
  free(Hd);
  free(bd);
  free(dx);
  return rc;
}

/* ---- phase API for the sharded (multi-rank) host logic tests: build / finish ------------- */

typedef struct {
  FN(depth_state) st;
  int t0, t1, motion_only;
} FN(phase_handle);

/* Phase 1: this rank's edges + owned frames -> dense (A - S), rhs (no damping), summed across
 * ranks by the caller.  Returns an opaque handle for phase 2 (NULL on error, *rc set). */
void *FN(droid_oracle_ba_build)(const real *poses, const real *disps, const real *intr,
                                const real *disps_sens, const real *targets, const real *weights,
                                const real *eta, int eta_rows, const int64_t *ii,
                                const int64_t *jj, int E, int nbuf, int ht, int wd, int t0, int t1,
                                int own0, int own1, int motion_only, double *Hd, double *bd,
                                int *rc) {
  FN(phase_handle) *h = (FN(phase_handle) *)calloc(1, sizeof(*h));
  h->t0 = t0;
  h->t1 = t1;
  h->motion_only = motion_only;
  *rc = FN(build_system)(poses, disps, intr, disps_sens, targets, weights, eta, eta_rows, ii, jj,
                         E, nbuf, ht, wd, t0, t1, own0, own1, motion_only, Hd, bd, &h->st, NULL,
                         NULL);
  if (*rc) {
    free(h);
    return NULL;
  }
  return h;
}

/* Phase 2: solve the (all-reduced) system, back-substitute this rank's depth slots, retract. */
int FN(droid_oracle_ba_finish)(void *handle, double *Hd, const double *bd, double lm, double ep,
                               real *poses, real *disps, real *dx_out) {
  FN(phase_handle) *h = (FN(phase_handle) *)handle;
  const int P = h->t1 - h->t0;
  real *dx = (real *)calloc(6 * P, sizeof(real));
  FN(solve_system)(Hd, bd, P, lm, ep, dx);
  if (!h->motion_only) {
    real *dz = (real *)malloc(sizeof(real) * ((size_t)h->st.M * h->st.HW + 1));
    FN(back_substitute)(&h->st, dx, dz);
    FN(retract_poses)(poses, dx, h->t0, h->t1);
    FN(retract_disps)(disps, dz, h->st.kx, h->st.M, h->st.HW);
    free(dz);
  } else {
    FN(retract_poses)(poses, dx, h->t0, h->t1);
  }
  if (dx_out) memcpy(dx_out, dx, sizeof(real) * 6 * P);
  free(dx);
  FN(free_depth_state)(&h->st);
  free(h);
  return 0;
}

/* frame_distance_kernel dk:518-657 (single pass: the `for n<1` loop runs once). */
void FN(droid_oracle_frame_distance)(const real *poses, const real *disps, const real *intr,
                                     const int64_t *ii, const int64_t *jj, int E, int ht, int wd,
                                     double beta_d, real *dist) {
  const int HW = ht * wd;
  const real fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const real beta = (real)beta_d;
  for (int e = 0; e < E; e++) {
    const int ix = (int)ii[e], jx = (int)jj[e];
    real tij[3], qij[4];
    FN(relSE3)(&poses[7 * ix], &poses[7 * ix + 3], &poses[7 * jx], &poses[7 * jx + 3], tij, qij);
    real accum = 0, valid = 0, total = 0;
    for (int k = 0; k < HW; k++) {
      const real u = (real)(k % wd), v = (real)(k / wd);
      real Xi[4], Xj[4];
      Xi[0] = (u - cx) / fx;
      Xi[1] = (v - cy) / fy;
      Xi[2] = 1;
      Xi[3] = disps[(size_t)ix * HW + k];
      FN(actSE3)(tij, qij, Xi, Xj);
      real du = fx * (Xj[0] / Xj[2]) + cx - u;
      real dv = fy * (Xj[1] / Xj[2]) + cy - v;
      real d = (real)sqrt((double)(du * du + dv * dv));
      total += beta;
      if (Xj[2] > MIN_DEPTH) {
        accum += beta * d;
        valid += beta;
      }
      Xj[0] = Xi[0] + Xi[3] * tij[0];
      Xj[1] = Xi[1] + Xi[3] * tij[1];
      Xj[2] = Xi[2] + Xi[3] * tij[2];
      du = fx * (Xj[0] / Xj[2]) + cx - u;
      dv = fy * (Xj[1] / Xj[2]) + cy - v;
      d = (real)sqrt((double)(du * du + dv * dv));
      total += (1 - beta);
      if (Xj[2] > MIN_DEPTH) {
        accum += (1 - beta) * d;
        valid += (1 - beta);
      }
    }
    dist[e] = (valid / (total + (real)1e-8) < (real)0.75) ? (real)1000.0 : accum / valid;
  }
}

/* projmap_kernel dk:427-516 */
void FN(droid_oracle_projmap)(const real *poses, const real *disps, const real *intr,
                              const int64_t *ii, const int64_t *jj, int E, int ht, int wd,
                              real *coords /* [E][HW][3] */, real *valid /* [E][HW] */) {
  const int HW = ht * wd;
  const real fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  for (int e = 0; e < E; e++) {
    const int ix = (int)ii[e], jx = (int)jj[e];
    real tij[3], qij[4];
    FN(relSE3)(&poses[7 * ix], &poses[7 * ix + 3], &poses[7 * jx], &poses[7 * jx + 3], tij, qij);
    for (int k = 0; k < HW; k++) {
      const real u = (real)(k % wd), v = (real)(k / wd);
      real Xi[4], Xj[4];
      Xi[0] = (u - cx) / fx;
      Xi[1] = (v - cy) / fy;
      Xi[2] = 1;
      Xi[3] = disps[(size_t)ix * HW + k];
      FN(actSE3)(tij, qij, Xi, Xj);
      real *c = &coords[((size_t)e * HW + k) * 3];
      c[0] = u;
      c[1] = v;
      c[2] = 0;
      if (Xj[2] > 0.01) {
        c[0] = fx * (Xj[0] / Xj[2]) + cx;
        c[1] = fy * (Xj[1] / Xj[2]) + cy;
      }
      valid[(size_t)e * HW + k] = (Xj[2] > MIN_DEPTH) ? (real)1.0 : (real)0.0;
    }
  }
}

/* iproj_kernel dk:779-850 */
void FN(droid_oracle_iproj)(const real *poses, const real *disps, const real *intr, int nm, int ht,
                            int wd, real *points /* [nm][HW][3] */) {
  const int HW = ht * wd;
  const real fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  for (int f = 0; f < nm; f++)
    for (int k = 0; k < HW; k++) {
      real Xi[4], Xj[4];
      Xi[0] = ((real)(k % wd) - cx) / fx;
      Xi[1] = ((real)(k / wd) - cy) / fy;
      Xi[2] = 1;
      Xi[3] = disps[(size_t)f * HW + k];
      FN(actSE3)(&poses[7 * f], &poses[7 * f + 3], Xi, Xj);
      real *p = &points[((size_t)f * HW + k) * 3];
      p[0] = Xj[0] / Xj[3];
      p[1] = Xj[1] / Xj[3];
      p[2] = Xj[2] / Xj[3];
    }
}

/* single-edge linearisation exposed for the Jacobian pinning tests */
void FN(droid_oracle_linearize_edge)(const real *target, const real *weight, const real *poses,
                                     const real *disps, const real *intr, int ix, int jx, int ht,
                                     int wd, real *Hs4, real *vs2, real *Eii, real *Eij, real *Cii,
                                     real *bz) {
  FN(linearize_edge)(target, weight, poses, disps, intr, ix, jx, ht, wd, Hs4, vs2, Eii, Eij, Cii,
                     bz);
}

void FN(droid_oracle_retr)(const real *xi, const real *t, const real *q, real *t1, real *q1) {
  FN(retrSE3)(xi, t, q, t1, q1);
}

#undef real
