/* or_rbd.c - rigid-body terms in pinocchio's conventions (TEST INFRASTRUCTURE; see oracle.h).
 *
 * Restates what tsid::RobotWrapper::computeAllTerms leaves in pinocchio::Data when the reference
 * calls formulation.computeProblemData(t, q, v) (main.py:119; WalkController.py:25,76):
 * forward kinematics, joint Jacobians, CRBA mass matrix (symmetrised), RNEA(q, v, 0) bias, centre
 * of mass / its velocity / its drift acceleration / its Jacobian, frame placements.
 * pinocchio is not vendored: these are the textbook recursions (Featherstone,
 * "Rigid Body Dynamics Algorithms", ch. 5-6) in pinocchio's layout:
 *   q = [p(3), quat xyzw(4), theta(20)],  v = [v_lin LOCAL(3), omega LOCAL(3), thetadot(20)],
 *   spatial vectors = [linear(3); angular(3)], joint order = pin_parent (name-sorted DFS).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

typedef struct { double R[9], p[3]; } SE3;

static void cross(const double *a, const double *b, double *c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static void matvec(const double *R, const double *v, double *o) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matTvec(const double *R, const double *v, double *o) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matmul3(const double *A, const double *B, double *C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static void se3_mul(const SE3 *a, const SE3 *b, SE3 *c) { /* c = a * b */
  SE3 t;
  matmul3(a->R, b->R, t.R);
  matvec(a->R, b->p, t.p);
  for (int i = 0; i < 3; i++) t.p[i] += a->p[i];
  *c = t;
}
/* motion transforms */
static void act_m(const SE3 *M, const double *m, double *o) {
  double l[3], w[3], c[3];
  matvec(M->R, m, l); matvec(M->R, m + 3, w); cross(M->p, w, c);
  for (int i = 0; i < 3; i++) { o[i] = l[i] + c[i]; o[3 + i] = w[i]; }
}
static void actinv_m(const SE3 *M, const double *m, double *o) {
  double c[3], t[3];
  cross(M->p, m + 3, c);
  for (int i = 0; i < 3; i++) t[i] = m[i] - c[i];
  double l[3], w[3];
  matTvec(M->R, t, l); matTvec(M->R, m + 3, w);
  for (int i = 0; i < 3; i++) { o[i] = l[i]; o[3 + i] = w[i]; }
}
/* force transforms */
static void act_f(const SE3 *M, const double *f, double *o) {
  double l[3], n[3], c[3];
  matvec(M->R, f, l); matvec(M->R, f + 3, n); cross(M->p, l, c);
  for (int i = 0; i < 3; i++) { o[i] = l[i]; o[3 + i] = n[i] + c[i]; }
}
static void cross_mm(const double *a, const double *b, double *o) { /* motion x motion */
  double t1[3], t2[3], t3[3];
  cross(a + 3, b, t1); cross(a, b + 3, t2); cross(a + 3, b + 3, t3);
  for (int i = 0; i < 3; i++) { o[i] = t1[i] + t2[i]; o[3 + i] = t3[i]; }
}
static void cross_mf(const double *v, const double *f, double *o) { /* motion x* force */
  double t1[3], t2[3], t3[3];
  cross(v + 3, f, t1); cross(v + 3, f + 3, t2); cross(v, f, t3);
  for (int i = 0; i < 3; i++) { o[i] = t1[i]; o[3 + i] = t2[i] + t3[i]; }
}
/* inertia = (m, c, I6 about com).  Y * motion -> force */
static void sym6_mul(const double *I, const double *w, double *o) {
  o[0] = I[0] * w[0] + I[1] * w[1] + I[2] * w[2];
  o[1] = I[1] * w[0] + I[3] * w[1] + I[4] * w[2];
  o[2] = I[2] * w[0] + I[4] * w[1] + I[5] * w[2];
}
static void inertia_mul(const double *Y, const double *v, double *f) {
  double cw[3], n[3], cf[3];
  cross(Y + 1, v + 3, cw);
  for (int i = 0; i < 3; i++) f[i] = Y[0] * (v[i] - cw[i]);
  sym6_mul(Y + 4, v + 3, n);
  cross(Y + 1, f, cf);
  for (int i = 0; i < 3; i++) f[3 + i] = n[i] + cf[i];
}
/* express inertia Y (given in frame B) in frame A, M = placement of B in A */
static void inertia_act(const SE3 *M, const double *Y, double *o) {
  double c[3], I[9] = {Y[4], Y[5], Y[6], Y[5], Y[7], Y[8], Y[6], Y[8], Y[9]}, T[9], RT[9];
  matvec(M->R, Y + 1, c);
  matmul3(M->R, I, T);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RT[3 * i + j] = M->R[3 * j + i];
  matmul3(T, RT, I);
  o[0] = Y[0];
  for (int i = 0; i < 3; i++) o[1 + i] = c[i] + M->p[i];
  o[4] = I[0]; o[5] = I[1]; o[6] = I[2]; o[7] = I[4]; o[8] = I[5]; o[9] = I[8];
}
static void inertia_add(double *a, const double *b) { /* a += b (same frame) */
  double m = a[0] + b[0];
  if (m <= 0) return;
  double c[3], da[3], db[3];
  for (int i = 0; i < 3; i++) c[i] = (a[0] * a[1 + i] + b[0] * b[1 + i]) / m;
  for (int i = 0; i < 3; i++) { da[i] = a[1 + i] - c[i]; db[i] = b[1 + i] - c[i]; }
  /* I += -m [d]x[d]x = m (|d|^2 I - d d^T) */
  double I[6];
  const int ix[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  double na = da[0] * da[0] + da[1] * da[1] + da[2] * da[2], nb = db[0] * db[0] + db[1] * db[1] + db[2] * db[2];
  for (int k = 0; k < 6; k++) {
    int i = ix[k][0], j = ix[k][1];
    I[k] = a[4 + k] + b[4 + k] + a[0] * ((i == j ? na : 0) - da[i] * da[j]) + b[0] * ((i == j ? nb : 0) - db[i] * db[j]);
  }
  a[0] = m;
  for (int i = 0; i < 3; i++) a[1 + i] = c[i];
  for (int k = 0; k < 6; k++) a[4 + k] = I[k];
}

static void quat_to_R(const double *q, double *R) { /* xyzw */
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double n = 1.0 / sqrt(x * x + y * y + z * z + w * w);
  x *= n; y *= n; z *= n; w *= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

static int dof_of(int j) { return j == 0 ? 0 : 5 + j; } /* first v index of joint j */

typedef struct {
  SE3 liMi[OR_NJ], oMi[OR_NJ];
  double v[OR_NJ][6], a[OR_NJ][6]; /* local spatial velocity / acceleration */
} Kin;

/* forward pass; a0w = spatial acceleration of the universe expressed in world (gravity trick) */
static void fwd_kin(const OrModel *m, const double *q, const double *v, const double *acc, const double *a0w, Kin *k) {
  for (int j = 0; j < OR_NJ; j++) {
    if (j == 0) {
      quat_to_R(q + 3, k->liMi[0].R);
      memcpy(k->liMi[0].p, q, 3 * sizeof(double));
      k->oMi[0] = k->liMi[0];
      memcpy(k->v[0], v, 6 * sizeof(double));
      actinv_m(&k->oMi[0], a0w, k->a[0]);
      if (acc) for (int i = 0; i < 6; i++) k->a[0][i] += acc[i];
      continue;
    }
    SE3 P, Rz;
    memcpy(P.R, m->pin_place[j], 9 * sizeof(double));
    memcpy(P.p, m->pin_place[j] + 9, 3 * sizeof(double));
    double c = cos(q[6 + j]), s = sin(q[6 + j]);
    double rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    memcpy(Rz.R, rz, sizeof rz);
    Rz.p[0] = Rz.p[1] = Rz.p[2] = 0;
    se3_mul(&P, &Rz, &k->liMi[j]);
    int p = m->pin_parent[j];
    se3_mul(&k->oMi[p], &k->liMi[j], &k->oMi[j]);
    double vj[6] = {0, 0, 0, 0, 0, v[5 + j]};
    actinv_m(&k->liMi[j], k->v[p], k->v[j]);
    k->v[j][5] += vj[5];
    actinv_m(&k->liMi[j], k->a[p], k->a[j]);
    double cx[6];
    cross_mm(k->v[j], vj, cx);
    for (int i = 0; i < 6; i++) k->a[j][i] += cx[i];
    if (acc) k->a[j][5] += acc[5 + j];
  }
}

void or_rnea(const OrModel *m, const double *q, const double *v, const double *a, double *tau) {
  Kin k;
  const double g0[6] = {0, 0, 9.81, 0, 0, 0}; /* -gravity as base acceleration */
  fwd_kin(m, q, v, a, g0, &k);
  double f[OR_NJ][6];
  for (int j = 0; j < OR_NJ; j++) {
    double Ya[6], Yv[6], vx[6];
    inertia_mul(m->pin_inertia[j], k.a[j], Ya);
    inertia_mul(m->pin_inertia[j], k.v[j], Yv);
    cross_mf(k.v[j], Yv, vx);
    for (int i = 0; i < 6; i++) f[j][i] = Ya[i] + vx[i];
  }
  for (int j = OR_NJ - 1; j >= 0; j--) {
    if (j == 0) { memcpy(tau, f[0], 6 * sizeof(double)); break; }
    tau[5 + j] = f[j][5];
    double fp[6];
    act_f(&k.liMi[j], f[j], fp);
    int p = m->pin_parent[j];
    for (int i = 0; i < 6; i++) f[p][i] += fp[i];
  }
}

void or_rbd_terms(const OrModel *m, const double *q, const double *v, OrTerms *t) {
  memset(t, 0, sizeof *t);
  Kin k;
  const double zero6[6] = {0};
  fwd_kin(m, q, v, NULL, zero6, &k); /* a = drift acceleration: zero joint accel, no gravity */

  /* ---- bias h = RNEA(q, v, 0) */
  double za[OR_NV] = {0};
  or_rnea(m, q, v, za, t->h);

  /* ---- CRBA */
  double Yc[OR_NJ][10];
  for (int j = 0; j < OR_NJ; j++) memcpy(Yc[j], m->pin_inertia[j], sizeof Yc[j]);
  for (int j = OR_NJ - 1; j > 0; j--) {
    double Yp[10];
    inertia_act(&k.liMi[j], Yc[j], Yp);
    inertia_add(Yc[m->pin_parent[j]], Yp);
  }
  for (int j = 0; j < OR_NJ; j++) {
    int nd = j == 0 ? 6 : 1;
    for (int d = 0; d < nd; d++) {
      double S[6] = {0}, F[6];
      S[j == 0 ? d : 5] = 1.0;
      inertia_mul(Yc[j], S, F);
      int col = dof_of(j) + d;
      int i = j;
      while (1) {
        if (i == 0) for (int r = 0; r < 6; r++) t->M[r][col] = F[r];
        else t->M[dof_of(i)][col] = F[5];
        if (i == 0) break;
        double Fp[6];
        act_f(&k.liMi[i], F, Fp);
        memcpy(F, Fp, sizeof F);
        i = m->pin_parent[i];
      }
    }
  }
  /* tsid symmetrises: lower = upper^T (RobotWrapper::computeAllTerms) */
  for (int i = 0; i < OR_NV; i++)
    for (int j = i + 1; j < OR_NV; j++) {
      /* column j holds rows of ancestors (<= j); fill whichever side is populated */
      double u = t->M[i][j];
      t->M[j][i] = u;
    }

  /* ---- world-frame joint Jacobian columns */
  double Jw[OR_NV][6];
  for (int j = 0; j < OR_NJ; j++) {
    int nd = j == 0 ? 6 : 1;
    for (int d = 0; d < nd; d++) {
      double S[6] = {0};
      S[j == 0 ? d : 5] = 1.0;
      act_m(&k.oMi[j], S, Jw[dof_of(j) + d]);
    }
  }

  /* ---- centre of mass, velocity, drift acceleration, Jacobian */
  double mass = 0;
  for (int j = 0; j < OR_NJ; j++) mass += m->pin_inertia[j][0];
  t->mass = mass;
  for (int j = 0; j < OR_NJ; j++) {
    const double *Y = m->pin_inertia[j];
    double cw[3], vl[3], wl[3], tmp[3], al[3];
    matvec(k.oMi[j].R, Y + 1, cw);
    for (int i = 0; i < 3; i++) t->com[i] += Y[0] * (cw[i] + k.oMi[j].p[i]) / mass;
    cross(k.v[j] + 3, Y + 1, tmp);
    for (int i = 0; i < 3; i++) vl[i] = k.v[j][i] + tmp[i]; /* local velocity of the com point */
    matvec(k.oMi[j].R, vl, wl);
    for (int i = 0; i < 3; i++) t->vcom[i] += Y[0] * wl[i] / mass;
    /* classical acceleration of the com point: a.lin + a.ang x c + w x v_c */
    cross(k.a[j] + 3, Y + 1, tmp);
    for (int i = 0; i < 3; i++) al[i] = k.a[j][i] + tmp[i];
    cross(k.v[j] + 3, vl, tmp);
    for (int i = 0; i < 3; i++) al[i] += tmp[i];
    matvec(k.oMi[j].R, al, wl);
    for (int i = 0; i < 3; i++) t->acom[i] += Y[0] * wl[i] / mass;
  }
  for (int j = 0; j < OR_NJ; j++) {
    /* com of body j in world; every dof on the path root..j moves it */
    const double *Y = m->pin_inertia[j];
    double cw[3];
    matvec(k.oMi[j].R, Y + 1, cw);
    for (int i = 0; i < 3; i++) cw[i] += k.oMi[j].p[i];
    for (int a = j;; a = m->pin_parent[a]) {
      int nd = a == 0 ? 6 : 1;
      for (int d = 0; d < nd; d++) {
        int col = dof_of(a) + d;
        double wxc[3];
        cross(Jw[col] + 3, cw, wxc); /* v(point) = lin(at origin) + w x point */
        for (int i = 0; i < 3; i++) t->Jcom[i][col] += Y[0] * (Jw[col][i] + wxc[i]) / mass;
      }
      if (a == 0) break;
    }
  }

  /* ---- centroidal angular momentum, body by body (independent of the composite recursion above):
   *      L_G = sum_j I_j w_j + m_j (c_j - C) x v_cj ; column `col` of A_G takes the unit motion of dof col */
  for (int j = 0; j < OR_NJ; j++) {
    const double *Y = m->pin_inertia[j];
    const double *R = k.oMi[j].R;
    double cw[3], Il[9] = {Y[4], Y[5], Y[6], Y[5], Y[7], Y[8], Y[6], Y[8], Y[9]}, Iw[9], tmp9[9];
    matvec(R, Y + 1, cw);
    for (int i = 0; i < 3; i++) cw[i] += k.oMi[j].p[i];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        tmp9[3 * r + c] = 0;
        for (int e = 0; e < 3; e++) tmp9[3 * r + c] += R[3 * r + e] * Il[3 * e + c];
      }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        Iw[3 * r + c] = 0;
        for (int e = 0; e < 3; e++) Iw[3 * r + c] += tmp9[3 * r + e] * R[3 * c + e];
      }
    double rc[3] = {cw[0] - t->com[0], cw[1] - t->com[1], cw[2] - t->com[2]};
    for (int a = j;; a = m->pin_parent[a]) {
      int nd = a == 0 ? 6 : 1;
      for (int d = 0; d < nd; d++) {
        int col = dof_of(a) + d;
        double wxc[3], vc[3], Iw_w[3], rxv[3];
        cross(Jw[col] + 3, cw, wxc);
        for (int i = 0; i < 3; i++) vc[i] = Jw[col][i] + wxc[i];
        matvec(Iw, Jw[col] + 3, Iw_w);
        cross(rc, vc, rxv);
        for (int i = 0; i < 3; i++) t->Aam[i][col] += Iw_w[i] + Y[0] * rxv[i];
      }
      if (a == 0) break;
    }
    /* drift: I alpha + w x I w + m (c - C) x a_c  with zero joint accelerations */
    double ww[3], aw[3], vl[3], al[3], tmp[3], acw[3], Iw_a[3], Iw_w2[3], wxIw[3], rxa[3];
    matvec(R, k.v[j] + 3, ww);
    matvec(R, k.a[j] + 3, aw);
    cross(k.v[j] + 3, Y + 1, tmp);
    for (int i = 0; i < 3; i++) vl[i] = k.v[j][i] + tmp[i];
    cross(k.a[j] + 3, Y + 1, tmp);
    for (int i = 0; i < 3; i++) al[i] = k.a[j][i] + tmp[i];
    cross(k.v[j] + 3, vl, tmp);
    for (int i = 0; i < 3; i++) al[i] += tmp[i];
    matvec(R, al, acw);
    matvec(Iw, aw, Iw_a);
    matvec(Iw, ww, Iw_w2);
    cross(ww, Iw_w2, wxIw);
    cross(rc, acw, rxa);
    for (int i = 0; i < 3; i++) t->dLam[i] += Iw_a[i] + wxIw[i] + Y[0] * rxa[i];
  }
  for (int i = 0; i < 3; i++)
    for (int c = 0; c < OR_NV; c++) t->Lam[i] += t->Aam[i][c] * v[c];

  /* ---- frames */
  for (int f = 0; f < OR_NF; f++) {
    int pj = m->frame_parent[f];
    SE3 P, oMf;
    memcpy(P.R, m->frame_place[f], 9 * sizeof(double));
    memcpy(P.p, m->frame_place[f] + 9, 3 * sizeof(double));
    se3_mul(&k.oMi[pj], &P, &oMf);
    memcpy(t->oMf[f], oMf.R, 9 * sizeof(double));
    memcpy(t->oMf[f] + 9, oMf.p, 3 * sizeof(double));
    actinv_m(&P, k.v[pj], t->vf[f]);
    actinv_m(&P, k.a[pj], t->af[f]);
    double wxv[3];
    cross(t->vf[f] + 3, t->vf[f], wxv);
    for (int i = 0; i < 3; i++) t->af[f][i] += wxv[i]; /* classical acceleration */
    for (int a = pj;; a = m->pin_parent[a]) {
      int nd = a == 0 ? 6 : 1;
      for (int d = 0; d < nd; d++) {
        int col = dof_of(a) + d;
        double loc[6];
        actinv_m(&oMf, Jw[col], loc);
        for (int i = 0; i < 6; i++) t->Jf[f][i][col] = loc[i];
      }
      if (a == 0) break;
    }
  }
}

/* ------------------------------------------------------------------ SE(3) exp / log, integrate */

void or_integrate(const double *q, const double *vdt, double *qout) {
  /* free-flyer: M1 = M0 * exp6(v) ; joints: theta + v */
  const double *vl = vdt, *w = vdt + 3;
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double b, c; /* V = I + b [w]x + c [w]x^2 */
  if (th < 1e-8) { b = 0.5 - th2 / 24; c = 1.0 / 6 - th2 / 120; }
  else { b = (1 - cos(th)) / th2; c = (th - sin(th)) / (th2 * th); }
  double wxv[3], wxwxv[3], pd[3];
  cross(w, vl, wxv); cross(w, wxv, wxwxv);
  for (int i = 0; i < 3; i++) pd[i] = vl[i] + b * wxv[i] + c * wxwxv[i];
  double R0[9], rp[3];
  quat_to_R(q + 3, R0);
  matvec(R0, pd, rp);
  for (int i = 0; i < 3; i++) qout[i] = q[i] + rp[i];
  /* quaternion: q0 (x) exp(w) */
  double s, cw;
  if (th < 1e-8) { s = 0.5 - th2 / 48; cw = 1 - th2 / 8; }
  else { s = sin(0.5 * th) / th; cw = cos(0.5 * th); }
  double dq[4] = {s * w[0], s * w[1], s * w[2], cw};
  const double *a = q + 3;
  double r[4];
  r[0] = a[3] * dq[0] + a[0] * dq[3] + a[1] * dq[2] - a[2] * dq[1];
  r[1] = a[3] * dq[1] - a[0] * dq[2] + a[1] * dq[3] + a[2] * dq[0];
  r[2] = a[3] * dq[2] + a[0] * dq[1] - a[1] * dq[0] + a[2] * dq[3];
  r[3] = a[3] * dq[3] - a[0] * dq[0] - a[1] * dq[1] - a[2] * dq[2];
  double n = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  for (int i = 0; i < 4; i++) qout[3 + i] = r[i] * n;
  for (int j = 0; j < OR_NA; j++) qout[7 + j] = q[7 + j] + vdt[6 + j];
}

/* log6 of a relative placement given as R row-major (9) + p (3) -> [v; w] */
void or_log6(const double *M, double *out) {
  const double *R = M, *p = M + 9;
  double tr = R[0] + R[4] + R[8];
  double ct = 0.5 * (tr - 1);
  if (ct > 1) ct = 1;
  if (ct < -1) ct = -1;
  double th = acos(ct);
  double w[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double k;
  if (th < 1e-8) k = 0.5 * (1 + th * th / 6);
  else if (th > 3.141592653589793 - 1e-6) {
    /* near pi: axis from the diagonal */
    double ax[3];
    for (int i = 0; i < 3; i++) {
      double d = (R[4 * i] - ct) / (1 - ct);
      ax[i] = sqrt(d > 0 ? d : 0);
    }
    if (w[0] < 0) ax[0] = -ax[0];
    if (w[1] < 0) ax[1] = -ax[1];
    if (w[2] < 0) ax[2] = -ax[2];
    for (int i = 0; i < 3; i++) w[i] = ax[i] * th;
    k = 0;
  } else k = 0.5 * th / sin(th);
  if (k != 0) for (int i = 0; i < 3; i++) w[i] *= k;
  /* v = Vinv p, Vinv = I - 1/2 [w]x + beta [w]x^2 */
  double th2 = th * th, beta;
  if (th < 1e-4) beta = 1.0 / 12 + th2 / 720;
  else beta = (1 - th * sin(th) / (2 * (1 - cos(th)))) / th2;
  double wxp[3], wxwxp[3];
  cross(w, p, wxp); cross(w, wxp, wxwxp);
  for (int i = 0; i < 3; i++) { out[i] = p[i] - 0.5 * wxp[i] + beta * wxwxp[i]; out[3 + i] = w[i]; }
}
