/* or_tsid.c - TSID problem assembly, dense dual active-set QP and the tick glue
 * (TEST INFRASTRUCTURE; see oracle.h).
 *
 * Follows the task stack the reference builds in ctrl/WalkController.py:54-187 and the tick in
 * main.py:119-129.  tsid / eiquadprog are not vendored; what is restated here is
 *   - tsid::InverseDynamicsFormulationAccForce::computeProblemData  (main.py:119)
 *   - tsid::Contact6d / TaskSE3Equality / TaskComEquality / TaskJointPosture /
 *     TaskActuationBounds / TaskJointBounds constraint rows        (WalkController.py:59-184)
 *   - tsid::SolverHQuadProgFast::solve data copy + eiquadprog::EiquadprogFast (Goldfarb-Idnani
 *     dual active set, as published: Math. Prog. 27 (1983) 1-33)   (main.py:121)
 *   - getActuatorForces / getAccelerations / integrate_dv / get_cop (main.py:126-132)
 * Row order of level 0: base dynamics, [LF motion, LF force], [RF motion, RF force], actuation
 * bounds, joint bounds (the order of the add* calls).  Two-sided rows become [A; -A] per block.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

static void cross(const double *a, const double *b, double *c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}

/* force-generator matrix T (6x12): Contact6d::updateForceGeneratorMatrix */
static void force_gen(const double *params, double T[6][12]) {
  memset(T, 0, sizeof(double) * 72);
  for (int i = 0; i < 4; i++) {
    const double *p = params + P_CPOINTS + 3 * i;
    for (int k = 0; k < 3; k++) T[k][3 * i + k] = 1.0;
    /* skew(p) */
    T[3][3 * i + 1] = -p[2]; T[3][3 * i + 2] = p[1];
    T[4][3 * i + 0] = p[2];  T[4][3 * i + 2] = -p[0];
    T[5][3 * i + 0] = -p[1]; T[5][3 * i + 1] = p[0];
  }
}

/* friction-cone rows B (17x12), lb, ub: Contact6d::updateForceInequalityConstraints */
static void force_cone(const double *params, double B[17][12], double *lb, double *ub) {
  const double *n = params + P_NORMAL;
  double mu = params[P_MU];
  double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, t1[3], t2[3];
  cross(n, ex, t1);
  if (sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]) < 1e-5) cross(n, ey, t1);
  cross(n, t1, t2);
  double n1 = sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
  double n2 = sqrt(t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]);
  for (int i = 0; i < 3; i++) { t1[i] /= n1; t2[i] /= n2; }
  memset(B, 0, sizeof(double) * 17 * 12);
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 3; k++) {
      B[4 * i + 0][3 * i + k] = -t1[k] - mu * n[k];
      B[4 * i + 1][3 * i + k] = t1[k] - mu * n[k];
      B[4 * i + 2][3 * i + k] = -t2[k] - mu * n[k];
      B[4 * i + 3][3 * i + k] = t2[k] - mu * n[k];
      B[16][3 * i + k] = n[k];
    }
  for (int i = 0; i < 16; i++) { lb[i] = -1e10; ub[i] = 0.0; }
  lb[16] = params[P_FMIN];
  ub[16] = params[P_FMAX];
}

/* TaskSE3Equality::compute in the LOCAL frame: a_des - drift.  ref = [p(3) Rcolmajor(9) v(6) a(6)] */
static void se3_task_rhs(const OrTerms *t, int f, const double *ref, int nref, double kp, double kd, double *rhs) {
  const double *R = t->oMf[f], *p = t->oMf[f] + 9;
  /* M^-1 * Mref */
  double Rr[9], rel[12], d[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rr[3 * i + j] = ref[3 + 3 * j + i]; /* col-major */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += R[3 * k + i] * Rr[3 * k + j];
    rel[3 * i + j] = s;
  }
  for (int i = 0; i < 3; i++) d[i] = ref[i] - p[i];
  for (int i = 0; i < 3; i++) rel[9 + i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
  double err[6];
  or_log6(rel, err);
  double vref[6] = {0}, aref[6] = {0};
  if (nref >= 24) {
    /* wMl.actInv: rotate world-aligned reference twists into the local frame */
    for (int h = 0; h < 2; h++)
      for (int i = 0; i < 3; i++) {
        vref[3 * h + i] = R[i] * ref[12 + 3 * h] + R[3 + i] * ref[13 + 3 * h] + R[6 + i] * ref[14 + 3 * h];
        aref[3 * h + i] = R[i] * ref[18 + 3 * h] + R[3 + i] * ref[19 + 3 * h] + R[6 + i] * ref[20 + 3 * h];
      }
  }
  for (int i = 0; i < 6; i++) rhs[i] = kp * err[i] + kd * (vref[i] - t->vf[f][i]) + aref[i] - t->af[f][i];
}

void or_tsid_assemble(const OrModel *m, const double *params, const OrTerms *t, const double *q,
                      const double *v, const double *com_ref, const double *posture_ref,
                      const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                      OrQP *qp) {
  or_tsid_assemble_cop(m, params, t, q, v, com_ref, posture_ref, foot_ref, contact_ref, contact_active, NULL, qp);
}

/* cop_ref (may be NULL): reference point of the CoP force task (legacy/biped.py:79-80, tsid::TaskCopEquality), used
 * when params[P_W_COP] != 0 */
void or_tsid_assemble_cop(const OrModel *m, const double *params, const OrTerms *t, const double *q,
                          const double *v, const double *com_ref, const double *posture_ref,
                          const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                          const double *cop_ref, OrQP *qp) {
  (void)m;
  memset(qp, 0, sizeof *qp);
  int nslot = 0;
  qp->slot_foot[0] = qp->slot_foot[1] = -1;
  for (int f = 0; f < OR_NF; f++) if (contact_active[f]) qp->slot_foot[nslot++] = f;
  const int k = 12 * nslot, n = OR_NV + k;
  qp->nvar = n;

  double T[6][12], B[17][12], lb[17], ub[17];
  force_gen(params, T);
  force_cone(params, B, lb, ub);

  /* Jc = T^T J  (k x nv), contact motion rhs */
  double Jc[24][OR_NV], crhs[OR_NF][6];
  for (int s = 0; s < nslot; s++) {
    int f = qp->slot_foot[s];
    for (int r = 0; r < 12; r++)
      for (int c = 0; c < OR_NV; c++) {
        double a = 0;
        for (int i = 0; i < 6; i++) a += T[i][r] * t->Jf[f][i][c];
        Jc[12 * s + r][c] = a;
      }
    se3_task_rhs(t, f, contact_ref + 12 * f, 12, params[P_KP_CONTACT], params[P_KD_CONTACT], crhs[s]);
  }

  /* ---- level 0 equalities: CE x + ce0 = 0 with ce0 = -vector */
  int ie = 0;
  for (int r = 0; r < 6; r++, ie++) { /* base dynamics: [M_u | -J_u^T] x = -h_u */
    for (int c = 0; c < OR_NV; c++) qp->CE[ie][c] = t->M[r][c];
    for (int c = 0; c < k; c++) qp->CE[ie][OR_NV + c] = -Jc[c][r];
    qp->ce0[ie] = t->h[r];
  }
  /* ---- level 0 inequalities, in add order */
  int ii = 0;
  for (int s = 0; s < nslot; s++) {
    int f = qp->slot_foot[s];
    for (int r = 0; r < 6; r++, ie++) {
      for (int c = 0; c < OR_NV; c++) qp->CE[ie][c] = t->Jf[f][r][c];
      qp->ce0[ie] = -crhs[s][r];
    }
    for (int r = 0; r < 17; r++, ii++) { /* lower: B f - lb >= 0 */
      for (int c = 0; c < 12; c++) qp->CI[ii][OR_NV + 12 * s + c] = B[r][c];
      qp->ci0[ii] = -lb[r];
    }
    for (int r = 0; r < 17; r++, ii++) { /* upper: -B f + ub >= 0 */
      for (int c = 0; c < 12; c++) qp->CI[ii][OR_NV + 12 * s + c] = -B[r][c];
      qp->ci0[ii] = ub[r];
    }
  }
  /* actuation bounds: tau_min - h_a <= [M_a | -J_a^T] x <= tau_max - h_a */
  for (int sgn = 0; sgn < 2; sgn++)
    for (int r = 0; r < OR_NA; r++, ii++) {
      double s = sgn ? -1.0 : 1.0;
      for (int c = 0; c < OR_NV; c++) qp->CI[ii][c] = s * t->M[6 + r][c];
      for (int c = 0; c < k; c++) qp->CI[ii][OR_NV + c] = -s * Jc[c][6 + r];
      double lo = -params[P_TAU_MAX + r] - t->h[6 + r], hi = params[P_TAU_MAX + r] - t->h[6 + r];
      qp->ci0[ii] = sgn ? hi : -lo;
    }
  /* joint bounds: TaskJointBounds (dt doubled inside the task), base rows +-1e10 */
  double dt2 = 2.0 * params[P_DT];
  for (int sgn = 0; sgn < 2; sgn++)
    for (int r = 0; r < OR_NV; r++, ii++) {
      double lo = -1e10, hi = 1e10;
      if (r >= 6) {
        double amax = (params[P_V_MAX + r - 6] - v[r]) / dt2, amin = (-params[P_V_MAX + r - 6] - v[r]) / dt2;
        hi = amax < 1e10 ? amax : 1e10;
        lo = amin > -1e10 ? amin : -1e10;
      }
      qp->CI[ii][r] = sgn ? -1.0 : 1.0;
      qp->ci0[ii] = sgn ? hi : -lo;
    }
  qp->neq = ie;
  qp->nin = ii;

  /* ---- level 1 cost: H = sum w A^T A + reg I ; g = -sum w A^T a */
  double A[20][OR_NVAR], a[20];
#define ACCUM(rows, w)                                                             \
  for (int r_ = 0; r_ < (rows); r_++)                                              \
    for (int i_ = 0; i_ < n; i_++) {                                               \
      if (A[r_][i_] == 0) continue;                                                \
      qp->g[i_] -= (w) * A[r_][i_] * a[r_];                                        \
      for (int j_ = 0; j_ < n; j_++) qp->H[i_][j_] += (w) * A[r_][i_] * A[r_][j_]; \
    }
  const double wreg[6] = {1, 1, 1e-3, 2, 2, 2}; /* Contact6d force-regularisation weights */
  for (int s = 0; s < nslot; s++) {
    int f = qp->slot_foot[s];
    /* force regularisation: diag(wreg) T f = 0 */
    memset(A, 0, sizeof A); memset(a, 0, sizeof a);
    for (int r = 0; r < 6; r++) for (int c = 0; c < 12; c++) A[r][OR_NV + 12 * s + c] = wreg[r] * T[r][c];
    ACCUM(6, params[P_W_FORCEREF]);
    (void)f;
  }
  if (params[P_W_COP] != 0.0 && cop_ref && nslot > 0) {
    /* CoP force task: the tangential moment of the contact forces about the reference point vanishes,
     * [n]x sum_c sum_i (p_c + R_c r_i - p_ref) x (R_c f_ci) = 0 - three rows (rank 2) over all force variables,
     * as a level-1 cost with weight w_cop (addForceTask(copTask, w_cop, 1), legacy/biped.py:80) */
    memset(A, 0, sizeof A); memset(a, 0, sizeof a);
    const double *nn = params + P_NORMAL;
    for (int s = 0; s < nslot; s++) {
      const int f = qp->slot_foot[s];
      const double *R = t->oMf[f], *pc = t->oMf[f] + 9;
      for (int i = 0; i < 4; i++) {
        const double *r = params + P_CPOINTS + 3 * i;
        double d[3];
        for (int k2 = 0; k2 < 3; k2++) d[k2] = pc[k2] + R[3 * k2] * r[0] + R[3 * k2 + 1] * r[1] + R[3 * k2 + 2] * r[2] - cop_ref[k2];
        for (int j = 0; j < 3; j++) { /* unit force along local axis j of the contact frame */
          const double fw[3] = {R[j], R[3 + j], R[6 + j]};
          const double mo[3] = {d[1] * fw[2] - d[2] * fw[1], d[2] * fw[0] - d[0] * fw[2], d[0] * fw[1] - d[1] * fw[0]};
          const int col = OR_NV + 12 * s + 3 * i + j;
          A[0][col] = nn[1] * mo[2] - nn[2] * mo[1];
          A[1][col] = nn[2] * mo[0] - nn[0] * mo[2];
          A[2][col] = nn[0] * mo[1] - nn[1] * mo[0];
        }
      }
    }
    ACCUM(3, params[P_W_COP]);
  }
  for (int f = 0; f < OR_NF; f++) { /* foot SE3 tasks (always in the stack) */
    memset(A, 0, sizeof A);
    for (int r = 0; r < 6; r++) for (int c = 0; c < OR_NV; c++) A[r][c] = t->Jf[f][r][c];
    se3_task_rhs(t, f, foot_ref + 24 * f, 24, params[P_KP_FOOT], params[P_KD_FOOT], a);
    ACCUM(6, params[P_W_FOOT]);
  }
  { /* CoM */
    memset(A, 0, sizeof A);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < OR_NV; c++) A[r][c] = t->Jcom[r][c];
      a[r] = -params[P_KP_COM] * (t->com[r] - com_ref[r]) - params[P_KD_COM] * (t->vcom[r] - com_ref[3 + r]) +
             com_ref[6 + r] - t->acom[r];
    }
    ACCUM(3, params[P_W_COM]);
  }
  if (params[P_W_AM] != 0.0) { /* angular momentum (legacy/biped.py:82-87; tsid::TaskAMEquality with a zero reference):
                                * A_G,ang dv = -Kp L - drift.  setKd is stored and never used by the task. */
    memset(A, 0, sizeof A);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < OR_NV; c++) A[r][c] = t->Aam[r][c];
      a[r] = -params[P_KP_AM + r] * t->Lam[r] - t->dLam[r];
    }
    ACCUM(3, params[P_W_AM]);
  }
  { /* posture */
    memset(A, 0, sizeof A);
    for (int r = 0; r < OR_NA; r++) {
      A[r][6 + r] = 1.0;
      a[r] = -params[P_KP_POSTURE + r] * (q[7 + r] - posture_ref[r]) - params[P_KD_POSTURE + r] * v[6 + r];
    }
    ACCUM(OR_NA, params[P_W_POSTURE]);
  }
#undef ACCUM
  for (int i = 0; i < n; i++) qp->H[i][i] += params[P_HESS_REG];
}

/* ================================================================= Goldfarb-Idnani dual active set
 * min 1/2 x'Gx + g0'x  s.t. CE x + ce0 = 0, CI x + ci0 >= 0.  Structure and tie-breaking follow
 * eiquadprog-fast: J = L^-T updated by Givens rotations, R the triangular factor of the active
 * normals, most-violated selection in row order, status codes 0 optimal / 1 infeasible /
 * 2 unbounded / 3 max-iter / 4 redundant equalities. */
#define NMAX OR_NVAR
typedef struct {
  int n;
  double J[NMAX][NMAX], R[NMAX][NMAX], d[NMAX], z[NMAX], r[NMAX], np[NMAX];
  double R_norm;
} GI;

static void compute_d(GI *w) {
  for (int j = 0; j < w->n; j++) {
    double s = 0;
    for (int i = 0; i < w->n; i++) s += w->J[i][j] * w->np[i];
    w->d[j] = s;
  }
}
static void update_z(GI *w, int iq) {
  for (int i = 0; i < w->n; i++) {
    double s = 0;
    for (int j = iq; j < w->n; j++) s += w->J[i][j] * w->d[j];
    w->z[i] = s;
  }
}
static void update_r(GI *w, int iq) {
  for (int i = iq - 1; i >= 0; i--) {
    double s = w->d[i];
    for (int j = i + 1; j < iq; j++) s -= w->R[i][j] * w->r[j];
    w->r[i] = s / w->R[i][i];
  }
}
static int add_constraint(GI *w, int *iq) {
  int n = w->n;
  for (int j = n - 1; j >= *iq + 1; j--) {
    double cc = w->d[j - 1], ss = w->d[j], h = hypot(cc, ss);
    if (h == 0.0) continue;
    w->d[j] = 0.0;
    ss /= h; cc /= h;
    if (cc < 0.0) { cc = -cc; ss = -ss; w->d[j - 1] = -h; }
    else w->d[j - 1] = h;
    double xny = ss / (1.0 + cc);
    for (int k = 0; k < n; k++) {
      double t1 = w->J[k][j - 1], t2 = w->J[k][j];
      w->J[k][j - 1] = t1 * cc + t2 * ss;
      w->J[k][j] = xny * (t1 + w->J[k][j - 1]) - t2;
    }
  }
  (*iq)++;
  for (int i = 0; i < *iq; i++) w->R[i][*iq - 1] = w->d[i];
  if (fabs(w->d[*iq - 1]) <= DBL_EPSILON * w->R_norm) return 0;
  if (fabs(w->d[*iq - 1]) > w->R_norm) w->R_norm = fabs(w->d[*iq - 1]);
  return 1;
}
static void delete_constraint(GI *w, int *A, double *u, int p, int *iq, int l) {
  int n = w->n, qq = -1;
  for (int i = p; i < *iq; i++) if (A[i] == l) { qq = i; break; }
  if (qq < 0) return;
  for (int i = qq; i < *iq - 1; i++) {
    A[i] = A[i + 1]; u[i] = u[i + 1];
    for (int j = 0; j < n; j++) w->R[j][i] = w->R[j][i + 1];
  }
  A[*iq - 1] = A[*iq]; u[*iq - 1] = u[*iq];
  A[*iq] = 0; u[*iq] = 0.0;
  for (int j = 0; j < *iq; j++) w->R[j][*iq - 1] = 0.0;
  (*iq)--;
  if (*iq == 0) return;
  for (int j = qq; j < *iq; j++) {
    double cc = w->R[j][j], ss = w->R[j + 1][j], h = hypot(cc, ss);
    if (h == 0.0) continue;
    cc /= h; ss /= h;
    w->R[j + 1][j] = 0.0;
    if (cc < 0.0) { w->R[j][j] = -h; cc = -cc; ss = -ss; }
    else w->R[j][j] = h;
    double xny = ss / (1.0 + cc);
    for (int k = j + 1; k < *iq; k++) {
      double t1 = w->R[j][k], t2 = w->R[j + 1][k];
      w->R[j][k] = t1 * cc + t2 * ss;
      w->R[j + 1][k] = xny * (t1 + w->R[j][k]) - t2;
    }
    for (int k = 0; k < n; k++) {
      double t1 = w->J[k][j], t2 = w->J[k][j + 1];
      w->J[k][j] = t1 * cc + t2 * ss;
      w->J[k][j + 1] = xny * (w->J[k][j] + t1) - t2;
    }
  }
}

int or_qp_solve(const OrQP *qp, int max_iter, OrQPSol *sol) {
  const int n = qp->nvar, p = qp->neq, mi = qp->nin;
  static __thread GI w;
  memset(&w, 0, sizeof w);
  memset(sol, 0, sizeof *sol);
  w.n = n;
  double *x = sol->x, *u = sol->u;
  int *A = sol->A;
  double L[NMAX][NMAX];
  double c1 = 0, c2 = 0;
  for (int i = 0; i < n; i++) c1 += qp->H[i][i];
  /* Cholesky G = L L^T */
  memset(L, 0, sizeof L);
  for (int j = 0; j < n; j++) {
    double s = qp->H[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (!(s > 0.0)) { sol->status = 2; return 2; }
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = qp->H[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  /* J = L^-T : solve L^T J = I column by column (upper triangular back substitution) */
  for (int c = 0; c < n; c++)
    for (int i = n - 1; i >= 0; i--) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = i + 1; k < n; k++) s -= L[k][i] * w.J[k][c];
      w.J[i][c] = s / L[i][i];
    }
  for (int i = 0; i < n; i++) c2 += w.J[i][i];
  w.R_norm = 1.0;
  /* x = -G^-1 g0 */
  {
    double y[NMAX];
    for (int i = 0; i < n; i++) {
      double s = -qp->g[i];
      for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
      y[i] = s / L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
      x[i] = s / L[i][i];
    }
  }
  double f_value = 0;
  for (int i = 0; i < n; i++) f_value += 0.5 * qp->g[i] * x[i];
  int iq = 0;
  /* equality constraints */
  for (int i = 0; i < p; i++) {
    for (int c = 0; c < n; c++) w.np[c] = qp->CE[i][c];
    compute_d(&w);
    update_z(&w, iq);
    update_r(&w, iq);
    double zz = 0, znp = 0, npx = 0;
    for (int c = 0; c < n; c++) { zz += w.z[c] * w.z[c]; znp += w.z[c] * w.np[c]; npx += w.np[c] * x[c]; }
    double t2 = 0.0;
    if (fabs(zz) > DBL_EPSILON) t2 = (-npx - qp->ce0[i]) / znp;
    for (int c = 0; c < n; c++) x[c] += t2 * w.z[c];
    u[iq] = t2;
    for (int c = 0; c < iq; c++) u[c] -= t2 * w.r[c];
    f_value += 0.5 * t2 * t2 * znp;
    A[i] = -i - 1;
    if (!add_constraint(&w, &iq)) { sol->status = 4; sol->iq = iq; return 4; }
  }
  int iai[OR_NIN], iaexcl[OR_NIN], A_old[OR_NEQ + OR_NIN];
  double s[OR_NIN], u_old[OR_NEQ + OR_NIN], x_old[NMAX];
  for (int i = 0; i < mi; i++) iai[i] = i;
  int iter = 0, ip = 0, l = 0;
  double ss, psi, t, t1, t2;

l1:
  iter++;
  if (iter >= max_iter) { sol->status = 3; goto done; }
  for (int i = p; i < iq; i++) iai[A[i]] = -1;
  ss = 0.0; psi = 0.0; ip = 0;
  for (int i = 0; i < mi; i++) {
    double a = qp->ci0[i];
    for (int c = 0; c < n; c++) a += qp->CI[i][c] * x[c];
    s[i] = a;
    iaexcl[i] = 1;
    psi += a < 0.0 ? a : 0.0;
  }
  if (fabs(psi) <= mi * DBL_EPSILON * c1 * c2 * 100.0) { sol->status = 0; goto done; }
  memcpy(u_old, u, sizeof(double) * iq);
  memcpy(A_old, A, sizeof(int) * iq);
  memcpy(x_old, x, sizeof(double) * n);

l2:
  for (int i = 0; i < mi; i++)
    if (s[i] < ss && iai[i] != -1 && iaexcl[i]) { ss = s[i]; ip = i; }
  if (ss >= 0.0) { sol->status = 0; goto done; }
  for (int c = 0; c < n; c++) w.np[c] = qp->CI[ip][c];
  u[iq] = 0.0;
  A[iq] = ip;

l2a:
  compute_d(&w);
  update_z(&w, iq);
  update_r(&w, iq);
  l = 0;
  t1 = INFINITY;
  for (int k = p; k < iq; k++)
    if (w.r[k] > 0.0 && u[k] / w.r[k] < t1) { t1 = u[k] / w.r[k]; l = A[k]; }
  {
    double zz = 0, znp = 0;
    for (int c = 0; c < n; c++) { zz += w.z[c] * w.z[c]; znp += w.z[c] * w.np[c]; }
    t2 = fabs(zz) > DBL_EPSILON ? -s[ip] / znp : INFINITY;
    t = t1 < t2 ? t1 : t2;
    if (t >= INFINITY) { sol->status = 1; goto done; }
    if (t2 >= INFINITY) { /* dual step only */
      for (int k = 0; k < iq; k++) u[k] -= t * w.r[k];
      u[iq] += t;
      iai[l] = l;
      delete_constraint(&w, A, u, p, &iq, l);
      goto l2a;
    }
    for (int c = 0; c < n; c++) x[c] += t * w.z[c];
    f_value += t * znp * (0.5 * t + u[iq]);
    for (int k = 0; k < iq; k++) u[k] -= t * w.r[k];
    u[iq] += t;
  }
  if (t == t2) { /* full step: constraint ip becomes active */
    if (!add_constraint(&w, &iq)) {
      iaexcl[ip] = 0;
      delete_constraint(&w, A, u, p, &iq, ip);
      for (int i = 0; i < mi; i++) iai[i] = i;
      for (int i = 0; i < iq; i++) { A[i] = A_old[i]; if (A[i] >= 0) iai[A[i]] = -1; u[i] = u_old[i]; }
      memcpy(x, x_old, sizeof(double) * n);
      goto l2;
    }
    iai[ip] = -1;
    goto l1;
  }
  /* partial step: drop constraint l, re-evaluate s[ip] */
  iai[l] = l;
  delete_constraint(&w, A, u, p, &iq, l);
  {
    double a = qp->ci0[ip];
    for (int c = 0; c < n; c++) a += qp->CI[ip][c] * x[c];
    s[ip] = a;
  }
  goto l2a;

done:
  sol->iq = iq;
  sol->iter = iter;
  sol->f_value = f_value;
  return sol->status;
}

/* ================================================================= tick */
/* sole placements (R row-major 9 + p 3, per foot) of the last or_tsid_tick on this thread: what the device tick
 * writes to `frames` (the data of this tick's computeProblemData, before integration) */
__thread double or_last_frames[24];
/* reward and done flag of the last or_tsid_tick on this thread (SURVEY.md 8d write list; no reference
 * counterpart): reward = exp(-|com - com_ref|^2 / sigma^2) - c_tau |tau|^2 on this tick's data, done = failed QP,
 * base lower than DONE_HEIGHT or base z-axis . world z below DONE_TILT at the new state */
__thread double or_last_rowx[2];

int or_tsid_tick(const OrModel *m, const double *params, double *q, double *v, const double *com_ref,
                 const double *posture_ref, const double *foot_ref, const double *contact_ref,
                 const uint8_t *contact_active, const double *cop_frames, double *tau, double *dv,
                 double *f, double *obs, int *iters) {
  return or_tsid_tick_cop(m, params, q, v, com_ref, posture_ref, foot_ref, contact_ref, contact_active, cop_frames, NULL, tau,
                          dv, f, obs, iters);
}

int or_tsid_tick_cop(const OrModel *m, const double *params, double *q, double *v, const double *com_ref,
                     const double *posture_ref, const double *foot_ref, const double *contact_ref,
                     const uint8_t *contact_active, const double *cop_frames, const double *cop_ref, double *tau,
                     double *dv, double *f, double *obs, int *iters) {
  static __thread OrTerms t;
  static __thread OrQP qp;
  static __thread OrQPSol sol;
  { /* a non-finite state or reference never enters the solver: HQP_STATUS_ERROR, state untouched, outputs zeroed */
    double chk = 0;
    for (int i = 0; i < OR_NQ; i++) chk += fabs(q[i]);
    for (int i = 0; i < OR_NV; i++) chk += fabs(v[i]);
    for (int i = 0; i < 9; i++) chk += fabs(com_ref[i]);
    for (int i = 0; i < OR_NA; i++) chk += fabs(posture_ref[i]);
    for (int i = 0; i < 48; i++) chk += fabs(foot_ref[i]);
    for (int i = 0; i < 24; i++) chk += fabs(contact_ref[i]);
    if (!(chk <= 1e300)) {
      if (iters) *iters = 0;
      memset(tau, 0, OR_NA * sizeof(double)); /* as after a failed solve: nothing of an earlier tick is handed on */
      memset(dv, 0, OR_NV * sizeof(double));
      memset(f, 0, 24 * sizeof(double));
      or_last_rowx[0] = 0.0; or_last_rowx[1] = 1.0;
      return 4;
    }
  }
  or_rbd_terms(m, q, v, &t);
  for (int i = 6; i < OR_NV; i++) t.M[i][i] += params[P_TSID_ARMATURE]; /* closed-loop knob, 0 by default */
  memcpy(or_last_frames, t.oMf, sizeof or_last_frames);
  or_tsid_assemble_cop(m, params, &t, q, v, com_ref, posture_ref, foot_ref, contact_ref, contact_active, cop_ref, &qp);
  int status = or_qp_solve(&qp, (int)params[P_MAX_ITER], &sol);
  if (iters) *iters = sol.iter;
  /* a failed QP (the reference stops its loop there, main.py:122-124, before it reads the solution): dv = f = tau = 0 */
  if (status != 0) memset(sol.x, 0, sizeof sol.x);
  memset(f, 0, 24 * sizeof(double));
  for (int i = 0; i < OR_NV; i++) dv[i] = sol.x[i];
  for (int s = 0; s < 2; s++)
    if (qp.slot_foot[s] >= 0)
      for (int c = 0; c < 12; c++) f[12 * qp.slot_foot[s] + c] = sol.x[OR_NV + 12 * s + c];
  /* tau = M_a dv + h_a - J_a^T f   (getActuatorForces) */
  double T[6][12];
  force_gen(params, T);
  for (int r = 0; r < OR_NA; r++) {
    double a = t.h[6 + r];
    for (int c = 0; c < OR_NV; c++) a += t.M[6 + r][c] * dv[c];
    for (int fo = 0; fo < OR_NF; fo++) {
      if (!contact_active[fo]) continue;
      for (int c = 0; c < 12; c++) {
        double jc = 0;
        for (int i = 0; i < 6; i++) jc += T[i][c] * t.Jf[fo][i][6 + r];
        a -= jc * f[12 * fo + c];
      }
    }
    if (params[P_FRICTION_COMP] != 0.0) { /* closed-loop knob: Coulomb-friction feed-forward */
      const double sat = (v[6 + r] + params[P_DT] * dv[6 + r]) * 20.0;
      a += params[P_FRICTION_COMP] * (sat > 1 ? 1.0 : (sat < -1 ? -1.0 : sat));
    }
    tau[r] = status == 0 ? a : 0.0;
  }
  /* observations use the data of this tick's computeProblemData (before integration), as
   * main.py:132-142 reads formulation.data() after integrate_dv without recomputing */
  if (obs) {
    double cop[3] = {0, 0, 0};
    double w[OR_NF][6], fz[OR_NF] = {0, 0}, copw[OR_NF][3];
    for (int fo = 0; fo < OR_NF; fo++) {
      for (int i = 0; i < 6; i++) {
        w[fo][i] = 0;
        for (int c = 0; c < 12; c++) w[fo][i] += T[i][c] * f[12 * fo + c];
      }
      double cl[3] = {0, 0, 0};
      if (contact_active[fo] && w[fo][2] > 1e-3) { cl[0] = w[fo][4] / w[fo][2]; cl[1] = w[fo][3] / w[fo][2]; }
      const double *F = (params[P_QUIRKS] != 0.0 && cop_frames) ? cop_frames + 12 * fo : t.oMf[fo];
      for (int i = 0; i < 3; i++) copw[fo][i] = F[3 * i] * cl[0] + F[3 * i + 1] * cl[1] + F[3 * i + 2] * cl[2] + F[9 + i];
      fz[fo] = w[fo][2];
    }
    if (contact_active[0] && contact_active[1] && fz[0] + fz[1] != 0.0) {
      cop[0] = (copw[0][0] * fz[0] + copw[1][0] * fz[1]) / (fz[0] + fz[1]);
      cop[1] = (copw[0][1] * fz[0] + copw[1][1] * fz[1]) / (fz[0] + fz[1]);
    }
    memcpy(obs + OR_NQ + OR_NV, t.com, 3 * sizeof(double));
    memcpy(obs + OR_NQ + OR_NV + 3, cop, 3 * sizeof(double));
    memcpy(obs + OR_NQ + OR_NV + 6, t.oMf[0] + 9, 3 * sizeof(double));
    memcpy(obs + OR_NQ + OR_NV + 9, t.oMf[1] + 9, 3 * sizeof(double));
  }
  if (status == 0) {
    /* integrate_dv: v_mean = v + dt/2 dv ; v += dt dv ; q = integrate(q, dt v_mean) */
    double dt = params[P_DT], vm[OR_NV], qn[OR_NQ];
    for (int i = 0; i < OR_NV; i++) { vm[i] = dt * (v[i] + 0.5 * dt * dv[i]); v[i] += dt * dv[i]; }
    or_integrate(q, vm, qn);
    memcpy(q, qn, sizeof qn);
  }
  if (obs) {
    memcpy(obs, q, OR_NQ * sizeof(double));
    memcpy(obs + OR_NQ, v, OR_NV * sizeof(double));
  }
  or_last_rowx[0] = 0.0; or_last_rowx[1] = 1.0;
  if (status == 0) {
    const double up = 1.0 - 2.0 * (q[3] * q[3] + q[4] * q[4]);
    const int fall = q[2] < params[P_DONE_HEIGHT] || up < params[P_DONE_TILT];
    double e2 = 0, tau2 = 0;
    for (int i = 0; i < 3; i++) e2 += (t.com[i] - com_ref[i]) * (t.com[i] - com_ref[i]);
    for (int i = 0; i < OR_NA; i++) tau2 += tau[i] * tau[i];
    or_last_rowx[1] = fall ? 1.0 : 0.0;
    or_last_rowx[0] = fall ? 0.0 : exp(-e2 / (params[P_REW_SIGMA] * params[P_REW_SIGMA])) - params[P_REW_CTAU] * tau2;
  }
  return status;
}
