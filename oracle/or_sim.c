/* or_sim.c - MuJoCo-subset forward dynamics + contact step (TEST INFRASTRUCTURE; see oracle.h).
 *
 * Restates what mujoco.mj_step(mj_model, mj_data) (main.py:195) does for the model that
 * robot/v1/mujoco/scene.xml + robot.xml compile to: one free joint + 20 z-hinges,
 * armature, frictionloss rows, position actuators (kp, kv from dampratio), floor-plane <-> convex
 * hull contacts, soft constraints (solref/solimp impedance), pyramidal cones (condim 3), Newton
 * solver with warm start, semi-implicit Euler - and, for robot/v0/robot.xml (-DOR_ROBOT_V0): several
 * collision geoms per body, condim 4 (torsional friction: two more pyramid rows per contact), joint damping
 * as a passive force integrated implicitly (mj_Euler's (M + h B) solve), the geom margin, and the position
 * actuators' ctrlrange / forcerange clamps.  mujoco is not vendored; this follows
 * the published algorithm (MuJoCo documentation, "Computation" chapter; Todorov 2014) with every
 * option at its default except timestep (main.py:52).
 *
 * Declared deviations (implementation-defined in MuJoCo, stated here so the GPU path can match):
 *  - plane<->hull support vertex: exhaustive arg-min over the hull's vertices; vertices within
 *    1e-9 m of the minimum tie-break to the lowest index (MuJoCo hill-climbs the hull graph).
 *  - line search: safeguarded Newton on the exact 1-D piecewise-quadratic cost to MuJoCo's
 *    gradient tolerance (MuJoCo's bracketing schedule differs; both stop at |dcost| < gtol).
 *  - robot<->robot hull pairs: or_collide.c (MPR run to 1e-10 instead of MuJoCo's 1e-6 tolerance).
 *  - at most OR_MAXCON contacts per env, of which at most OR_MAXHH robot<->robot ones (flag bit 8 when a
 *    penetrating pair had to be dropped).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define NV OR_NV
#define NB OR_NB
static const double MINVAL = 1e-15;
#define TIE_TOL 1e-9

static void cross(const double *a, const double *b, double *c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matvec(const double *R, const double *v, double *o) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matmul3(const double *A, const double *B, double *C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static void quat_wxyz_to_R(const double *q, double *R) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double n = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= n; x *= n; y *= n; z *= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
/* spatial (Pluecker, world origin) helpers: [lin; ang] */
static void cross_mm(const double *a, const double *b, double *o) {
  double t1[3], t2[3], t3[3];
  cross(a + 3, b, t1); cross(a, b + 3, t2); cross(a + 3, b + 3, t3);
  for (int i = 0; i < 3; i++) { o[i] = t1[i] + t2[i]; o[3 + i] = t3[i]; }
}
static void cross_mf(const double *v, const double *f, double *o) {
  double t1[3], t2[3], t3[3];
  cross(v + 3, f, t1); cross(v + 3, f + 3, t2); cross(v, f, t3);
  for (int i = 0; i < 3; i++) { o[i] = t1[i]; o[3 + i] = t2[i] + t3[i]; }
}
/* Y = (m, c, I3x3 about c) all in world axes; Y * twist(at world origin) -> wrench (at world origin) */
typedef struct { double m, c[3], I[9]; } WInertia;
static void wi_mul(const WInertia *Y, const double *v, double *f) {
  double cw[3], n[3], cf[3];
  cross(Y->c, v + 3, cw);
  for (int i = 0; i < 3; i++) f[i] = Y->m * (v[i] - cw[i]);
  matvec(Y->I, v + 3, n);
  cross(Y->c, f, cf);
  for (int i = 0; i < 3; i++) f[3 + i] = n[i] + cf[i];
}
static void wi_add(WInertia *a, const WInertia *b) {
  double m = a->m + b->m, c[3], da[3], db[3];
  for (int i = 0; i < 3; i++) c[i] = (a->m * a->c[i] + b->m * b->c[i]) / m;
  for (int i = 0; i < 3; i++) { da[i] = a->c[i] - c[i]; db[i] = b->c[i] - c[i]; }
  double na = dot3(da, da), nb = dot3(db, db);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      a->I[3 * i + j] += b->I[3 * i + j] + a->m * ((i == j ? na : 0) - da[i] * da[j]) + b->m * ((i == j ? nb : 0) - db[i] * db[j]);
  a->m = m;
  memcpy(a->c, c, sizeof c);
}

static int chol(double A[NV][NV], int n) { /* in place lower Cholesky */
  for (int j = 0; j < n; j++) {
    double s = A[j][j];
    for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
    if (!(s > 0)) return -1;
    A[j][j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = A[i][j];
      for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
      A[i][j] = t / A[j][j];
    }
  }
  return 0;
}
static void chol_solve(double L[NV][NV], int n, const double *b, double *x) {
  double y[NV];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

/* rank-1 update (sigma = +1) or downdate (-1) of a lower Cholesky factor: L L^T <- L L^T + sigma x x^T (x is destroyed).
 * Returns -1 when a downdate loses positive definiteness (the caller then rebuilds the factor). */
static int chol_rank1(double L[NV][NV], int n, double *x, double sigma) {
  for (int k = 0; k < n; k++) {
    if (x[k] == 0.0) continue;
    /* s = x_k / L_kk, c = sqrt(1 + sigma s^2) = the diagonal's growth factor.  A downdate that takes the pivot to (nearly)
     * nothing - 1 - s^2 <= 1e-10: the row carried all but that fraction of the diagonal - has lost its digits to
     * cancellation: rebuild instead (same rule on the device) */
    const double s = x[k] / L[k][k], q = 1.0 + sigma * s * s;
    if (!(q > 1e-10)) return -1;
    const double c = sqrt(q), r = L[k][k] * c;
    L[k][k] = r;
    for (int i = k + 1; i < n; i++) {
      L[i][k] = (L[i][k] + sigma * s * x[i]) / c;
      x[i] = c * x[i] - s * L[i][k];
    }
  }
  return 0;
}

/* diagnostics (tools/newton_stats.py): rows whose state changed per Newton iteration after the first, all threads */
int or_newton_incr_max = OR_NEWTON_INCR_MAX;
long or_newton_hist[OR_MAXEFC + 2];
void or_newton_hist_get(long *out, int reset) {
  for (int i = 0; i < OR_MAXEFC + 2; i++) { out[i] = or_newton_hist[i]; if (reset) or_newton_hist[i] = 0; }
}

typedef struct {
  int nefc, ncon;
  double J[OR_MAXEFC][NV], aref[OR_MAXEFC], D[OR_MAXEFC], R[OR_MAXEFC], floss[OR_MAXEFC];
  int type[OR_MAXEFC]; /* 0 friction-dof, 1 pyramidal contact row */
} Efc;

/* constraint cost / force / active flags for given jar */
static double efc_update(const Efc *e, const double *jar, double *force, int *active) {
  double cost = 0;
  for (int i = 0; i < e->nefc; i++) {
    if (e->type[i] == 0) {
      double f = e->floss[i], r = e->R[i];
      if (jar[i] <= -r * f) { force[i] = f; active[i] = 0; cost += -0.5 * r * f * f - f * jar[i]; }
      else if (jar[i] >= r * f) { force[i] = -f; active[i] = 0; cost += -0.5 * r * f * f + f * jar[i]; }
      else { force[i] = -e->D[i] * jar[i]; active[i] = 1; cost += 0.5 * e->D[i] * jar[i] * jar[i]; }
    } else {
      if (jar[i] < 0) { force[i] = -e->D[i] * jar[i]; active[i] = 1; cost += 0.5 * e->D[i] * jar[i] * jar[i]; }
      else { force[i] = 0; active[i] = 0; }
    }
  }
  return cost;
}

/* value, first and second derivative of the total cost along the search direction at alpha */
static void ls_eval(const Efc *e, const double *jar, const double *Jv, const double *qg, double alpha,
                    double *cost, double *d1, double *d2) {
  double c = alpha * alpha * qg[2] + alpha * qg[1] + qg[0], g = 2 * alpha * qg[2] + qg[1], h = 2 * qg[2];
  for (int i = 0; i < e->nefc; i++) {
    double x = jar[i] + alpha * Jv[i];
    if (e->type[i] == 0) {
      double f = e->floss[i], r = e->R[i];
      if (x <= -r * f) { c += -0.5 * r * f * f - f * x; g += -f * Jv[i]; }
      else if (x >= r * f) { c += -0.5 * r * f * f + f * x; g += f * Jv[i]; }
      else { c += 0.5 * e->D[i] * x * x; g += e->D[i] * x * Jv[i]; h += e->D[i] * Jv[i] * Jv[i]; }
    } else if (x < 0) {
      c += 0.5 * e->D[i] * x * x; g += e->D[i] * x * Jv[i]; h += e->D[i] * Jv[i] * Jv[i];
    }
  }
  *cost = c; *d1 = g; *d2 = h;
}

/* envp (may be NULL = nominal): per-env randomisation of BASELINE config 5 -
 * [0] mass scale (every sim body's mass and inertia), [1] contact friction, [2..4] unit floor normal,
 * [5] floor offset d (plane n.x = d).  Compile-time constants derived at qpos0 (inverse weights, kv,
 * meaninertia) stay nominal, as when body_mass is edited in a compiled MuJoCo model. */
int or_sim_step_env(const OrModel *m, double *qpos, double *qvel, const double *ctrl, double *qacc_ws,
                    const double *envp, OrSimInfo *info) {
  return or_sim_step_full(m, qpos, qvel, ctrl, NULL, qacc_ws, envp, info);
}

/* motor_tau (may be NULL): closed-loop mode - joint torques in TSID joint order applied as motor forces
 * instead of the position servos */
int or_sim_step_full(const OrModel *m, double *qpos, double *qvel, const double *ctrl, const double *motor_tau,
                     double *qacc_ws, const double *envp, OrSimInfo *info) {
  return or_sim_step_ext(m, qpos, qvel, ctrl, motor_tau, qacc_ws, envp, NULL, 0, info);
}

/* mju_makeFrame: tangents of a contact frame from its normal (t1 from y unless |n_y| >= 0.5, t2 = n x t1) */
static void make_frame(const double *nrm, double *t1, double *t2) {
  double t[3] = {0, 0, 0};
  if (fabs(nrm[1]) < 0.5) t[1] = 1; else t[2] = 1;
  double dn = dot3(nrm, t), nn = 0;
  for (int i = 0; i < 3; i++) { t1[i] = t[i] - dn * nrm[i]; nn += t1[i] * t1[i]; }
  nn = 1.0 / sqrt(nn);
  for (int i = 0; i < 3; i++) t1[i] *= nn;
  cross(nrm, t1, t2);
}

int or_sim_step_ext(const OrModel *m, double *qpos, double *qvel, const double *ctrl, const double *motor_tau,
                    double *qacc_ws, const double *envp, const double *terr, int self_collision, OrSimInfo *info) {
  const double dt = m->opt[0], gz = m->opt[1], tol = m->opt[2];
  const int maxiter = (int)m->opt[3], ls_iter = (int)m->opt[4];
  const double ls_tol = m->opt[5];
  static __thread OrSimInfo local;
  if (!info) info = &local;
  memset(info, 0, sizeof *info);
  { /* non-finite state / targets, or a state that has diverged (sum |qpos| + |qvel| > 1e6: the reference's own
     * teleported sim gets there): the step is skipped */
    double chk = 0, big = 0;
    for (int i = 0; i < OR_NQ; i++) big += fabs(qpos[i]);
    for (int i = 0; i < NV; i++) { big += fabs(qvel[i]); chk += fabs(qacc_ws[i]); }
    for (int i = 0; i < OR_NA; i++) chk += fabs(ctrl[i]) + (motor_tau ? fabs(motor_tau[i]) : 0.0);
    if (!(chk <= 1e300) || !(big <= 1e6)) return 4;
  }

  /* ---------------- kinematics */
  double Rb[NB][9], pb[NB][3];
  quat_wxyz_to_R(qpos + 3, Rb[0]);
  memcpy(pb[0], qpos, sizeof pb[0]);
  for (int b = 1; b < NB; b++) {
    int p = m->mj_parent[b];
    double Rq[9], c = cos(qpos[6 + b]), s = sin(qpos[6 + b]);
    double rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    quat_wxyz_to_R(m->mj_quat[b], Rq);
    matmul3(Rb[p], Rq, Rb[b]);
    matmul3(Rb[b], rz, Rb[b]);
    matvec(Rb[p], m->mj_pos[b], pb[b]);
    for (int i = 0; i < 3; i++) pb[b][i] += pb[p][i];
  }
  /* motion subspaces (Pluecker at world origin) and dof -> body */
  double S[NV][6];
  int dof_body[NV];
  memset(S, 0, sizeof S);
  for (int k = 0; k < 3; k++) { S[k][k] = 1.0; dof_body[k] = 0; }
  for (int k = 0; k < 3; k++) {
    double a[3] = {Rb[0][k], Rb[0][3 + k], Rb[0][6 + k]};
    cross(pb[0], a, S[3 + k]);
    memcpy(S[3 + k] + 3, a, sizeof a);
    dof_body[3 + k] = 0;
  }
  for (int b = 1; b < NB; b++) {
    double a[3] = {Rb[b][2], Rb[b][5], Rb[b][8]};
    cross(pb[b], a, S[5 + b]);
    memcpy(S[5 + b] + 3, a, sizeof a);
    dof_body[5 + b] = b;
  }
  /* body inertias in world axes */
  WInertia Y[NB], Yc[NB];
  for (int b = 0; b < NB; b++) {
    const double *in = m->mj_inertia[b];
    const double ms = envp ? envp[0] : 1.0;
    double I[9] = {ms * in[4], ms * in[5], ms * in[6], ms * in[5], ms * in[7], ms * in[8], ms * in[6], ms * in[8], ms * in[9]}, T[9], RT[9];
    Y[b].m = ms * in[0];
    matvec(Rb[b], in + 1, Y[b].c);
    for (int i = 0; i < 3; i++) Y[b].c[i] += pb[b][i];
    matmul3(Rb[b], I, T);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RT[3 * i + j] = Rb[b][3 * j + i];
    matmul3(T, RT, Y[b].I);
    Yc[b] = Y[b];
  }
  for (int b = NB - 1; b > 0; b--) wi_add(&Yc[m->mj_parent[b]], &Yc[b]);

  /* ---------------- mass matrix (composite rigid body) + armature */
  double (*M)[NV] = info->M;
  for (int j = 0; j < NV; j++) {
    double F[6];
    wi_mul(&Yc[dof_body[j]], S[j], F);
    /* rows: dofs on the path from dof j's body to the root with index <= j */
    for (int b = dof_body[j]; b >= 0; b = m->mj_parent[b]) {
      int k0 = b == 0 ? 0 : 5 + b, k1 = b == 0 ? 5 : 5 + b;
      for (int k = k0; k <= k1; k++) {
        if (k > j) continue;
        double v = 0;
        for (int i = 0; i < 6; i++) v += S[k][i] * F[i];
        M[k][j] = v;
        M[j][k] = v;
      }
    }
    M[j][j] += m->mj_armature[j];
  }

  /* ---------------- bias forces: RNE with qacc = 0, gravity as base acceleration */
  double V[NB][6], Ab[NB][6], fb[NB][6];
  for (int b = 0; b < NB; b++) {
    int p = m->mj_parent[b];
    if (b == 0) {
      memset(V[0], 0, sizeof V[0]);
      for (int k = 0; k < 6; k++) for (int i = 0; i < 6; i++) V[0][i] += S[k][i] * qvel[k];
      double a0[6] = {0, 0, -gz, 0, 0, 0};
      memcpy(Ab[0], a0, sizeof a0);
      /* body-fixed angular axes: dS = V x S ; world-fixed linear axes: dS = 0 */
      for (int k = 3; k < 6; k++) {
        double dS[6];
        cross_mm(V[0], S[k], dS);
        for (int i = 0; i < 6; i++) Ab[0][i] += dS[i] * qvel[k];
      }
    } else {
      int k = 5 + b;
      for (int i = 0; i < 6; i++) V[b][i] = V[p][i] + S[k][i] * qvel[k];
      double dS[6];
      cross_mm(V[b], S[k], dS);
      for (int i = 0; i < 6; i++) Ab[b][i] = Ab[p][i] + dS[i] * qvel[k];
    }
    double Ya[6], Yv[6], vx[6];
    wi_mul(&Y[b], Ab[b], Ya);
    wi_mul(&Y[b], V[b], Yv);
    cross_mf(V[b], Yv, vx);
    for (int i = 0; i < 6; i++) fb[b][i] = Ya[i] + vx[i];
  }
  for (int b = NB - 1; b > 0; b--) for (int i = 0; i < 6; i++) fb[m->mj_parent[b]][i] += fb[b][i];
  for (int k = 0; k < NV; k++) {
    double v = 0;
    for (int i = 0; i < 6; i++) v += S[k][i] * fb[dof_body[k]][i];
    info->qfrc_bias[k] = v;
  }

  /* ---------------- actuation: position servo  kp (ctrl - q) - kv qdot */
  memset(info->qfrc_actuator, 0, sizeof info->qfrc_actuator);
  for (int a = 0; a < OR_NA; a++) {
    int d = m->mj_act_dof[a];
    if (motor_tau) info->qfrc_actuator[d] += motor_tau[m->mj_ctrl_qidx[a] - 7];
    else {
      /* ctrllimited: the control is clamped to ctrlrange; forcelimited: the actuator force to forcerange */
      const double *rg = m->act_range[a];
      const double c = fmin(fmax(ctrl[a], rg[0]), rg[1]);
      const double frc = m->mj_act_kp[a] * (c - qpos[d + 1]) - m->mj_act_kv[a] * qvel[d];
      info->qfrc_actuator[d] += fmin(fmax(frc, rg[2]), rg[3]);
    }
  }
  /* qfrc_smooth = passive (joint damping) + actuator - bias */
  double qfrc_smooth[NV];
  int any_damping = 0;
  for (int k = 0; k < NV; k++) {
    qfrc_smooth[k] = info->qfrc_actuator[k] - info->qfrc_bias[k] - m->mj_damping[k] * qvel[k];
    if (m->mj_damping[k] > 0) any_damping = 1;
  }
  double L[NV][NV];
  memcpy(L, M, sizeof L);
  if (chol(L, NV)) return -1;
  chol_solve(L, NV, qfrc_smooth, info->qacc_smooth);

  /* ---------------- collision: floor plane (z = 0, normal +z) vs every body's hull */
  /* floor plane n.x = d and its contact frame (mju_makeFrame: t1 from y unless |n_y| >= 0.5, t2 = n x t1) */
  double nrm[3] = {0, 0, 1}, pd = 0.0;
  if (envp) { nrm[0] = envp[2]; nrm[1] = envp[3]; nrm[2] = envp[4]; pd = envp[5]; }
  /* stepped terrain (BASELINE configs[4]; or_collide.c): the floor surface is raised along the normal by the
   * height of the cell under each point */
  double hmax = 0.0;
  if (terr) for (int i = 0; i < 16; i++) hmax = fmax(hmax, terr[4 + i]);
  const double margin = m->contact[10]; /* max of the two geoms' margins: the floor inherits the robot's (robot/v0/robot.xml:4,59) */
  int ncon = 0;
  for (int g = 0; g < OR_NG; g++) {
    const int b = m->geom_body[g];
    double cw[3];
    matvec(Rb[b], m->rbound[g], cw);
    for (int i = 0; i < 3; i++) cw[i] += pb[b][i];
    if (dot3(nrm, cw) - pd - m->rbound[g][3] - hmax > margin) continue;
    /* floor normal in the body frame, plane offset seen from the body origin */
    const double rn[3] = {nrm[0] * Rb[b][0] + nrm[1] * Rb[b][3] + nrm[2] * Rb[b][6],
                          nrm[0] * Rb[b][1] + nrm[1] * Rb[b][4] + nrm[2] * Rb[b][7],
                          nrm[0] * Rb[b][2] + nrm[1] * Rb[b][5] + nrm[2] * Rb[b][8]};
    const double pz = dot3(nrm, pb[b]) - pd;
    int v0 = m->hull_adr[g], v1 = m->hull_adr[g + 1];
    double zmin = INFINITY;
#define VERT_DIST(i, z)                                                                     \
  do {                                                                                      \
    const double *v_ = m->hull_vert + 3 * (i);                                              \
    (z) = rn[0] * v_[0] + rn[1] * v_[1] + rn[2] * v_[2] + pz;                                \
    if (terr) {                                                                             \
      double w_[3];                                                                         \
      matvec(Rb[b], v_, w_);                                                                \
      (z) -= or_terrain_height(terr, w_[0] + pb[b][0], w_[1] + pb[b][1]);                    \
    }                                                                                       \
  } while (0)
    for (int i = v0; i < v1; i++) {
      double z;
      VERT_DIST(i, z);
      if (z < zmin) zmin = z;
    }
    int best = -1;
    for (int i = v0; i < v1 && best < 0; i++) {
      double z;
      VERT_DIST(i, z);
      if (z <= zmin + TIE_TOL) best = i;
    }
    if (zmin > margin) continue;
    /* support vertex, then its hull-graph neighbours within the margin */
    int cand[64], nc = 0;
    cand[nc++] = best;
    for (int e = m->hull_eadr[best]; e < m->hull_eadr[best + 1] && nc < 64; e++) cand[nc++] = v0 + m->hull_edge[e];
    if (m->hull_eadr[best + 1] - m->hull_eadr[best] > 63) info->flags |= 16;
    /* plane <-> mesh multi-contact rule.  0: every neighbour within the margin.  1 (upstream's mjc_PlaneConvex, as far as
     * it is known here): in graph order, at most 3 more contacts, each at least 0.3 rbound away from the FIRST one */
    double first[3] = {0, 0, 0};
    int extra = 0;
    for (int c = 0; c < nc; c++) {
      const double *v = m->hull_vert + 3 * cand[c];
      double w[3];
      matvec(Rb[b], v, w);
      for (int i = 0; i < 3; i++) w[i] += pb[b][i];
      double dist = dot3(nrm, w) - pd - or_terrain_height(terr, w[0], w[1]);
      if (c > 0 && dist > margin) continue;
      double cp[3];
      for (int i = 0; i < 3; i++) cp[i] = w[i] - 0.5 * dist * nrm[i];
      if (m->plane_mesh) {
        if (c == 0) memcpy(first, cp, sizeof cp);
        else {
          const double d3[3] = {cp[0] - first[0], cp[1] - first[1], cp[2] - first[2]}, thr = 0.3 * m->rbound[g][3];
          if (extra >= 3 || dot3(d3, d3) < thr * thr) continue;
          extra++;
        }
      }
      if (ncon >= OR_MAXCON) { info->flags |= 8; break; }
      info->con_geom[ncon] = g;
      info->con_body2[ncon] = b;
      info->con_vert[ncon] = cand[c] - v0;
      info->con_body1[ncon] = -1;
      memcpy(info->con_frame[ncon], nrm, sizeof nrm);
      info->con_dist[ncon] = dist;
      for (int i = 0; i < 3; i++) info->con_pos[ncon][i] = w[i] - 0.5 * dist * nrm[i];
      ncon++;
    }
  }
#undef VERT_DIST
  /* ---------------- collision: robot<->robot convex-hull pairs (robot.xml:13-15,18-52; or_collide.c) */
  if (self_collision) {
    int g1[OR_MAXHH], g2[OR_MAXHH], over = 0;
    double hd[OR_MAXHH], hp[OR_MAXHH][3], hn[OR_MAXHH][3];
    const int nh = or_collide_pairs(m, Rb, pb, ncon, g1, g2, hd, hp, hn, &over);
    for (int k = 0; k < nh; k++, ncon++) {
      info->con_geom[ncon] = g2[k];
      info->con_body2[ncon] = m->geom_body[g2[k]];
      info->con_vert[ncon] = 0x8000 | g1[k];
      info->con_body1[ncon] = m->geom_body[g1[k]];
      memcpy(info->con_frame[ncon], hn[k], 24);
      info->con_dist[ncon] = hd[k];
      memcpy(info->con_pos[ncon], hp[k], 24);
    }
    if (over & 1) info->flags |= 8;
    if (over & 2) info->flags |= 32;
  }
  info->ncon = ncon;

  /* ---------------- constraint rows: frictionloss dofs, then pyramidal contact rows */
  static __thread Efc e;
  e.nefc = 0;
  const double mu_floor = envp ? envp[1] : m->contact[0];
  const double timeconst = m->contact[1] > 2 * dt ? m->contact[1] : 2 * dt, dampratio = m->contact[2];
  const double dmin = m->contact[3], dmax = m->contact[4], width = m->contact[5], mid = m->contact[6], power = m->contact[7];
  const double kk = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), bb = 2.0 / (dmax * timeconst);
  /* frictionloss rows use the joint's solreffriction / solimpfriction, which neither MJCF sets: MuJoCo's defaults
   * (0.02, 1) and the default solimp (the geoms' solref of robot/v0/robot.xml:4 does not apply to them) */
  const double tc_f = 0.02 > 2 * dt ? 0.02 : 2 * dt, bb_f = 2.0 / (dmax * tc_f);
  for (int k = 0; k < NV; k++) {
    if (m->mj_frictionloss[k] <= 0) continue;
    int r = e.nefc++;
    memset(e.J[r], 0, sizeof e.J[r]);
    e.J[r][k] = 1.0;
    double imp = dmin; /* pos = 0 */
    e.aref[r] = -bb_f * qvel[k];
    e.R[r] = fmax(MINVAL, (1 - imp) / imp * m->mj_dof_invw0[k]);
    e.D[r] = 1.0 / e.R[r];
    e.floss[r] = m->mj_frictionloss[k] * m->floss_scale;
    e.type[r] = 0;
  }
  for (int c = 0; c < ncon; c++) {
    int b = info->con_body2[c];
    const int b1 = info->con_body1[c];
    const double *r = info->con_pos[c], *nrm = info->con_frame[c];
    const double mu = b1 >= 0 ? m->contact[0] : mu_floor; /* robot geoms keep the model's friction */
    double t1[3], t2[3];
    make_frame(nrm, t1, t2);
    /* point Jacobian of geom2's body minus that of geom1's body (the floor does not move) */
    double Jp[3][NV], Jr[3][NV]; /* Jr: relative angular velocity of the two bodies per unit dof rate */
    memset(Jp, 0, sizeof Jp);
    memset(Jr, 0, sizeof Jr);
    for (int side = 0; side < 2; side++) {
      const double sg = side == 0 ? 1.0 : -1.0;
      for (int a = side == 0 ? b : b1; a >= 0; a = m->mj_parent[a]) {
        int k0 = a == 0 ? 0 : 5 + a, k1 = a == 0 ? 5 : 5 + a;
        for (int k = k0; k <= k1; k++) {
          double wxr[3];
          cross(S[k] + 3, r, wxr);
          for (int i = 0; i < 3; i++) { Jp[i][k] += sg * (S[k][i] + wxr[i]); Jr[i][k] += sg * S[k][3 + i]; }
        }
      }
    }
    /* friction directions of the pyramid: the two tangents (sliding, coefficient mu) and, with condim 4, the
     * rotation about the normal (torsional, coefficient contact[9]) */
    double Jn[NV], Jt[OR_CONDIM - 1][NV], muk[OR_CONDIM - 1];
    for (int k = 0; k < NV; k++) {
      Jn[k] = nrm[0] * Jp[0][k] + nrm[1] * Jp[1][k] + nrm[2] * Jp[2][k];
      Jt[0][k] = t1[0] * Jp[0][k] + t1[1] * Jp[1][k] + t1[2] * Jp[2][k];
      Jt[1][k] = t2[0] * Jp[0][k] + t2[1] * Jp[1][k] + t2[2] * Jp[2][k];
      if (OR_CONDIM > 3) Jt[OR_CONDIM - 2][k] = nrm[0] * Jr[0][k] + nrm[1] * Jr[1][k] + nrm[2] * Jr[2][k];
    }
    muk[0] = muk[1] = mu;
    if (OR_CONDIM > 3) muk[OR_CONDIM - 2] = m->contact[9];
    double dist = info->con_dist[c];
    /* impedance from penetration */
    double x = fabs(dist - margin) / width, imp;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      double y;
      if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = dmin + y * (dmax - dmin);
    }
    double tran = m->mj_body_invw0[b][0] + (b1 >= 0 ? m->mj_body_invw0[b1][0] : 0.0); /* world body contributes 0 */
    double diagA = tran + mu * mu * tran;
    double R0 = fmax(MINVAL, (1 - imp) / imp * diagA);
    double Rpy = 2 * mu * mu * R0;
    /* (every row of the pyramid gets Rpy, derived from the first row's diagonal approximation: mj_makeImpedance) */
    for (int tdir = 0; tdir < OR_CONDIM - 1; tdir++)
      for (int sg = 0; sg < 2; sg++) {
        int rI = e.nefc++;
        double vel = 0;
        for (int k = 0; k < NV; k++) {
          e.J[rI][k] = Jn[k] + (sg ? -muk[tdir] : muk[tdir]) * Jt[tdir][k];
          vel += e.J[rI][k] * qvel[k];
        }
        e.aref[rI] = -bb * vel - kk * imp * (dist - margin);
        e.R[rI] = Rpy;
        e.D[rI] = 1.0 / Rpy;
        e.floss[rI] = 0;
        e.type[rI] = 1;
      }
  }
  e.ncon = ncon;
  info->nefc = e.nefc;

  /* ---------------- constraint solve (Newton on the primal problem in qacc) */
  double qacc[NV];
  const int nefc = e.nefc;
  if (nefc == 0) {
    memcpy(qacc, info->qacc_smooth, sizeof qacc);
  } else {
    double jar[OR_MAXEFC], force[OR_MAXEFC], Ma[NV], grad[NV], search[NV], Mv[NV], Jv[OR_MAXEFC];
    int active[OR_MAXEFC];
#define MULM(x, out) for (int i_ = 0; i_ < NV; i_++) { double s_ = 0; for (int j_ = 0; j_ < NV; j_++) s_ += M[i_][j_] * (x)[j_]; (out)[i_] = s_; }
#define MULJ(x, out) for (int i_ = 0; i_ < nefc; i_++) { double s_ = 0; for (int j_ = 0; j_ < NV; j_++) s_ += e.J[i_][j_] * (x)[j_]; (out)[i_] = s_; }
    /* warm start: keep qacc_warmstart only if it is cheaper than qacc_smooth */
    memcpy(qacc, qacc_ws, sizeof qacc);
    MULM(qacc, Ma);
    MULJ(qacc, jar);
    for (int i = 0; i < nefc; i++) jar[i] -= e.aref[i];
    double cost_w = efc_update(&e, jar, force, active);
    for (int k = 0; k < NV; k++) cost_w += 0.5 * (Ma[k] - qfrc_smooth[k]) * (qacc[k] - info->qacc_smooth[k]);
    double jar_s[OR_MAXEFC];
    MULJ(info->qacc_smooth, jar_s);
    for (int i = 0; i < nefc; i++) jar_s[i] -= e.aref[i];
    double cost_s = efc_update(&e, jar_s, force, active);
    if (cost_w > cost_s) memcpy(qacc, info->qacc_smooth, sizeof qacc);

    const double scale = 1.0 / (m->meaninertia * NV);
    MULM(qacc, Ma);
    MULJ(qacc, jar);
    for (int i = 0; i < nefc; i++) jar[i] -= e.aref[i];
    double cost = 0;
    int iter = 0;
    /* Newton Hessian H = M + J^T D J over the rows in their quadratic zone.  Built and factored in full on the first
     * iteration; afterwards only the rows whose state changed are applied to the FACTOR as rank-1 updates / downdates
     * with sqrt(D_i) J_i (what MuJoCo's Newton solver does for pyramidal cones: HessianIncremental / mju_cholUpdate),
     * unless more than OR_NEWTON_INCR_MAX rows changed or a downdate loses definiteness - then it is rebuilt. */
    double Hf[NV][NV];
    int have_fac = 0, act_prev[OR_MAXEFC];
    for (;;) {
      /* update: constraint state, cost, gradient, Newton direction */
      double ccost = efc_update(&e, jar, force, active), gauss = 0;
      for (int k = 0; k < NV; k++) gauss += 0.5 * (Ma[k] - qfrc_smooth[k]) * (qacc[k] - info->qacc_smooth[k]);
      double newcost = gauss + ccost;
      for (int k = 0; k < NV; k++) {
        double s = 0;
        for (int i = 0; i < nefc; i++) s += e.J[i][k] * force[i];
        grad[k] = Ma[k] - qfrc_smooth[k] - s;
      }
      if (iter > 0) {
        double gn = 0;
        for (int k = 0; k < NV; k++) gn += grad[k] * grad[k];
        double improvement = scale * (cost - newcost), gradient = scale * sqrt(gn);
        cost = newcost;
        if (improvement < tol || gradient < tol) break;
      }
      cost = newcost;
      if (iter >= maxiter) break;
      int need_full = !have_fac;
      if (have_fac) {
        int nchange = 0;
        for (int i = 0; i < nefc; i++) nchange += active[i] != act_prev[i];
        __atomic_fetch_add(&or_newton_hist[nchange], 1, __ATOMIC_RELAXED);
        if (nchange > or_newton_incr_max) need_full = 1;
        for (int i = 0; i < nefc && !need_full; i++) {
          if (active[i] == act_prev[i]) continue;
          double vec[NV];
          const double sd = sqrt(e.D[i]);
          for (int k = 0; k < NV; k++) vec[k] = sd * e.J[i][k];
          if (chol_rank1(Hf, NV, vec, active[i] ? 1.0 : -1.0)) need_full = 1;
          else info->newton_rank1++;
        }
      }
      if (need_full) {
        memcpy(Hf, M, sizeof Hf);
        for (int i = 0; i < nefc; i++) {
          if (!active[i]) continue;
          for (int a = 0; a < NV; a++) {
            if (e.J[i][a] == 0) continue;
            double da = e.D[i] * e.J[i][a];
            for (int b2 = 0; b2 < NV; b2++) Hf[a][b2] += da * e.J[i][b2];
          }
        }
        if (chol(Hf, NV)) return -2;
        info->newton_full++;
      }
      have_fac = 1;
      memcpy(act_prev, active, sizeof(int) * nefc);
      chol_solve(Hf, NV, grad, search);
      for (int k = 0; k < NV; k++) search[k] = -search[k];

      /* exact line search */
      MULM(search, Mv);
      MULJ(search, Jv);
      double qg[3] = {gauss, 0, 0}, snorm = 0;
      for (int k = 0; k < NV; k++) {
        qg[1] += search[k] * (Ma[k] - qfrc_smooth[k]);
        qg[2] += 0.5 * search[k] * Mv[k];
        snorm += search[k] * search[k];
      }
      snorm = sqrt(snorm);
      if (snorm < MINVAL) break;
      double gtol = tol * ls_tol * snorm * m->meaninertia * NV;
      double c0, d1, d2, alpha = 0, lo = 0, hi = INFINITY, ca, g1, g2;
      ls_eval(&e, jar, Jv, qg, 0.0, &c0, &d1, &d2);
      ca = c0; g1 = d1; g2 = d2;
      for (int li = 0; li < ls_iter && fabs(g1) >= gtol; li++) {
        if (g1 < 0) lo = alpha; else hi = alpha;
        double an = alpha - g1 / g2;
        if (!(an > lo) || !(an < hi)) an = isinf(hi) ? 2 * alpha + 1 : 0.5 * (lo + hi);
        alpha = an;
        ls_eval(&e, jar, Jv, qg, alpha, &ca, &g1, &g2);
      }
      if (!(ca < c0) || alpha == 0.0) break; /* no improvement */
      for (int k = 0; k < NV; k++) { qacc[k] += alpha * search[k]; Ma[k] += alpha * Mv[k]; }
      for (int i = 0; i < nefc; i++) jar[i] += alpha * Jv[i];
      iter++;
    }
    info->solver_iter = iter;
    efc_update(&e, jar, force, active);
    memcpy(info->efc_force, force, sizeof(double) * nefc);
#undef MULM
#undef MULJ
  }
  memcpy(info->qacc, qacc, sizeof qacc);
  memcpy(qacc_ws, qacc, sizeof qacc);

  /* ---------------- semi-implicit Euler; with joint damping mj_Euler integrates it implicitly:
   * (M + h B) qacc_int = qfrc_smooth + qfrc_constraint = M qacc   (the warm start keeps the plain qacc) */
  if (any_damping) {
    double Mq[NV], A[NV][NV];
    for (int i = 0; i < NV; i++) {
      double s_ = 0;
      for (int j = 0; j < NV; j++) s_ += M[i][j] * qacc[j];
      Mq[i] = s_;
    }
    memcpy(A, M, sizeof A);
    for (int i = 0; i < NV; i++) A[i][i] += dt * m->mj_damping[i];
    if (chol(A, NV)) return -3;
    chol_solve(A, NV, Mq, qacc);
  }
  for (int k = 0; k < NV; k++) qvel[k] += dt * qacc[k];
  for (int i = 0; i < 3; i++) qpos[i] += dt * qvel[i];
  {
    const double *w = qvel + 3;
    double th = sqrt(dot3(w, w)) * dt, dq[4] = {1, 0, 0, 0};
    if (th > 0) {
      double s = sin(0.5 * th) * dt / th;
      dq[0] = cos(0.5 * th); dq[1] = s * w[0]; dq[2] = s * w[1]; dq[3] = s * w[2];
    }
    double *a = qpos + 3, r[4];
    r[0] = a[0] * dq[0] - a[1] * dq[1] - a[2] * dq[2] - a[3] * dq[3];
    r[1] = a[0] * dq[1] + a[1] * dq[0] + a[2] * dq[3] - a[3] * dq[2];
    r[2] = a[0] * dq[2] - a[1] * dq[3] + a[2] * dq[0] + a[3] * dq[1];
    r[3] = a[0] * dq[3] + a[1] * dq[2] - a[2] * dq[1] + a[3] * dq[0];
    double n = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int i = 0; i < 4; i++) a[i] = r[i] * n;
  }
  for (int k = 6; k < NV; k++) qpos[k + 1] += dt * qvel[k];
  return 0;
}

int or_sim_step(const OrModel *m, double *qpos, double *qvel, const double *ctrl, double *qacc_ws,
                OrSimInfo *info) {
  return or_sim_step_env(m, qpos, qvel, ctrl, qacc_ws, NULL, info);
}
