/* oracle.h - CPU float64 restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (tsid_control_amd/, libtsidb.so) never links, imports or calls it.
 *
 * PARITY UNPINNED for the native stages: the reference's arithmetic lives in tsid, pinocchio,
 * eiquadprog and mujoco (none vendored, pinned or installed) and the reference holds
 * no tests or golden vectors.  Each function below restates the published algorithm of the library
 * the cited reference call site reaches, and is pinned only by the invariants in tests/
 * (CRBA == RNEA columns, KKT residuals, Newton-Euler balance, closed-form sim cases).
 */
#ifndef TSIDB_ORACLE_H
#define TSIDB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One library per robot, like libtsidb: liboracle.so = the v1 robot (ctrl/conf.py:9-15, robot/v1/mujoco/robot.xml),
 * liboracle_v0.so (-DOR_ROBOT_V0) = robot/v0 (robot.urdf + robot.xml: several collision meshes per body, condim 4,
 * joint damping, geom margin, actuator ranges).  OR_NG = collision geoms, OR_CONDIM = contact dimension (3: sliding
 * friction; 4: + torsional friction about the normal). */
#ifdef OR_ROBOT_V0
enum { OR_NJ = 19, OR_NQ = 25, OR_NV = 24, OR_NA = 18, OR_NB = 19, OR_NF = 2, OR_HAS_SIM = 1, OR_NG = 52, OR_CONDIM = 4,
       OR_MAXPAIR = 1100 };
#else
enum { OR_NJ = 21, OR_NQ = 27, OR_NV = 26, OR_NA = 20, OR_NB = 21, OR_NF = 2, OR_HAS_SIM = 1, OR_NG = 21, OR_CONDIM = 3,
       OR_MAXPAIR = 256 };
#endif
enum { OR_NVAR = OR_NV + 24, OR_NEQ = 18, OR_NIN = 68 + 2 * OR_NA + 2 * OR_NV };
enum { OR_NROWC = 2 * (OR_CONDIM - 1) /* pyramid rows per contact */ };
enum { OR_NEWTON_INCR_MAX = 8 /* Newton: at most this many changed rows are applied to the factor as rank-1 updates;
                                 more, and the Hessian is rebuilt and factored (same constant in csrc/tsidb_sim.hpp) */ };
extern int or_newton_incr_max;  /* run-time copy (diagnostics may change it) */
enum { OR_MAXCON = 32, OR_MAXHH = 12 /* robot<->robot contacts per env */, OR_MAXEFC = OR_NA + OR_NROWC * OR_MAXCON,
       OR_NOBS = OR_NQ + OR_NV + 12 };
int or_dims(int *out6); /* NJ, NQ, NV, NA, sim bodies, has_sim of this build */

/* parameter vector indices (RobotConfig values; ctrl/conf.py:21-72) */
enum {
  P_DT = 0, P_MU, P_FMIN, P_FMAX, P_W_FORCEREF, P_KP_CONTACT, P_KD_CONTACT,
  P_W_FOOT, P_KP_FOOT, P_KD_FOOT, P_W_COM, P_KP_COM, P_KD_COM, P_W_POSTURE,
  P_HESS_REG, P_QUIRKS, P_NORMAL /*3*/, P_CPOINTS = P_NORMAL + 3 /*12*/,
  P_KP_POSTURE = P_CPOINTS + 12 /*20*/, P_KD_POSTURE = P_KP_POSTURE + 20,
  P_TAU_MAX = P_KD_POSTURE + 20, P_V_MAX = P_TAU_MAX + 20, P_MAX_ITER = P_V_MAX + 20,
  P_SIM_ENABLED, P_CLOSED_LOOP, P_W_AM, P_KP_AM /*3*/, P_REW_SIGMA = P_KP_AM + 3, P_REW_CTAU, P_DONE_HEIGHT, P_DONE_TILT,
  P_SELF_COLLISION, P_W_COP, P_SIM_FLOSS_SCALE, P_TSID_ARMATURE, P_FRICTION_COMP, P_PLANE_MESH, P_COUNT = 128
};

typedef struct {
  /* TSID side (pinocchio conventions) */
  int pin_parent[OR_NJ];
  double pin_place[OR_NJ][12];   /* R row-major (9), p (3): joint frame in parent joint frame */
  double pin_inertia[OR_NJ][10]; /* mass, com(3), Ixx Ixy Ixz Iyy Iyz Izz about com */
  int frame_parent[OR_NF];
  double frame_place[OR_NF][12];
  double effort[OR_NA], velocity[OR_NA], q0[OR_NQ];
  /* sim side (MuJoCo conventions) */
  int mj_parent[OR_NB];
  double mj_pos[OR_NB][3], mj_quat[OR_NB][4], mj_inertia[OR_NB][10];
  double mj_armature[OR_NV], mj_frictionloss[OR_NV], mj_dof_M0[OR_NV], mj_dof_invw0[OR_NV];
  double mj_body_invw0[OR_NB][2];
  int mj_act_dof[OR_NA];
  double mj_act_kp[OR_NA], mj_act_kv[OR_NA];
  int mj_ctrl_qidx[OR_NA];
  double mj_damping[OR_NV];    /* joint damping (robot/v0/robot.xml:3; 0 for the v1 robot) */
  double act_range[OR_NA][4];  /* ctrl lo, hi, force lo, hi (robot/v0/robot.xml:5; +-1e300 = unlimited) */
  /* collision geoms: convex hulls of the collision meshes, each carried by a body, vertices in that body's frame */
  int geom_body[OR_NG];
  int hull_adr[OR_NG + 1];
  int nhullvert, nhulledge;
  const double *hull_vert; /* [nhullvert][3], body frame */
  const int *hull_eadr;    /* [nhullvert+1] */
  const int *hull_edge;    /* neighbour ids local to the geom's hull */
  double rbound[OR_NG][4];
  double hull_center[OR_NG][3]; /* centre of mass of the solid hull, body frame */
  double hull_box[OR_NG][6];    /* body-frame bounding box: centre, half extents */
  int npair, pairs[2 * OR_MAXPAIR]; /* robot<->robot candidate geom pairs (excludes + parent-child + same-body filtered) */
  double opt[7];      /* dt gz tol iters ls_iters ls_tol impratio */
  double contact[12]; /* mu solref[2] solimp[5] condim mu_torsion margin spare */
  double meaninertia;
  int foot_body[2];   /* sim bodies carrying the left / right sole frame */
  uint64_t foot_geoms[2]; /* bit g set: geom g is on that body */
  double floss_scale; /* closed-loop knob params[P_SIM_FLOSS_SCALE] (1 = robot.xml:8), set by the batch entry points */
  int plane_mesh;     /* params[P_PLANE_MESH]: 0 = every hull-graph neighbour of the support vertex within the margin is a contact;
                       * 1 = upstream's plane <-> mesh rule (at most 3 more, each at least 0.3 rbound from the first) */
  void *owned;
} OrModel;

typedef struct {
  double M[OR_NV][OR_NV], h[OR_NV];
  double com[3], vcom[3], acom[3], Jcom[3][OR_NV];
  double oMf[OR_NF][12];          /* frame placement R(9) p(3) */
  double Jf[OR_NF][6][OR_NV];     /* frame Jacobian, LOCAL */
  double vf[OR_NF][6], af[OR_NF][6]; /* frame velocity, classical drift acceleration (LOCAL) */
  double mass;
  /* centroidal angular momentum (tsid::TaskAMEquality, legacy/biped.py:82-87): A_G angular rows, L = A v,
   * drift = dA/dt v (what computeCentroidalMomentumTimeVariation leaves at zero joint acceleration) */
  double Aam[3][OR_NV], Lam[3], dLam[3];
} OrTerms;

typedef struct {
  int nvar, neq, nin; /* nin = one-sided rows */
  double H[OR_NVAR][OR_NVAR], g[OR_NVAR];
  double CE[OR_NEQ][OR_NVAR], ce0[OR_NEQ];
  double CI[OR_NIN][OR_NVAR], ci0[OR_NIN];
  int slot_foot[OR_NF]; /* force slot -> foot id, -1 if unused */
} OrQP;

typedef struct {
  double x[OR_NVAR], u[OR_NEQ + OR_NIN];
  int A[OR_NEQ + OR_NIN], iq, iter, status;
  double f_value;
} OrQPSol;

OrModel *or_model_load(const void *blob, size_t nbytes);
void or_model_free(OrModel *m);

/* pinocchio-side terms: what tsid::RobotWrapper::computeAllTerms leaves in Data (main.py:119) */
void or_rbd_terms(const OrModel *m, const double *q, const double *v, OrTerms *t);
void or_rnea(const OrModel *m, const double *q, const double *v, const double *a, double *tau);
void or_integrate(const double *q, const double *vdt, double *qout); /* pin.integrate, WalkController.py:294 */
void or_log6(const double *Mrel12, double *out6);

/* TSID problem (formulation.computeProblemData + SolverHQuadProgFast data copy; main.py:119-121) */
void or_tsid_assemble(const OrModel *m, const double *params, const OrTerms *t, const double *q,
                      const double *v, const double *com_ref, const double *posture_ref,
                      const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                      OrQP *qp);
void or_tsid_assemble_cop(const OrModel *m, const double *params, const OrTerms *t, const double *q,
                          const double *v, const double *com_ref, const double *posture_ref,
                          const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                          const double *cop_ref /* [3] or NULL: CoP task reference (params[P_W_COP]) */, OrQP *qp);
int or_qp_solve(const OrQP *qp, int max_iter, OrQPSol *sol); /* eiquadprog-fast restatement */

/* one TSID tick for one env: main.py:119-129,132-142 */
int or_tsid_tick(const OrModel *m, const double *params, double *q, double *v, const double *com_ref,
                 const double *posture_ref, const double *foot_ref, const double *contact_ref,
                 const uint8_t *contact_active, const double *cop_frames /*2x12, quirk (e)*/,
                 double *tau, double *dv, double *f, double *obs, int *iters);

int or_tsid_tick_cop(const OrModel *m, const double *params, double *q, double *v, const double *com_ref,
                     const double *posture_ref, const double *foot_ref, const double *contact_ref,
                     const uint8_t *contact_active, const double *cop_frames, const double *cop_ref, double *tau,
                     double *dv, double *f, double *obs, int *iters);

void or_model_set_plane_mesh(OrModel *m, int rule); /* OrModel.plane_mesh for the single-env entry points */

/* MuJoCo-subset step for one env: main.py:195 */
typedef struct {
  int ncon, nefc, solver_iter;
  int con_geom[OR_MAXCON]; /* geom2 (a mesh); geom1 is the floor plane unless con_body1 >= 0 */
  int con_vert[OR_MAXCON]; /* floor contacts: hull vertex; robot<->robot contacts: 0x8000 | geom1 */
  double con_dist[OR_MAXCON], con_pos[OR_MAXCON][3];
  double efc_force[OR_MAXEFC];
  double qacc[OR_NV], qacc_smooth[OR_NV], qfrc_bias[OR_NV], qfrc_actuator[OR_NV], M[OR_NV][OR_NV];
  int con_body1[OR_MAXCON];      /* body of geom1; -1 = floor */
  double con_frame[OR_MAXCON][3]; /* contact normal (geom1 -> geom2), world */
  int flags;                      /* 8: a contact was dropped at a cap (OR_MAXCON, OR_MAXHH); 16: a support vertex has more than 63
                                   * hull-graph neighbours (the rest is not looked at); 32: more than 64 pairs survived the mid phase */
  int con_body2[OR_MAXCON];      /* body of geom2 */
  int newton_full, newton_rank1; /* Newton Hessian factorisations from scratch / rows applied as rank-1 updates */
} OrSimInfo;
int or_sim_step(const OrModel *m, double *qpos, double *qvel, const double *ctrl, double *qacc_ws,
                OrSimInfo *info);
/* same with the per-env randomisation of BASELINE config 5: envp = mass scale, friction, floor normal (3), offset */
int or_sim_step_env(const OrModel *m, double *qpos, double *qvel, const double *ctrl, double *qacc_ws,
                    const double *envp, OrSimInfo *info);
int or_sim_step_full(const OrModel *m, double *qpos, double *qvel, const double *ctrl, const double *motor_tau,
                     double *qacc_ws, const double *envp, OrSimInfo *info);
/* full form: terr = stepped-terrain table (20 doubles, or_collide.c) or NULL; self_collision != 0 collides the
 * robot<->robot hull pairs as mj_step does (robot.xml:13-15,18-52) */
int or_sim_step_ext(const OrModel *m, double *qpos, double *qvel, const double *ctrl, const double *motor_tau,
                    double *qacc_ws, const double *envp, const double *terr, int self_collision, OrSimInfo *info);
double or_terrain_height(const double *terr, double X, double Y);
/* a, b: geoms, each placed by the given rotation / position of the body that carries it */
int or_mpr_penetration(const OrModel *m, int a, const double *Ra, const double *pa, int b, const double *Rb,
                       const double *pb, double *depth, double *dir_out, double *pos);
int or_mpr_penetration_margin(const OrModel *m, int a, const double *Ra, const double *pa, int b, const double *Rb,
                              const double *pb, double margin, double *depth, double *dir_out, double *pos);
int or_collide_pairs(const OrModel *m, const double Rb[][9], const double pb[][3], int ncon0, int *geom1, int *geom2,
                     double *dist, double (*pos)[3], double (*nrm)[3], int *overflow);

/* whole env step (tick + base teleport + ctrl map + sim step): main.py:119-129,192-195 */
int or_env_step_batch(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                      double *qvel, double *qacc_ws, const double *com_ref, const double *posture_ref,
                      const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                      const double *cop_frames, double *tau, double *dv, double *f, int32_t *status,
                      double *obs, int32_t *ncon, int32_t *con_geom, int nthreads);
int or_env_step_batch_env(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                          double *qvel, double *qacc_ws, const double *com_ref, const double *posture_ref,
                          const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                          const double *cop_frames, const double *env_params /* [n][8] or NULL */, double *tau,
                          double *dv, double *f, int32_t *status, double *obs, int32_t *ncon, int32_t *con_geom,
                          int nthreads);

/* walking reference update for one tick (or_walk.c): the device kernel behind tsidb_walk_update, restated from
 * ctrl/Foot_Trajectory.py:21-27, ctrl/Walk_Planner.py:23-31, ctrl/WalkController.py:189-253, ctrl/LIPM.py:34-49 */
void or_walk_update(int n, const double *coef, const int32_t *side, const int32_t *nsteps, const double *rest,
                    const double *com, int K, double t, const double *t_off, double T, double t_start, double omega,
                    double z0, double dz, const double *frames, double *foot_ref, double *contact_ref,
                    uint8_t *contact_active, double *com_ref);

void or_walk_update_fb(int n, const double *coef, const int32_t *side, const int32_t *nsteps, const double *rest,
                       const double *com, int K, double t, const double *t_off, double T, double t_start, double omega,
                       double z0, double dz, const double *frames, double *foot_ref, double *contact_ref,
                       uint8_t *contact_active, double *com_ref, const int32_t *ncon, const int32_t *con_geom,
                       int32_t *latch, uint64_t fgeoms0, uint64_t fgeoms1 /* geoms of the left / right foot body */,
                       double td_frac);

/* episode plan (or_walk.c): footsteps, swing polynomials, rest placements and the LIPM / DCM CoM plan per env - the twin of
 * the device kernel behind tsidb_walk_plan; layouts in or_walk.c */
void or_walk_plan(int n, const double *pp, const double *cop_frames, const double *com_ref, const double *path,
                  const int32_t *npts, int P, const double *scale, const int32_t *episode, int K, double *steps_out,
                  double *coef, int32_t *side, int32_t *nsteps, double *rest, double *com, int32_t *flags);
uint64_t or_plan_hash(uint64_t seed, uint64_t env, uint64_t episode);

/* walking tables for or_env_step_batch_walk (env-major, layouts as or_walk_update) */
typedef struct {
  const double *coef, *rest, *com, *t_off;
  const int32_t *side, *nsteps;
  int K;
  double t, T, t_start, omega, z0, dz;
  int32_t *td_latch; /* [n] contact-timing feedback state or NULL (or_walk_update_fb) */
  double td_frac;
} OrWalkTables;
int or_env_step_batch_walk(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                           double *qvel, double *qacc_ws, double *com_ref, const double *posture_ref,
                           double *foot_ref, double *contact_ref, uint8_t *contact_active,
                           const double *cop_frames, const double *env_params, double *tau, double *dv, double *f,
                           int32_t *status, double *obs, int32_t *ncon, int32_t *con_geom, int nthreads,
                           const OrWalkTables *w /* NULL: references as given */, double *frames /* [n,2,12] or NULL */,
                           double *rewdone /* [n,2] reward, done; or NULL */,
                           const double *terrain /* [n,20] stepped-terrain tables (or_collide.c) or NULL */,
                           const double *cop_ref /* [n,3] CoP task reference or NULL */);

#ifdef __cplusplus
}
#endif
#endif
