/* or_api.c - whole env step over a batch (TEST INFRASTRUCTURE; see oracle.h).
 * One env step = main.py:119-129 (TSID tick) then main.py:192-195 (base teleport, ctrl map,
 * mj_step).  OpenMP over envs is used only by bench.py's cpu_baseline leg. */
#include "oracle.h"
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* closed loop: TSID state from the sim state.  qpos = p, quat wxyz, joints in sim
 * order; qvel = world-frame linear velocity, body-frame angular velocity, joint rates.  TSID wants quat
 * xyzw, body-frame linear velocity and its own joint order. */
static void sim_to_tsid(const OrModel *m, const double *qpos, const double *qvel, double *q, double *v) {
  for (int i = 0; i < 3; i++) q[i] = qpos[i];
  q[3] = qpos[4]; q[4] = qpos[5]; q[5] = qpos[6]; q[6] = qpos[3];
  double w = qpos[3], x = qpos[4], y = qpos[5], z = qpos[6], nn = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= nn; x *= nn; y *= nn; z *= nn;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                       2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                       2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  for (int i = 0; i < 3; i++) v[i] = R[i] * qvel[0] + R[3 + i] * qvel[1] + R[6 + i] * qvel[2];
  for (int i = 3; i < 6; i++) v[i] = qvel[i];
  for (int a = 0; a < OR_NA; a++) {
    q[m->mj_ctrl_qidx[a]] = qpos[7 + a];
    v[m->mj_ctrl_qidx[a] - 1] = qvel[6 + a];
  }
}

extern __thread double or_last_frames[24], or_last_rowx[2];

int or_env_step_batch_env(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                          double *qvel, double *qacc_ws, const double *com_ref, const double *posture_ref,
                          const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                          const double *cop_frames, const double *env_params, double *tau, double *dv, double *f,
                          int32_t *status, double *obs, int32_t *ncon, int32_t *con_geom, int nthreads) {
  return or_env_step_batch_walk(m, params, n, q, v, qpos, qvel, qacc_ws, (double *)com_ref, posture_ref, (double *)foot_ref,
                                (double *)contact_ref, (uint8_t *)contact_active, cop_frames, env_params, tau, dv, f, status, obs,
                                ncon, con_geom, nthreads, NULL, NULL, NULL, NULL, NULL);
}

/* the same env step preceded, per env, by the walking reference update of or_walk.c when `w` is given (the
 * config-3 workload: what bench.py's GPU loop does with tsidb_walk_update + tsidb_tick + tsidb_sim); `frames`
 * [n,2,12] carries the sole placements from one tick to the next update, as on the device */
int or_env_step_batch_walk(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                           double *qvel, double *qacc_ws, double *com_ref, const double *posture_ref,
                           double *foot_ref, double *contact_ref, uint8_t *contact_active,
                           const double *cop_frames, const double *env_params, double *tau, double *dv, double *f,
                           int32_t *status, double *obs, int32_t *ncon, int32_t *con_geom, int nthreads,
                           const OrWalkTables *w, double *frames, double *rewdone, const double *terrain, const double *cop_ref) {
  ((OrModel *)m)->floss_scale = params[P_SIM_FLOSS_SCALE] != 0.0 ? params[P_SIM_FLOSS_SCALE] : 1.0;
  ((OrModel *)m)->plane_mesh = params[P_PLANE_MESH] != 0.0;
  const int sim = params[P_SIM_ENABLED] != 0.0;
  const int closed = params[P_CLOSED_LOOP] != 0.0;
  const int quirks = params[P_QUIRKS] != 0.0 && !closed;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int e = 0; e < n; e++) {
    double *qe = q + (size_t)e * OR_NQ, *ve = v + (size_t)e * OR_NV;
    if (w)
      or_walk_update_fb(1, w->coef + (size_t)e * w->K * 16, w->side + (size_t)e * w->K, w->nsteps + e,
                        w->rest + (size_t)e * (w->K + 1) * 8, w->com + (size_t)e * (w->K + 2) * 6, w->K, w->t,
                        w->t_off ? w->t_off + e : NULL, w->T, w->t_start, w->omega, w->z0, w->dz, frames + (size_t)e * 24,
                        foot_ref + (size_t)e * 48, contact_ref + (size_t)e * 24, contact_active + (size_t)e * 2,
                        com_ref + (size_t)e * 9, w->td_latch ? ncon + e : NULL,
                        w->td_latch ? con_geom + (size_t)e * OR_MAXCON : NULL, w->td_latch ? w->td_latch + e : NULL,
                        m->foot_geoms[0], m->foot_geoms[1], w->td_frac);
    if (closed) sim_to_tsid(m, qpos + (size_t)e * OR_NQ, qvel + (size_t)e * OR_NV, qe, ve);
    int st = or_tsid_tick_cop(m, params, qe, ve, com_ref + (size_t)e * 9, posture_ref + (size_t)e * OR_NA,
                              foot_ref + (size_t)e * 48, contact_ref + (size_t)e * 24, contact_active + (size_t)e * 2,
                              cop_frames ? cop_frames + (size_t)e * 24 : NULL, cop_ref ? cop_ref + (size_t)e * 3 : NULL,
                              tau + (size_t)e * OR_NA, dv + (size_t)e * OR_NV, f + (size_t)e * 24,
                              obs ? obs + (size_t)e * OR_NOBS : NULL, NULL);
    status[e] = st;
    if (frames && st != 4) memcpy(frames + (size_t)e * 24, or_last_frames, sizeof or_last_frames);
    if (rewdone) { rewdone[2 * (size_t)e] = or_last_rowx[0]; rewdone[2 * (size_t)e + 1] = or_last_rowx[1]; }
    if (!sim) continue;
    double *qp = qpos + (size_t)e * OR_NQ, *qv = qvel + (size_t)e * OR_NV;
    double ctrl[OR_NA] = {0};
    if (!closed) {
      /* main.py:192  mj_data.qpos[:7] = q[:7]  (quirk F6a: xyzw copied into the wxyz slot) */
      for (int i = 0; i < 3; i++) qp[i] = qe[i];
      if (quirks) for (int i = 0; i < 4; i++) qp[3 + i] = qe[3 + i];
      else {
        qp[3] = qe[6]; qp[4] = qe[3]; qp[5] = qe[4]; qp[6] = qe[5];
        /* without the quirks the base is teleported with its velocity (the reference leaves qvel alone):
         * TSID linear velocity is in the body frame, the sim's in the world frame */
        const double x = qe[3], y = qe[4], z = qe[5], w = qe[6];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                             2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
        for (int i = 0; i < 3; i++) qv[i] = R[3 * i] * ve[0] + R[3 * i + 1] * ve[1] + R[3 * i + 2] * ve[2];
        for (int i = 3; i < 6; i++) qv[i] = ve[i];
      }
      /* main.py:193-194  ctrl = map_tsid_to_mujoco(q) */
      for (int a = 0; a < OR_NA; a++) ctrl[a] = qe[m->mj_ctrl_qidx[a]];
    }
    OrSimInfo info;
    int rc = or_sim_step_ext(m, qp, qv, ctrl, closed ? tau + (size_t)e * OR_NA : NULL, qacc_ws + (size_t)e * OR_NV,
                             env_params ? env_params + (size_t)e * 8 : NULL, terrain ? terrain + (size_t)e * 20 : NULL,
                             params[P_SELF_COLLISION] != 0.0, &info);
    if (rc) status[e] |= 0x100;
    if (ncon) ncon[e] = info.ncon;
    if (con_geom) {
      for (int c = 0; c < OR_MAXCON; c++)
        con_geom[(size_t)e * OR_MAXCON + c] = c < info.ncon ? (info.con_geom[c] << 16 | info.con_vert[c]) : -1;
    }
  }
  return 0;
}

int or_env_step_batch(const OrModel *m, const double *params, int n, double *q, double *v, double *qpos,
                      double *qvel, double *qacc_ws, const double *com_ref, const double *posture_ref,
                      const double *foot_ref, const double *contact_ref, const uint8_t *contact_active,
                      const double *cop_frames, double *tau, double *dv, double *f, int32_t *status,
                      double *obs, int32_t *ncon, int32_t *con_geom, int nthreads) {
  return or_env_step_batch_env(m, params, n, q, v, qpos, qvel, qacc_ws, com_ref, posture_ref, foot_ref, contact_ref,
                               contact_active, cop_frames, NULL, tau, dv, f, status, obs, ncon, con_geom, nthreads);
}
