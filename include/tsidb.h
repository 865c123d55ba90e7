/* tsidb.h - C-ABI of the MI355X-native batched TSID + contact-dynamics path (libtsidb.so).
 *
 * The reference (UW-RoboSoccer/tsid_control) has no FFI for this path: its boundary is the Python
 * call surface of main.py:119-129,192-195 against ctrl/WalkController.py and ctrl/conf.py.  Each
 * entry point below names the reference calls it stands for.  All state/output buffers are device
 * pointers owned by the caller (torch-ROCm tensors), env-major and contiguous, in the arithmetic
 * type chosen at create time (TSIDB_F64 = the reference's float64, TSIDB_F32); the library owns only
 * the handle (model constants).  Calls are asynchronous on the given HIP stream (hipStream_t passed
 * as void*).  Return 0 = OK, non-zero = library-level failure (message via tsidb_last_error);
 * a per-env QP failure is data in status[e] (tsid HQPStatus codes: 0 optimal, 1 infeasible,
 * 2 unbounded, 3 max-iter, 4 error), never a call failure - mirroring main.py:122-124 without
 * aborting the batch: that env's tau, dv and f are 0 for the tick, its TSID state is left as it was, done = 1.
 * A handle is not thread-safe; one handle per GPU.
 */
#ifndef TSIDB_H
#define TSIDB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsidb_ctx *tsidb_handle;

enum { TSIDB_F64 = 0, TSIDB_F32 = 1 };
enum { TSIDB_NQ = 27, TSIDB_NV = 26, TSIDB_NA = 20, TSIDB_NOBS = 65, TSIDB_NROW = 67, TSIDB_MAXCON = 32 };

/* parameter vector (float64, host): RobotConfig values the reference hands to its task
 * constructors (ctrl/conf.py:21-72 via ctrl/WalkController.py:55-184) */
enum {
  TSIDB_P_DT = 0, TSIDB_P_MU, TSIDB_P_FMIN, TSIDB_P_FMAX, TSIDB_P_W_FORCEREF, TSIDB_P_KP_CONTACT,
  TSIDB_P_KD_CONTACT, TSIDB_P_W_FOOT, TSIDB_P_KP_FOOT, TSIDB_P_KD_FOOT, TSIDB_P_W_COM, TSIDB_P_KP_COM,
  TSIDB_P_KD_COM, TSIDB_P_W_POSTURE, TSIDB_P_HESS_REG, TSIDB_P_QUIRKS, TSIDB_P_NORMAL /*3*/,
  TSIDB_P_CPOINTS = TSIDB_P_NORMAL + 3 /*4x3*/, TSIDB_P_KP_POSTURE = TSIDB_P_CPOINTS + 12 /*20*/,
  TSIDB_P_KD_POSTURE = TSIDB_P_KP_POSTURE + 20, TSIDB_P_TAU_MAX = TSIDB_P_KD_POSTURE + 20,
  TSIDB_P_V_MAX = TSIDB_P_TAU_MAX + 20, TSIDB_P_MAX_ITER = TSIDB_P_V_MAX + 20, TSIDB_P_SIM_ENABLED, TSIDB_P_CLOSED_LOOP,
  TSIDB_P_W_AM /* angular-momentum task weight (legacy/biped.py:82-87), 0 = not in the stack */, TSIDB_P_KP_AM /*3*/,
  /* reward / done outputs (SURVEY.md 8d write list; no reference counterpart): reward = exp(-|com - com_ref|^2 / sigma^2)
   * - c_tau |tau|^2, done = failed QP, base height < DONE_HEIGHT or base z-axis . world z < DONE_TILT */
  TSIDB_P_REW_SIGMA = TSIDB_P_KP_AM + 3, TSIDB_P_REW_CTAU, TSIDB_P_DONE_HEIGHT, TSIDB_P_DONE_TILT,
  TSIDB_P_SELF_COLLISION /* sim: collide the robot<->robot hull pairs as mj_step does (main.py:195); 0 = floor only */,
  TSIDB_P_W_COP /* CoP force task weight (legacy/biped.py:79-80), 0 = not in the stack */,
  /* closed-loop knobs (all neutral by default = the reference's models): scale of the sim's joint
   * frictionloss (robot.xml:8), rotor inertia added to the diagonal of TSID's mass matrix on the actuated joints (the sim
   * has armature 0.005, the URDF none), Coulomb-friction feed-forward added to tau [N m] */
  TSIDB_P_SIM_FLOSS_SCALE, TSIDB_P_TSID_ARMATURE, TSIDB_P_FRICTION_COMP,
  /* sim: plane <-> mesh multi-contact rule (main.py:195, mj_step's plane-convex routine).  0: the support vertex plus EVERY
   * hull-graph neighbour of it within the margin; 1: upstream's rule as far as it is known here (mujoco is not available to
   * check): in graph order at most 3 more contacts, each at least 0.3 x the geom's bounding radius from the first */
  TSIDB_P_PLANE_MESH,
  TSIDB_P_COUNT = 128
};

/* WalkController.__init__ (ctrl/WalkController.py:12-187) + MjModel.from_xml_path (main.py:50-52):
 * parse the compiled model blob (include/tsidb_model.h), derive the constant QP blocks from
 * `params`, upload to `device`. */
int tsidb_create(const void *model_blob, size_t nbytes, const double *params, int n_params, int num_envs,
                 int device, int dtype, tsidb_handle *out);
int tsidb_destroy(tsidb_handle h);
const char *tsidb_last_error(tsidb_handle h);

/* RobotConfig edits after construction (the reference edits ctrl/conf.py and rebuilds) */
int tsidb_set_params(tsidb_handle h, const double *params, int n_params);

/* task references: comTask.setReference (WalkController.py:152), postureTask.setReference (:165),
 * task_LF/RF.setReference (:195-196), contactLF/RF.setReference (:81,122,240,248), contact on/off
 * flags (:87,128,225,232,245,253), and the init-time frames get_cop reads (:77,281-282).
 * com_ref [N,9] pos vel acc; posture_ref [N,20]; foot_ref [N,2,24] = p(3) R col-major(9) v(6) a(6);
 * contact_ref [N,2,12] = p(3) R col-major(9); contact_active [N,2] u8; cop_frames [N,2,12] =
 * R row-major(9) p(3).  Pointers are remembered, not copied. */
int tsidb_set_refs(tsidb_handle h, const void *com_ref, const void *posture_ref, const void *foot_ref,
                   const void *contact_ref, const uint8_t *contact_active, const void *cop_frames);

/* execution options that do not change results.  TSIDB_OPT_SIM_WAVES: wavefronts per env in the sim kernel - 1 (one wavefront
 * per env: the throughput-optimal shape once the batch fills the GPU) or 2 (collision phase on a second wavefront beside the
 * unconstrained dynamics: shorter step latency for small batches).  Default: 2 for up to 512 envs, else 1.  Bit-identical.
 * TSIDB_OPT_LDS_PAD (diagnostic): bytes of unused dynamic LDS added to every k_tick / k_sim workgroup (0 .. 40960) - lowers the
 * number of resident workgroups per CU, for occupancy measurements; default 0.
 * TSIDB_OPT_CU_SPLIT: whether tsidb_stream_create hands out streams on disjoint halves of the CUs: 1 always, 0 never, -1 (default)
 * for up to 512 envs.
 * TSIDB_OPT_SIM_PACK: 1 = the sim kernel runs TWO envs per wavefront (32 lanes each; tsidb_sim2.hpp) where one step per launch
 * is issued (tsidb_sim, tsidb_step, tsidb_sim_batch with one snapshot); per env bit-identical to the one-env kernel.  Only for
 * robots whose bodies, dofs, geoms and contacts fit 32 lanes (the v1 robot); the library of another robot rejects the option.
 * TSIDB_OPT_QP_FAST_EQ (default 1; float64, reference task stack): the tick first computes the equality-constrained optimum of
 * the QP by the range-space route (a Cholesky of the 6-18 x 6-18 matrix B^T B instead of the Householder QR applied to the
 * 26-50 x 26-50 factor) and runs the feasibility sweep there; an env with a violated inequality - or whose equality block is
 * ill conditioned (a pivot ratio below 1e-4) - continues with the QR and the dual active-set iterations exactly as with 0.  The
 * optimum is unique: results agree to rounding (1e-12 observed), status and iteration counts are the same. */
enum { TSIDB_OPT_SIM_WAVES = 1, TSIDB_OPT_LDS_PAD = 2, TSIDB_OPT_CU_SPLIT = 3, TSIDB_OPT_SIM_PACK = 4, TSIDB_OPT_QP_FAST_EQ = 5 };
int tsidb_set_option(tsidb_handle h, int option, int value);
int tsidb_get_option(tsidb_handle h, int option, int *value); /* the EFFECTIVE setting (TSIDB_OPT_CU_SPLIT: 1 if tsidb_stream_create
                                                                * masks its streams for this handle's batch size) */

/* HIP streams for the pipelined step (tick of step t+1 on one stream beside the sim of step t on another; the reference couples
 * the two stages one way, main.py:119-129 vs :192-195).  While every wavefront of both kernels is resident at once (up to 512
 * envs: 2 x 512 wavefronts on 1024 SIMDs) the two kernels slow each other down when they share CU groups - k_tick 54 us alone,
 * 70 us beside a running k_sim - so the streams handed out here are restricted to disjoint halves of the device's CUs
 * (hipExtStreamCreateWithCUMask): + 11-14 % env-steps/s at 256 / 512 envs; above 512 envs (each kernel alone fills more than
 * half of the SIMDs) plain streams are returned.  role: TSIDB_STREAM_TICK / TSIDB_STREAM_SIM.  Results do not depend on it. */
enum { TSIDB_STREAM_TICK = 0, TSIDB_STREAM_SIM = 1 };
int tsidb_stream_create(tsidb_handle h, int role, void **stream);
int tsidb_stream_destroy(tsidb_handle h, void *stream);

/* reference point of the CoP force task (legacy/biped.py:79-80 copTask; params[W_COP] != 0): cop_ref [N,3], world
 * frame; written by tsidb_reset (midpoint of the soles on the floor).  The pointer is remembered, not copied. */
int tsidb_set_cop_ref(tsidb_handle h, const void *cop_ref);

/* per-env randomisation of the sim stage (BASELINE.json configs[4]; no reference counterpart):
 * env_params [N,8] in the path's arithmetic type = mass scale applied to every sim body's mass and inertia,
 * contact friction, unit floor normal (3), floor offset d (plane n.x = d), 2 spare.  NULL = nominal
 * model (floor z = 0, friction 1).  terrain [N,20] (may be NULL = flat) = stepped floor: direction (2, unit, world xy),
 * phase, 1 / step length, heights[16] - the floor surface is raised along its normal by heights[cell & 15] with
 * cell = floor((direction . x_world_xy - phase) / step length) ("rough-terrain contacts", 1 cm steps).  The pointers
 * are remembered, not copied. */
int tsidb_set_env_params(tsidb_handle h, const void *env_params, const void *terrain);

/* reset: WalkController.py:22-26,72-79 (standing state, soles onto z = 0), the references of
 * :81,122,151-152,164-165, and main.py:57-64 (mj_data.qpos = q).  env_ids (device, int32) selects
 * envs; NULL = all (a non-NULL list with n_ids = 0 resets nothing).  Writes state AND the reference buffers
 * registered with tsidb_set_refs. */
int tsidb_reset(tsidb_handle h, const int32_t *env_ids, int n_ids, void *q, void *v, void *qpos, void *qvel,
                void *qacc_ws, void *stream);

/* one TSID tick for every env: main.py:119-129 (+ readouts :132-142).
 * q [N,27], v [N,26] updated in place; tau [N,20], dv [N,26], f [N,24] (LF 12, RF 12), status [N],
 * obs [N,obs_ld] = q v com cop LF RF (65 values; may be NULL; obs_ld >= 65 is the row stride in elements -
 * with obs_ld >= TSIDB_NROW columns 65, 66 receive reward and done, so that one contiguous [N,67] buffer is
 * what the all-gather sends), frames [N,2,12] sole placements R row-major + p (may be NULL), info [N,4]
 * int32 = qp iterations, active-set size, -, - (may be NULL). */
int tsidb_tick(tsidb_handle h, void *q, void *v, void *tau, void *dv, void *f, int32_t *status, void *obs,
               int obs_ld, void *frames, int32_t *info, void *stream);

/* one sim step for every env: main.py:192-195.  q_tsid [N,27] (NULL = no teleport, ctrl = 0); v_tsid
 * [N,26] (may be NULL) is used only with params[QUIRKS] = 0: the base velocity is then set together with
 * the base pose (the reference writes qpos[:7] only, which leaves the sim's base velocity to drift once
 * the TSID state moves).  qpos [N,27], qvel [N,26], qacc_ws [N,26] updated in place; qacc [N,26], ncon [N],
 * con_pairs [N,32] = (geom << 16 | hull vertex) for floor contacts, (geom2 << 16 | 0x8000 | geom1) for robot<->robot
 * ones, -1 padded (geom = collision geom in the blob's order; the v1 robot has one per body, in body order);
 * info [N,4] slots 2,3 = solver iterations, flag bits (all may be NULL).  Flag bits: 1 the damped-Euler matrix, 2 the Newton
 * Hessian was not positive definite (the step ends with what it has); 4 the step was skipped (non-finite or diverged state /
 * targets); 8 a penetrating contact was dropped at a cap (TSIDB_MAXCON contacts per env, 12 of them robot<->robot);
 * 16 a support vertex has more than 63 hull-graph neighbours (the rest is not looked at); 32 more than 64 candidate pairs
 * survived the mid phase (the rest is not collided). */
int tsidb_sim(tsidb_handle h, const void *q_tsid, const void *v_tsid, void *qpos, void *qvel, void *qacc_ws,
              void *qacc, int32_t *ncon, int32_t *con_pairs, int32_t *info, void *stream);

/* n_steps (1 .. TSIDB_MAX_SIM_BATCH) consecutive sim steps in ONE launch: step b teleports to / takes its joint targets from
 * slot slots[b] (host array, 0 .. 15) of the snapshot rings q_ring [K,N,27], v_ring [K,N,26] (v_ring may be NULL) - what n_steps
 * calls of tsidb_sim with q_tsid = q_ring[slots[b]] do, without the launch gaps between them (the pipelined open-loop step
 * hands over the TSID states of several ticks at once; envs do not interact, so each steps on its own).  Bit for bit, in
 * float64 and float32: the library is built with -ffp-contract=on, so the separately compiled instantiations of the sim kernel
 * (single- / multi-step, one / two wavefronts per env, two envs per wavefront) fuse exactly the multiply-adds the source writes
 * as one expression (with hipcc's default, fast, the multi-step kernel differed from the single-step one by 1 ulp in float32
 * after 231 walking steps; tests: test_shard_invariance_across_kernel_shapes).
 * ncon / con_pairs / info are the last step's. */
enum { TSIDB_MAX_SIM_BATCH = 8 };
int tsidb_sim_batch(tsidb_handle h, int n_steps, const void *q_ring, const void *v_ring, const int32_t *slots, void *qpos, void *qvel,
                    void *qacc_ws, void *qacc, int32_t *ncon, int32_t *con_pairs, int32_t *info, void *stream);

/* whole env step, n_substeps times: tsidb_tick then (if params[SIM_ENABLED]) tsidb_sim.
 * With params[CLOSED_LOOP] (not in the reference, whose coupling is one-way, main.py:126-129,
 * 192-195): each tick first reads the TSID state from the sim state (quat wxyz -> xyzw, world-frame base
 * linear velocity -> body frame, sim joint order -> TSID order), and the sim stage applies tau as motor
 * torques and keeps its own base pose instead of the teleport + position servos. */
int tsidb_step(tsidb_handle h, void *q, void *v, void *qpos, void *qvel, void *qacc_ws, void *tau, void *dv,
               void *f, int32_t *status, void *obs, int obs_ld, void *frames, int32_t *ncon, int32_t *con_pairs,
               int32_t *info, int n_substeps, void *stream);

/* walking reference update for every env, on the device (config 3): what the reference's main.py:117
 * intends with controller.update_tasks(sampleLF, sampleRF, contact_LF, contact_RF) when the samples
 * come from ctrl/Walk_Planner.py:23-31 swing trajectories (ctrl/Foot_Trajectory.py polynomials) over
 * a ctrl/Footstep_Planner.py plan.  coef [N,K,4,4] = x, y, z, yaw cubic coefficients (ascending, in
 * time since the step started); side [N,K] int32 = swinging foot of step k (0 left); nsteps [N] int32;
 * rest [N,K+1,2,4] = (x, y, yaw, z) of [left, right] foot before step k; com [N,K+2,2,3] = linear-inverted-
 * pendulum segment (zmp, d, c) per planar axis for the start phase, each step and the final stand
 * (ctrl/LIPM.py:34-49 about a fixed ZMP, in closed form x(s) = zmp + d/2 e^{omega s} + c e^{-omega s}).
 * Timeline: [0, t_start) both feet down while the CoM height goes from com_z0 to com_z0 - com_drop; step
 * k occupies [t_start + k T, t_start + (k+1) T); afterwards both feet are down.  Writes the registered
 * foot_ref / contact_ref / contact_active / com_ref (position, velocity, acceleration) buffers; contact
 * on/off edges re-reference at `frames` [N,2,12] (current sole placements from the last tsidb_tick), as
 * ctrl/WalkController.py:215-253 intends.  t_offset [N] (may be NULL) delays each env's timeline: env e runs
 * on the clock max(t - t_offset[e], 0), so that the envs of one batch need not step in phase.  Contact-timing
 * feedback (closed loop; td_latch [N] int32, initialised to -1, may be NULL = off): when the last sim step's contact
 * list (ncon, con_pairs of tsidb_sim / tsidb_step) shows the swing foot on the floor after td_fraction of its
 * swing, the touch-down is taken at once - the foot is a stance foot for the rest of that step.  t_device (may be
 * NULL): one float64 value (whatever the path's arithmetic type) in device memory that replaces `t` - the launch can
 * then be captured in a HIP graph and replayed while the caller advances the clock on the device. */
int tsidb_walk_update(tsidb_handle h, const void *coef, const int32_t *side, const int32_t *nsteps,
                      const void *rest, const void *com, int K, double t, double step_duration, double t_start,
                      double omega, double com_z0, double com_drop, const void *frames, const void *t_offset,
                      const int32_t *ncon, const int32_t *con_pairs, int32_t *td_latch, double td_fraction,
                      const void *t_device, void *stream);

/* ---- episode lifecycle on the device (SURVEY.md 8f-2, section 5 "auto-reset mask"; no reference counterpart: the reference
 * runs one episode and exits, main.py:113-124) */

/* tsidb_reset for exactly the envs whose done flag is set: rows [N, rows_ld >= TSIDB_NROW] as tsidb_tick wrote them
 * (done in column TSIDB_NOBS + 1) - no host round trip between `done` and the restart.  frames (may be NULL) [N,2,12]
 * receives the reset envs' sole placements (what the next tsidb_walk_update re-references contacts at). */
int tsidb_reset_done(tsidb_handle h, const void *rows, int rows_ld, void *q, void *v, void *qpos, void *qvel, void *qacc_ws,
                     void *frames, void *stream);

/* [NA] values (the path's arithmetic type, device) added to the posture reference every reset captures
 * (ctrl/WalkController.py:164-165 takes q0's joints; a walking workload keeps its knees bent).  NULL = none.  The pointer
 * is remembered, not copied. */
int tsidb_set_posture_bias(tsidb_handle h, const void *posture_bias);

/* Episode plan for the selected envs, built on the device: env_ids (device int32, NULL = all) and / or done_rows (as for
 * tsidb_reset_done: only envs whose done flag is set).  For each: footsteps along its path (ctrl/Footstep_Planner.py:92-125),
 * swing polynomials from footstep k to k + 2 (ctrl/Walk_Planner.py:23-31, ctrl/Foot_Trajectory.py:5-27), rest placements and
 * the LIPM / DCM CoM plan (ctrl/LIPM.py:34-49 in closed form) - the tables tsidb_walk_update reads (layouts there), starting
 * from the sole placements and the CoM that tsidb_reset left in the registered cop_frames / com_ref buffers.
 * plan_params [TSIDB_PLAN_NPARAMS] (host): step_length, step_width, step_height, step_duration, rise_ratio
 * (ctrl/conf.py:24-28), t_start, com_drop, foot_press (every swing is aimed this far below the floor), resample_ds (path
 * vertices at most this far apart before planning; 0 = as given), unicycle path v, w, dt, n (Footstep_Planner.py:131-141),
 * scale_lo, scale_hi, seed.  Path per env: `path` [N,P,2] float64 + npts [N] = an explicit polyline in world coordinates,
 * used as given; NULL = the unicycle path scaled by scale[e] ([N] float64) or, with scale NULL and episode [N] int32 given,
 * by U(scale_lo, scale_hi) drawn from hash(seed, env, episode[e]) (bump_episode != 0 increments episode[e] first), rotated
 * into the robot's heading and started between its feet.  Outputs: steps [N,K+2,4] float64 = x, y, yaw, side of every
 * footstep (the two initial ones first); nsteps [N]; coef, side, rest, com as tsidb_walk_update reads them; flags [N] (may
 * be NULL) bit 0 = the plan needed more than K steps and was cut, bit 1 = the path had no direction (fewer than two distinct
 * vertices; npts[e] is clamped to P): no step planned, the env stands; bit 2 = a path piece longer than 4096 resample intervals
 * (or with a non-finite vertex) was resampled coarser; bit 3 (with bit 1) = the CoM is not above the feet after the descent
 * (com_ref[2] - com_drop <= 1 mm: plan before a reset, or com_drop >= the standing height): no step planned, finite tables.
 * Rejected by the call (error): non-finite parameters, step_width <= 0, resample_ds in (0, step_length / 1000), a unicycle
 * path of more than 65536 vertices or with dt <= 0, a scale range that is not 0 < lo <= hi.  The env's clock restarts: t_offset [N] (may be NULL)
 * receives `t` (or *t_device, float64 device), td_latch [N] (may be NULL) -1. */
enum { TSIDB_PLAN_NPARAMS = 16 };
int tsidb_walk_plan(tsidb_handle h, const int32_t *env_ids, int n_ids, const void *done_rows, int rows_ld,
                    const double *plan_params, int n_plan_params, const double *path, const int32_t *npts, int P,
                    const double *scale, int32_t *episode, int bump_episode, int K, double *steps, void *coef, int32_t *side,
                    int32_t *nsteps, void *rest, void *com, int32_t *flags, void *t_offset, int32_t *td_latch, double t,
                    const double *t_device, void *stream);

/* tsidb_walk_update's arguments as one block (same meaning, same order) */
typedef struct tsidb_walk_args {
  const void *coef; const int32_t *side; const int32_t *nsteps; const void *rest; const void *com; int K;
  double t, step_duration, t_start, omega, com_z0, com_drop;
  const void *frames; const void *t_offset; const int32_t *ncon; const int32_t *con_pairs; int32_t *td_latch; double td_fraction;
  const double *t_device;
} tsidb_walk_args;

/* tsidb_tick with two fusions for the tick stream of a pipelined step (what bounds small batches: three launches less):
 * walk (may be NULL): this tick's walking reference update (exactly tsidb_walk_update(walk...), run per env in the tick kernel's
 * prologue: controller.update_tasks(...) of main.py:117 and formulation.computeProblemData of main.py:119 in one launch);
 * q_snapshot / v_snapshot (may be NULL) [N,27] / [N,26]: a second copy of the TSID state the tick ends on, for a sim stage that
 * runs on another stream while the next tick already overwrites q / v.  Results are those of the separate calls, bit for bit. */
int tsidb_tick_walk(tsidb_handle h, const tsidb_walk_args *walk, void *q, void *v, void *tau, void *dv, void *f, int32_t *status,
                    void *obs, int obs_ld, void *frames, int32_t *info, void *q_snapshot, void *v_snapshot, void *stream);

/* probe of formulation.computeProblemData's rigid-body terms (main.py:119): M [N,26,26],
 * hbias [N,26], Jcom [N,3,26], Jf [N,2,6,26] (LOCAL), oMf [N,2,12], com [N,3].  Test/debug use. */
int tsidb_rbd_terms(tsidb_handle h, const void *q, const void *v, void *M, void *hbias, void *Jcom, void *Jf,
                    void *oMf, void *com, void *stream);

/* dimensions of the robot this library was built for (one library per robot: libtsidb.so = the v1 robot of ctrl/conf.py:9-15,
 * libtsidb_v0.so = robot/v0): out9 = NJ, NQ, NV, NA, sim bodies, 1 if the sim stage is built, collision geoms, contact
 * dimension, 1 if joints are damped - the nine ints of the blob's model_dims section, which tsidb_create compares.  The
 * TSIDB_N* constants above are the v1 robot's. */
int tsidb_dims(int *out9);

/* bytes of LDS one env occupies in kernel `which` (0 tick, 1 sim) for `dtype` */
int tsidb_lds_bytes(int dtype, int which);

#ifdef __cplusplus
}
#endif
#endif
