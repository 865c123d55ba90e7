/* or_walk.c - walking reference update for one tick (TEST INFRASTRUCTURE; see oracle.h).
 *
 * Independent CPU restatement of what libtsidb.so's tsidb_walk_update does on the device, written from
 * the semantics of the reference pieces it stands for, not from the kernel:
 *   - swing-foot sample = ctrl/Foot_Trajectory.py:21-27 evaluated as one polynomial per coordinate
 *     (x, y, yaw: the 2-knot CubicSpline is a straight line; z: the 3-knot one is a parabola, the 4-knot
 *     not-a-knot one a single cubic), between footstep k and k+2 (ctrl/Walk_Planner.py:23-31);
 *   - contact switching = ctrl/WalkController.py:189-209 (update_tasks: set both foot references, then
 *     add_contact / remove_contact on the edges of the flags) with the working bodies of
 *     legacy/biped.py:168-212 (re-reference at the CURRENT placement);
 *   - CoM reference = ctrl/LIPM.py:34-49 about a fixed ZMP in closed form,
 *     x(s) = zmp + d/2 e^{w s} + c e^{-w s}, quintic descent in height during the start phase.
 * Polynomials are evaluated by Horner's rule here (the device sums monomials): agreement is to rounding,
 * not bit-exact.
 *
 * Tables (env-major): coef [n,K,4,4] x y z yaw ascending in time since the step began; side [n,K];
 * nsteps [n]; rest [n,K+1,2,4] = (x, y, yaw, z) of [left, right] before step k; com [n,K+2,2,3] =
 * (zmp, d, c) per planar axis for the start phase, each step, the final stand.  frames [n,2,12] =
 * current sole placements R row-major + p.  t_off (may be NULL): per-env start delay; env time =
 * max(t - t_off[e], 0).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

static double horner(const double *c, double s) { return c[0] + s * (c[1] + s * (c[2] + s * c[3])); }
static double horner_d1(const double *c, double s) { return c[1] + s * (2.0 * c[2] + s * 3.0 * c[3]); }
static double horner_d2(const double *c, double s) { return 2.0 * c[2] + 6.0 * c[3] * s; }

/* tsid SE3ToVector layout of a placement given as R row-major (9) + p (3): p, then R column-major */
static void se3vec(const double *fr, double *out12) {
  for (int i = 0; i < 3; i++) out12[i] = fr[9 + i];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 3; row++) out12[3 + 3 * col + row] = fr[3 * row + col];
}

void or_walk_update(int n, const double *coef, const int32_t *side, const int32_t *nsteps, const double *rest,
                    const double *com, int K, double t, const double *t_off, double T, double t_start, double omega,
                    double z0, double dz, const double *frames, double *foot_ref, double *contact_ref,
                    uint8_t *contact_active, double *com_ref) {
  or_walk_update_fb(n, coef, side, nsteps, rest, com, K, t, t_off, T, t_start, omega, z0, dz, frames, foot_ref, contact_ref,
                    contact_active, com_ref, NULL, NULL, NULL, 0, 0, 0.0);
}

/* the same with contact-timing feedback (closed loop): latch [n] (initialised to -1, NULL = off) - when the last
 * sim step's contact list (ncon [n], con_geom [n, OR_MAXCON] = geom << 16 | vertex) shows one of the swing foot's
 * geoms (bit set of fgeoms[side]) on the floor after td_frac of its swing, the touch-down is taken at once: the foot
 * counts as a stance foot for the rest of that step */
void or_walk_update_fb(int n, const double *coef, const int32_t *side, const int32_t *nsteps, const double *rest,
                       const double *com, int K, double t, const double *t_off, double T, double t_start, double omega,
                       double z0, double dz, const double *frames, double *foot_ref, double *contact_ref,
                       uint8_t *contact_active, double *com_ref, const int32_t *ncon, const int32_t *con_geom,
                       int32_t *latch, uint64_t fgeoms0, uint64_t fgeoms1, double td_frac) {
  for (int e = 0; e < n; e++) {
    double te = t - (t_off ? t_off[e] : 0.0);
    if (te < 0) te = 0;
    const int ns = nsteps[e];
    /* phase: k = -1 during the double-support start, else the running step; s = time inside it */
    int k = -1;
    double s = te;
    if (te >= t_start) {
      k = (int)floor((te - t_start) / T);
      s = (te - t_start) - k * T;
    }
    const int walking = k >= 0 && k < ns;
    const int kpos = k < 0 ? 0 : k;
    const int kstep = ns > 0 ? (kpos < ns - 1 ? kpos : ns - 1) : 0; /* step whose polynomial / side applies */
    const int krest = kpos < ns ? kpos : ns;
    const double *c = coef + ((size_t)e * K + kstep) * 16;
    const int sw_side = side[(size_t)e * K + kstep];
    int early = 0;
    if (latch) {
      early = walking && latch[e] == k;
      if (walking && !early && s > td_frac * T) {
        const uint64_t fg = sw_side == 0 ? fgeoms0 : fgeoms1;
        for (int c = 0; c < ncon[e]; c++) {
          const int cp = con_geom[(size_t)e * OR_MAXCON + c];
          if (((fg >> (cp >> 16)) & 1u) && !(cp & 0x8000)) early = 1;
        }
      }
      if (early) latch[e] = k;
    }
    for (int f = 0; f < 2; f++) {
      const int swing = walking && sw_side == f && !early;
      double smp[24];
      memset(smp, 0, sizeof smp);
      double yaw;
      if (swing) {
        smp[0] = horner(c + 0, s); smp[1] = horner(c + 4, s); smp[2] = horner(c + 8, s);
        yaw = horner(c + 12, s);
        smp[12] = horner_d1(c + 0, s); smp[13] = horner_d1(c + 4, s); smp[14] = horner_d1(c + 8, s);
        smp[17] = horner_d1(c + 12, s);
        smp[18] = horner_d2(c + 0, s); smp[19] = horner_d2(c + 4, s); smp[20] = horner_d2(c + 8, s);
        smp[23] = horner_d2(c + 12, s);
      } else {
        const double *r = rest + (((size_t)e * (K + 1) + krest) * 2 + f) * 4;
        smp[0] = r[0]; smp[1] = r[1]; smp[2] = r[3];
        yaw = r[2];
      }
      /* Rz(yaw), column-major */
      smp[3] = cos(yaw); smp[4] = sin(yaw); smp[6] = -sin(yaw); smp[7] = cos(yaw); smp[11] = 1.0;
      double *fo = foot_ref + (size_t)e * 48 + 24 * f;
      memcpy(fo, smp, sizeof smp); /* task_LF/RF.setReference (WalkController.py:195-196) */
      const int want_contact = !swing, active = contact_active[(size_t)e * 2 + f] != 0;
      double cur[12];
      se3vec(frames + (size_t)e * 24 + 12 * f, cur);
      if (want_contact && !active) { /* add_contact: the contact re-references at the current placement */
        memcpy(contact_ref + (size_t)e * 24 + 12 * f, cur, sizeof cur);
        contact_active[(size_t)e * 2 + f] = 1;
      } else if (!want_contact && active) { /* remove_contact: the foot task restarts at the current placement */
        memcpy(fo, cur, sizeof cur);
        memset(fo + 12, 0, 12 * sizeof(double));
        contact_active[(size_t)e * 2 + f] = 0;
      }
    }
    /* CoM reference */
    int ph = k + 1;
    if (ph > ns + 1) ph = ns + 1;
    double sc = s;
    if (k >= 0 && ph > ns) sc = (te - t_start) - ns * T; /* the final stand's clock keeps running */
    double *cr = com_ref + (size_t)e * 9;
    for (int a = 0; a < 2; a++) {
      const double *sg = com + (((size_t)e * (K + 2) + ph) * 2 + a) * 3;
      const double ep = exp(omega * sc), em = exp(-omega * sc);
      cr[a] = sg[0] + 0.5 * sg[1] * ep + sg[2] * em;
      cr[3 + a] = omega * (0.5 * sg[1] * ep - sg[2] * em);
      cr[6 + a] = omega * omega * (0.5 * sg[1] * ep + sg[2] * em);
    }
    double q = 1.0;
    if (t_start > 0 && te < t_start) q = te / t_start;
    const double sz = q * q * q * (10.0 - 15.0 * q + 6.0 * q * q);
    const double dsz = t_start > 0 ? 30.0 * q * q * (1 - q) * (1 - q) / t_start : 0.0;
    const double ddsz = t_start > 0 ? 60.0 * q * (1 - q) * (1 - 2 * q) / (t_start * t_start) : 0.0;
    cr[2] = z0 - dz * sz; cr[5] = -dz * dsz; cr[8] = -dz * ddsz;
  }
}

/* ================================================================= episode plan (device twin: k_plan / tsidb_walk_plan)
 * Everything a walking episode needs, per env, from a path: footsteps (ctrl/Footstep_Planner.py:92-125), the swing
 * polynomials between footstep k and k+2 (ctrl/Walk_Planner.py:23-31, ctrl/Foot_Trajectory.py:5-27), and the CoM plan:
 * DCM end points backwards from the final stand, then one LIPM segment (ctrl/LIPM.py:34-49 in closed form) per phase.
 * Written from those semantics, not from the kernel; the host-side WalkSchedule.__init__ builds the same tables with numpy.
 *
 * pp[16]: step_length, step_width, step_height, step_duration, rise_ratio, t_start, com_drop, foot_press, resample_ds,
 *         unicycle v, w, dt, n (Footstep_Planner.py:131-141), scale_lo, scale_hi, seed.
 * path [n,P,2] + npts [n]: explicit polyline per env, used as given; NULL = the unicycle path, scaled by scale[e] (or by
 *         U(scale_lo, scale_hi) drawn from hash(seed, env, episode[e]) when scale is NULL and episode is not), resampled to
 *         vertices at most resample_ds apart, rotated into the robot's heading and started between its feet.
 * cop_frames [n,2,12] (R row-major, p): the sole placements the episode starts from; com_ref [n,9]: its CoM.
 * Outputs: steps [n,K+2,4] = x, y, yaw (world), side; nsteps [n]; coef [n,K,4,4]; side [n,K]; rest [n,K+1,2,4]; com
 * [n,K+2,2,3]; flags [n] bit 0 = the plan had more than K steps and was cut, bit 1 = the
 * path had no direction (fewer than two distinct vertices): no step planned. */
uint64_t or_plan_hash(uint64_t seed, uint64_t env, uint64_t episode) {
  uint64_t x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (episode * 0xD1B54A32D192ED03ull);
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

typedef struct { double x, y, yaw; int side; } PlanStep;

static void plan_add_step(const double *pp, double dx, double dy, int side, const double *pos, PlanStep *out) {
  const double nrm = sqrt(dx * dx + dy * dy), tx = dx / nrm, ty = dy / nrm;
  const double sign = side == 0 ? 1.0 : -1.0;
  out->x = pos[0] + tx * (pp[0] / 2) + (-ty) * (pp[1] / 2 * sign);
  out->y = pos[1] + ty * (pp[0] / 2) + tx * (pp[1] / 2 * sign);
  out->yaw = atan2(dy, dx);
  out->side = side;
}

/* lowest-degree polynomial through (x_i, f_i), x_0 = 0, nk = 2..4 knots: Newton's divided differences expanded to
 * ascending monomial coefficients */
static void poly_through(int nk, const double *x, const double *f, double *c) {
  double d1[3] = {0, 0, 0}, d2[2] = {0, 0}, d3 = 0;
  for (int i = 0; i + 1 < nk; i++) d1[i] = (f[i + 1] - f[i]) / (x[i + 1] - x[i]);
  for (int i = 0; i + 2 < nk; i++) d2[i] = (d1[i + 1] - d1[i]) / (x[i + 2] - x[i]);
  if (nk == 4) d3 = (d2[1] - d2[0]) / (x[3] - x[0]);
  const double x1 = nk > 2 ? x[1] : 0.0, x2 = nk > 3 ? x[2] : 0.0;
  c[0] = f[0];
  c[1] = d1[0] - d2[0] * x1 + d3 * x1 * x2;
  c[2] = d2[0] - d3 * (x1 + x2);
  c[3] = d3;
}

void or_walk_plan(int n, const double *pp, const double *cop_frames, const double *com_ref, const double *path,
                  const int32_t *npts, int P, const double *scale, const int32_t *episode, int K, double *steps_out,
                  double *coef, int32_t *side, int32_t *nsteps, double *rest, double *com, int32_t *flags) {
  const double L = pp[0], T = pp[3], rise_ratio = pp[4], t_start = pp[5], foot_press = pp[7], ds = pp[8];
  const double GRAV = 9.80665;
  for (int e = 0; e < n; e++) {
    const double *fr = cop_frames + (size_t)e * 24;
    const double lf[2] = {fr[9], fr[10]}, rf[2] = {fr[21], fr[22]};
    const double com0[3] = {com_ref[(size_t)e * 9], com_ref[(size_t)e * 9 + 1], com_ref[(size_t)e * 9 + 2]};
    const double heading = atan2(-(lf[0] - rf[0]), lf[1] - rf[1]);
    const double ch = cos(heading), sh = sin(heading), mid[2] = {0.5 * (lf[0] + rf[0]), 0.5 * (lf[1] + rf[1])};
    PlanStep *st = (PlanStep *)malloc(sizeof(PlanStep) * (size_t)(K + 4));
    int nst = 0, cut = 0;
    st[nst++] = (PlanStep){lf[0], lf[1], heading, 0};
    st[nst++] = (PlanStep){rf[0], rf[1], heading, 1};
    /* ---- Footstep_Planner.py:92-125 over the (resampled) polyline */
    int sd = 1;
    double travelled = 0, dx = 0, dy = 0, prev[2] = {0, 0}, a[2] = {0, 0};
    double sc = 1.0;
    if (!path) {
      if (scale) sc = scale[e];
      else if (episode) sc = pp[13] + (pp[14] - pp[13]) * ((double)(or_plan_hash((uint64_t)pp[15], (uint64_t)e, (uint64_t)episode[e]) >> 11) * (1.0 / 9007199254740992.0));
    }
    int nv = path ? npts[e] : (int)pp[12];
    if (path && nv > P) nv = P;
    double ux = 0, uy = 0, uth = 0;
    for (int i = 0; i < nv; i++) {
      double b[2];
      if (path) { b[0] = path[((size_t)e * P + i) * 2]; b[1] = path[((size_t)e * P + i) * 2 + 1]; }
      else {
        ux += pp[9] * pp[11] * cos(uth); uy += pp[9] * pp[11] * sin(uth); uth += pp[10] * pp[11];
        const double px = ux * sc, py = uy * sc;
        b[0] = (ch * px - sh * py) + mid[0]; b[1] = (sh * px + ch * py) + mid[1];
      }
      if (i == 0) { prev[0] = a[0] = b[0]; prev[1] = a[1] = b[1]; continue; }
      const double seg = sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]));
      int m = ds > 0 ? (int)ceil(seg / ds) : 1;
      if (m < 1) m = 1;
      for (int j = 1; j <= m; j++) {
        const double q = (double)j / m, p[2] = {a[0] + (b[0] - a[0]) * q, a[1] + (b[1] - a[1]) * q};
        const double ddx = p[0] - prev[0], ddy = p[1] - prev[1];
        if (ddx != 0 || ddy != 0) { dx = ddx; dy = ddy; } /* (a repeated vertex keeps the direction of the piece before it) */
        travelled += hypot(ddx, ddy);
        if (travelled >= L) {
          sd = !sd;
          if (nst < K + 2) plan_add_step(pp, dx, dy, sd, prev, &st[nst++]); else cut = 1;
          travelled = 0;
        }
        prev[0] = p[0]; prev[1] = p[1];
      }
      a[0] = b[0]; a[1] = b[1];
    }
    /* closing step at the end of the path, one more if the last stretch was partial (:114-123); a path without a direction
     * (fewer than two distinct vertices: the reference raises) plans no step, flag bit 1 */
    const int degenerate = !(dx != 0 || dy != 0);
    if (!degenerate) {
      sd = !sd;
      if (nst < K + 2) plan_add_step(pp, dx, dy, sd, prev, &st[nst++]); else cut = 1;
      if (travelled > 0) {
        sd = !sd;
        if (nst < K + 2) plan_add_step(pp, dx, dy, sd, prev, &st[nst++]); else cut = 1;
      }
    }
    const int ns = nst - 2;
    nsteps[e] = ns;
    if (flags) flags[e] = cut | (degenerate ? 2 : 0);
    for (int k = 0; k < K + 2; k++) {
      double *o = steps_out + ((size_t)e * (K + 2) + k) * 4;
      if (k < nst) { o[0] = st[k].x; o[1] = st[k].y; o[2] = st[k].yaw; o[3] = st[k].side; }
      else o[0] = o[1] = o[2] = o[3] = 0;
    }
    /* ---- swing polynomials and rest placements (Walk_Planner.py:23-31): yaw relative to the initial heading */
    const double hd = heading;
    double cur[2][4]; /* x, y, yaw, z of [left, right] */
    for (int s2 = 0; s2 < 2; s2++) { cur[st[s2].side][0] = st[s2].x; cur[st[s2].side][1] = st[s2].y; cur[st[s2].side][2] = st[s2].yaw - hd; cur[st[s2].side][3] = 0; }
    for (int k = 0; k <= K; k++) {
      memcpy(rest + ((size_t)e * (K + 1) + k) * 8, cur, sizeof cur);
      if (k >= K) break;
      double *c = coef + ((size_t)e * K + k) * 16;
      memset(c, 0, 16 * sizeof(double));
      side[(size_t)e * K + k] = 0;
      if (k < ns) {
        const int sw = st[k].side;
        const double nxt[4] = {st[k + 2].x, st[k + 2].y, st[k + 2].yaw - hd, -foot_press}, *c0 = cur[sw];
        const double x2[2] = {0, T};
        double f2[2];
        f2[0] = c0[0]; f2[1] = nxt[0]; poly_through(2, x2, f2, c + 0);
        f2[0] = c0[1]; f2[1] = nxt[1]; poly_through(2, x2, f2, c + 4);
        f2[0] = c0[2]; f2[1] = nxt[2]; poly_through(2, x2, f2, c + 12);
        if (rise_ratio != 0.5) {
          const double rise = T * rise_ratio, x4[4] = {0, rise, T - rise, T}, f4[4] = {c0[3], c0[3] + pp[2], nxt[3] + pp[2], nxt[3]};
          poly_through(4, x4, f4, c + 8);
        } else {
          const double x3[3] = {0, T * rise_ratio, T}, f3[3] = {c0[3], c0[3] + pp[2], nxt[3]};
          poly_through(3, x3, f3, c + 8);
        }
        side[(size_t)e * K + k] = sw;
        memcpy(cur[sw], nxt, sizeof nxt);
      }
    }
    /* ---- CoM plan: DCM end points backwards from the final stand, then forwards one LIPM segment per phase */
    const double w = sqrt(GRAV / (com0[2] - pp[6])), ewT = exp(-w * T);
    double *cm = com + (size_t)e * (K + 2) * 6;
    double fin[2] = {com0[0], com0[1]}, xi[2];
    if (ns > 0) { fin[0] = 0.5 * (st[ns].x + st[ns + 1].x); fin[1] = 0.5 * (st[ns].y + st[ns + 1].y); }
    xi[0] = fin[0]; xi[1] = fin[1];
    for (int k = ns - 1; k >= 0; k--) { /* xi_k kept in the d slot of phase k + 1 until the forward pass */
      const double z[2] = {st[k + 1].x, st[k + 1].y};
      for (int ax = 0; ax < 2; ax++) { xi[ax] = z[ax] + (xi[ax] - z[ax]) * ewT; cm[((k + 1) * 2 + ax) * 3 + 1] = xi[ax]; }
    }
    double x[2] = {com0[0], com0[1]};
    {
      const double E0 = exp(w * t_start), ep = exp(w * t_start), em = exp(-w * t_start);
      for (int ax = 0; ax < 2; ax++) {
        const double x0k = ns > 0 ? cm[(1 * 2 + ax) * 3 + 1] : fin[ax];
        const double z = (x0k - x[ax] * E0) / (1.0 - E0), d = x[ax] - z, c2 = (x[ax] - z) - 0.5 * d;
        cm[ax * 3] = z; cm[ax * 3 + 1] = d; cm[ax * 3 + 2] = c2;
        x[ax] = z + (0.5 * d * ep + c2 * em);
      }
    }
    {
      const double ep = exp(w * T), em = exp(-w * T);
      for (int k = 0; k < ns; k++)
        for (int ax = 0; ax < 2; ax++) {
          double *sg = cm + ((k + 1) * 2 + ax) * 3;
          const double z = ax == 0 ? st[k + 1].x : st[k + 1].y, d = sg[1] - z, c2 = (x[ax] - z) - 0.5 * d;
          sg[0] = z; sg[1] = d; sg[2] = c2;
          x[ax] = z + (0.5 * d * ep + c2 * em);
        }
    }
    for (int k = ns + 1; k < K + 2; k++)
      for (int ax = 0; ax < 2; ax++) { double *sg = cm + (k * 2 + ax) * 3; sg[0] = fin[ax]; sg[1] = 0; sg[2] = x[ax] - fin[ax]; }
    free(st);
  }
}
