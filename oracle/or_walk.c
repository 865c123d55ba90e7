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
