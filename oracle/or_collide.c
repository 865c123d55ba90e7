/* or_collide.c - terrain height and convex-hull <-> convex-hull narrow phase (TEST INFRASTRUCTURE; see oracle.h).
 *
 * What mujoco.mj_step (main.py:195) does for the robot<->robot candidate geom pairs that survive the
 * excludes and the parent-child / same-body filter (170 for robot/v1/mujoco/robot.xml:18-52, 1044 for
 * robot/v0/robot.xml whose "visual" meshes collide too: they inherit contype 1 from robot.xml:4): each pair of
 * collision meshes is collided as a pair of convex hulls.  A geom margin (robot/v0/robot.xml:4) inflates both
 * hulls by margin / 2 along the support direction (mjc_Convex's support function); the caller then takes
 * dist = margin - depth and keeps the contact while dist < margin.  mujoco is not vendored: PARITY
 * UNPINNED.  This restates the published algorithm MuJoCo's mjc_Convex has used for mesh pairs - Minkowski
 * Portal Refinement (Snethen 2008, "XenoCollide"; libccd's ccdMPRPenetration): portal discovery from the
 * interior point centre1 - centre2, portal refinement, penetration = distance from the origin to the final
 * portal triangle, contact point from the portal's barycentric weights, ONE contact per pair (multiccd is
 * off by default), dist = -depth, normal from geom1 to geom2.
 * Declared deviations (so that the GPU path can match to rounding instead of to the solver tolerance):
 *  - support vertex: exhaustive arg-max, ties within 1e-12 to the lowest index (MuJoCo hill-climbs);
 *  - refinement runs until the portal stops advancing by more than 1e-10 m (MuJoCo: mpr_tolerance 1e-6,
 *    at most mpr_iterations = 50); the cap here is 64 iterations;
 *  - mid phase: bounding spheres, then a 15-axis separating-axis test on the hulls' body-frame boxes
 *    (MuJoCo filters with AABBs + bounding spheres; any conservative filter gives the same contacts).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define MPR_TOL 1e-10
#define MPR_MAXIT 64
#define SUP_TIE 1e-12

static void cross(const double *a, const double *b, double *c) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static double dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void sub(const double *a, const double *b, double *c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
static int normalize(double *a) {
  double n = sqrt(dot(a, a));
  if (!(n > 0)) return 0;
  a[0] /= n; a[1] /= n; a[2] /= n;
  return 1;
}

/* ---- terrain: per-env stepped floor (BASELINE.json configs[4] "rough-terrain contacts"; SURVEY.md 8d cfg5
 * "random plane tilt + 1 cm steps"; no reference counterpart).  terr = dir_x, dir_y, phase, 1 / step_len,
 * heights[16]: the floor surface is the plane n.x = d raised along n by heights[cell mod 16], cell =
 * floor((dir . x_world_xy - phase) / step_len).  NULL = flat. */
double or_terrain_height(const double *terr, double X, double Y) {
  if (!terr) return 0.0;
  const double u = terr[0] * X + terr[1] * Y;
  int cell = (int)floor((u - terr[2]) * terr[3]);
  cell &= 15;
  return terr[4 + cell];
}

/* support vertex of geom b's hull (its body placed at R, p) along the world direction d: index (hull-local) and world point */
static int hull_support(const OrModel *m, int b, const double *R, const double *p, const double *d, double *w) {
  const double db[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                        R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
  const int v0 = m->hull_adr[b], v1 = m->hull_adr[b + 1];
  double best = -INFINITY;
  for (int i = v0; i < v1; i++) {
    const double s = dot(m->hull_vert + 3 * i, db);
    if (s > best) best = s;
  }
  int bi = v0;
  for (int i = v0; i < v1; i++)
    if (dot(m->hull_vert + 3 * i, db) >= best - SUP_TIE) { bi = i; break; }
  const double *v = m->hull_vert + 3 * bi;
  for (int i = 0; i < 3; i++) w[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2] + p[i];
  return bi - v0;
}

typedef struct { double v[3]; int ia, ib; } Sup; /* point of A - B, with the hull vertices it came from */

typedef struct {
  const OrModel *m;
  int a, b;
  const double *Ra, *pa, *Rb, *pb;
  double margin;
} Pair;

/* dir is a unit vector at every call */
static void support(const Pair *P, const double *dir, Sup *s) {
  double wa[3], wb[3], nd[3] = {-dir[0], -dir[1], -dir[2]};
  s->ia = hull_support(P->m, P->a, P->Ra, P->pa, dir, wa);
  s->ib = hull_support(P->m, P->b, P->Rb, P->pb, nd, wb);
  sub(wa, wb, s->v);
  /* each hull grown by margin / 2 along its own support direction: A - B grows by margin along dir */
  if (P->margin != 0) for (int i = 0; i < 3; i++) s->v[i] += P->margin * dir[i];
}

static void portal_dir(const Sup *v1, const Sup *v2, const Sup *v3, double *dir) {
  double e1[3], e2[3];
  sub(v2->v, v1->v, e1);
  sub(v3->v, v1->v, e2);
  cross(e1, e2, dir);
  normalize(dir);
}

static void expand_portal(const Sup *v0, Sup *v1, Sup *v2, Sup *v3, const Sup *v4) {
  double v4v0[3];
  cross(v4->v, v0->v, v4v0);
  if (dot(v1->v, v4v0) > 0) {
    if (dot(v2->v, v4v0) > 0) *v1 = *v4; else *v3 = *v4;
  } else {
    if (dot(v3->v, v4v0) > 0) *v2 = *v4; else *v1 = *v4;
  }
}

/* squared distance from the origin to triangle (a, b, c) and the closest point (Ericson, Real-Time
 * Collision Detection 5.1.5) */
static double origin_tri_closest(const double *a, const double *b, const double *c, double *q) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub(b, a, ab); sub(c, a, ac);
  const double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(q, a, 24); return dot(q, q); }
  double bp[3] = {-b[0], -b[1], -b[2]};
  const double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(q, b, 24); return dot(q, q); }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const double t = d1 / (d1 - d3);
    for (int i = 0; i < 3; i++) q[i] = a[i] + t * ab[i];
    return dot(q, q);
  }
  double cp[3] = {-c[0], -c[1], -c[2]};
  const double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(q, c, 24); return dot(q, q); }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const double t = d2 / (d2 - d6);
    for (int i = 0; i < 3; i++) q[i] = a[i] + t * ac[i];
    return dot(q, q);
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; i++) q[i] = b[i] + t * (c[i] - b[i]);
    return dot(q, q);
  }
  const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
  for (int i = 0; i < 3; i++) q[i] = a[i] + ab[i] * v + ac[i] * w;
  return dot(q, q);
}

static void world_vertex(const OrModel *m, int b, const double *R, const double *p, int iv, double *w) {
  const double *v = m->hull_vert + 3 * (m->hull_adr[b] + iv);
  for (int i = 0; i < 3; i++) w[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2] + p[i];
}

/* Penetration of hull a (geom1) and hull b (geom2).  Returns 1 with depth >= 0, unit dir (from a to b: moving b by
 * depth * dir separates them) and the contact point pos; 0 if the hulls do not intersect (or only touch). */
int or_mpr_penetration(const OrModel *m, int a, const double *Ra, const double *pa, int b, const double *Rb,
                       const double *pb, double *depth, double *dir_out, double *pos) {
  return or_mpr_penetration_margin(m, a, Ra, pa, b, Rb, pb, 0.0, depth, dir_out, pos);
}

int or_mpr_penetration_margin(const OrModel *m, int a, const double *Ra, const double *pa, int b, const double *Rb,
                              const double *pb, double margin, double *depth, double *dir_out, double *pos) {
  const Pair P = {m, a, b, Ra, pa, Rb, pb, margin};
  Sup v0, v1, v2, v3, v4;
  double ca[3], cb[3], dir[3], va[3], vb[3];
  /* interior point: centre of a minus centre of b */
  for (int i = 0; i < 3; i++) {
    ca[i] = Ra[3 * i] * m->hull_center[a][0] + Ra[3 * i + 1] * m->hull_center[a][1] + Ra[3 * i + 2] * m->hull_center[a][2] + pa[i];
    cb[i] = Rb[3 * i] * m->hull_center[b][0] + Rb[3 * i + 1] * m->hull_center[b][1] + Rb[3 * i + 2] * m->hull_center[b][2] + pb[i];
  }
  sub(ca, cb, v0.v);
  v0.ia = v0.ib = -1;
  if (dot(v0.v, v0.v) < 1e-30) { v0.v[0] = 1e-10; v0.v[1] = 0; v0.v[2] = 0; }
  /* ---- portal discovery */
  for (int i = 0; i < 3; i++) dir[i] = -v0.v[i];
  normalize(dir);
  support(&P, dir, &v1);
  if (dot(v1.v, dir) <= 0) return 0;
  cross(v0.v, v1.v, dir);
  if (dot(dir, dir) < 1e-30) {
    /* the origin lies on the ray v0 -> v1: the segment case.  Penetration along that ray. */
    const double d1 = sqrt(dot(v1.v, v1.v));
    if (!(d1 > 0)) return 0;
    *depth = d1;
    for (int i = 0; i < 3; i++) dir_out[i] = v1.v[i] / d1;
    double wa[3], wb[3];
    world_vertex(m, a, Ra, pa, v1.ia, wa);
    world_vertex(m, b, Rb, pb, v1.ib, wb);
    for (int i = 0; i < 3; i++) pos[i] = 0.5 * (wa[i] + wb[i]);
    return 1;
  }
  normalize(dir);
  support(&P, dir, &v2);
  if (dot(v2.v, dir) <= 0) return 0;
  sub(v1.v, v0.v, va); sub(v2.v, v0.v, vb);
  cross(va, vb, dir);
  normalize(dir);
  if (dot(dir, v0.v) > 0) { /* orient the portal normal away from v0 */
    Sup t = v1; v1 = v2; v2 = t;
    for (int i = 0; i < 3; i++) dir[i] = -dir[i];
  }
  for (int it = 0;; it++) {
    if (it > MPR_MAXIT) return 0;
    support(&P, dir, &v3);
    if (dot(v3.v, dir) <= 0) return 0;
    int cont = 0;
    cross(v1.v, v3.v, va);
    if (dot(va, v0.v) < -1e-14) { v2 = v3; cont = 1; }          /* origin outside (v1, v0, v3) */
    if (!cont) {
      cross(v3.v, v2.v, va);
      if (dot(va, v0.v) < -1e-14) { v1 = v3; cont = 1; }        /* origin outside (v3, v0, v2) */
    }
    if (!cont) break;
    sub(v1.v, v0.v, va); sub(v2.v, v0.v, vb);
    cross(va, vb, dir);
    normalize(dir);
  }
  /* ---- portal refinement until the portal reaches the surface of A - B; the origin must be inside it */
  int inside = 0;
  for (int it = 0;; it++) {
    portal_dir(&v1, &v2, &v3, dir);
    if (dot(dir, v1.v) >= 0) inside = 1; /* the origin is on the inner side of the portal: the hulls intersect */
    support(&P, dir, &v4);
    const double d4 = dot(v4.v, dir);
    const double adv = fmin(fmin(d4 - dot(v1.v, dir), d4 - dot(v2.v, dir)), d4 - dot(v3.v, dir));
    if (!inside && d4 < 0) return 0;              /* the origin is beyond the support plane: separated */
    if (adv <= MPR_TOL || it >= MPR_MAXIT) {
      if (!inside) return 0;
      break;
    }
    expand_portal(&v0, &v1, &v2, &v3, &v4);
  }
  /* ---- penetration: closest point of the final portal to the origin */
  double q[3];
  const double d2 = origin_tri_closest(v1.v, v2.v, v3.v, q);
  *depth = sqrt(d2);
  portal_dir(&v1, &v2, &v3, dir);
  if (*depth > 1e-14) for (int i = 0; i < 3; i++) dir_out[i] = q[i] / *depth;
  else memcpy(dir_out, dir, 24);
  /* ---- contact point: barycentric weights of the origin's ray in the portal (libccd findPos) */
  double bw[4], t1[3];
  cross(v1.v, v2.v, t1); bw[0] = dot(t1, v3.v);
  cross(v3.v, v2.v, t1); bw[1] = dot(t1, v0.v);
  cross(v0.v, v1.v, t1); bw[2] = dot(t1, v3.v);
  cross(v2.v, v1.v, t1); bw[3] = dot(t1, v0.v);
  double sum = bw[0] + bw[1] + bw[2] + bw[3];
  if (sum <= 0) {
    bw[0] = 0;
    cross(v2.v, v3.v, t1); bw[1] = dot(t1, dir);
    cross(v3.v, v1.v, t1); bw[2] = dot(t1, dir);
    cross(v1.v, v2.v, t1); bw[3] = dot(t1, dir);
    sum = bw[1] + bw[2] + bw[3];
  }
  const Sup *sv[4] = {&v0, &v1, &v2, &v3};
  double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
  for (int k = 0; k < 4; k++) {
    double wa[3], wb[3];
    if (k == 0) { memcpy(wa, ca, 24); memcpy(wb, cb, 24); }
    else { world_vertex(m, a, Ra, pa, sv[k]->ia, wa); world_vertex(m, b, Rb, pb, sv[k]->ib, wb); }
    for (int i = 0; i < 3; i++) { p1[i] += bw[k] * wa[i]; p2[i] += bw[k] * wb[i]; }
  }
  for (int i = 0; i < 3; i++) pos[i] = 0.5 * (p1[i] + p2[i]) / sum;
  return 1;
}

/* 15-axis separating-axis test of two boxes (centre c, half extents h in frames R, world centres given) */
static int boxes_overlap(const double *Ra, const double *ca, const double *ha, const double *Rb, const double *cb,
                         const double *hb) {
  double Rm[3][3], A[3][3], t[3], d[3];
  sub(cb, ca, d);
  for (int i = 0; i < 3; i++) {
    t[i] = Ra[i] * d[0] + Ra[3 + i] * d[1] + Ra[6 + i] * d[2];
    for (int j = 0; j < 3; j++) {
      Rm[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[i][j] = fabs(Rm[i][j]) + 1e-12;
    }
  }
  for (int i = 0; i < 3; i++)
    if (fabs(t[i]) > ha[i] + A[i][0] * hb[0] + A[i][1] * hb[1] + A[i][2] * hb[2]) return 0;
  for (int j = 0; j < 3; j++)
    if (fabs(t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j]) > ha[0] * A[0][j] + ha[1] * A[1][j] + ha[2] * A[2][j] + hb[j]) return 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = ha[i1] * A[i2][j] + ha[i2] * A[i1][j], rb = hb[j1] * A[i][j2] + hb[j2] * A[i][j1];
      if (fabs(t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j]) > ra + rb) return 0;
    }
  return 1;
}

/* Robot<->robot contacts for one env: every candidate geom pair (mj_pairs order) through bounding spheres, boxes,
 * MPR.  Rb / pb: world placements of the bodies.  Appends at most OR_MAXHH contacts (and never beyond OR_MAXCON in
 * total); returns the number appended; *overflow bit 0: a penetrating pair had to be dropped at a cap, bit 1: more than
 * 64 pairs survived the mid phase (the device works through at most one wavefront's worth of candidates). */
int or_collide_pairs(const OrModel *m, const double Rb[][9], const double pb[][3], int ncon0, int *geom1, int *geom2,
                     double *dist, double (*pos)[3], double (*nrm)[3], int *overflow) {
  int n = 0, ncand = 0;
  const double margin = m->contact[10], hm = 0.5 * margin;
  *overflow = 0;
  for (int k = 0; k < m->npair; k++) {
    const int a = m->pairs[2 * k], b = m->pairs[2 * k + 1];
    const int Ba = m->geom_body[a], Bb = m->geom_body[b];
    double ca[3], cb[3], d[3], ba[3], bb[3];
    for (int i = 0; i < 3; i++) {
      ca[i] = Rb[Ba][3 * i] * m->rbound[a][0] + Rb[Ba][3 * i + 1] * m->rbound[a][1] + Rb[Ba][3 * i + 2] * m->rbound[a][2] + pb[Ba][i];
      cb[i] = Rb[Bb][3 * i] * m->rbound[b][0] + Rb[Bb][3 * i + 1] * m->rbound[b][1] + Rb[Bb][3 * i + 2] * m->rbound[b][2] + pb[Bb][i];
      ba[i] = Rb[Ba][3 * i] * m->hull_box[a][0] + Rb[Ba][3 * i + 1] * m->hull_box[a][1] + Rb[Ba][3 * i + 2] * m->hull_box[a][2] + pb[Ba][i];
      bb[i] = Rb[Bb][3 * i] * m->hull_box[b][0] + Rb[Bb][3 * i + 1] * m->hull_box[b][1] + Rb[Bb][3 * i + 2] * m->hull_box[b][2] + pb[Bb][i];
    }
    sub(ca, cb, d);
    const double rr = m->rbound[a][3] + m->rbound[b][3] + margin;
    if (dot(d, d) > rr * rr) continue;
    const double ha[3] = {m->hull_box[a][3] + hm, m->hull_box[a][4] + hm, m->hull_box[a][5] + hm};
    const double hb[3] = {m->hull_box[b][3] + hm, m->hull_box[b][4] + hm, m->hull_box[b][5] + hm};
    if (!boxes_overlap(Rb[Ba], ba, ha, Rb[Bb], bb, hb)) continue;
    if (++ncand > 64) { *overflow |= 2; break; }
    double depth, dir[3], p[3];
    if (!or_mpr_penetration_margin(m, a, Rb[Ba], pb[Ba], b, Rb[Bb], pb[Bb], margin, &depth, dir, p)) continue;
    if (n >= OR_MAXHH || ncon0 + n >= OR_MAXCON) { *overflow |= 1; continue; }
    geom1[n] = a; geom2[n] = b;
    dist[n] = margin - depth;
    memcpy(pos[n], p, 24);
    memcpy(nrm[n], dir, 24);
    n++;
  }
  return n;
}
