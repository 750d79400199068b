// rect_overlap.hpp -- exact overlap area of two rotated rectangles in the plane, shared by
// iou_stage.hip (IoU matrices, rotated NMS) and eval_stage.hip (the AP evaluator's per-frame overlaps).
// See iou_stage.hip's header for the algorithm and how it differs from the reference's.
#pragma once
#include "common.hpp"

namespace {

constexpr int IB = 256;                 // threads per workgroup of every kernel that calls overlap_area (the LDS stride)
constexpr int MAXV = 8;                 // a quadrilateral clipped by four half-planes has at most 8 vertices

struct Rect {                           // centre, half extents along its own axes, first axis (unit vector)
  float cx, cy, hu, hv, ax, ay;
};

// fmt 7: (x, y, z, dx, dy, dz, heading), heading turns the box counter-clockwise (iou3d_nms convention);
// fmt 5: (cx, cy, w, h, angle) of rotate_iou.py, whose corner formula turns the box CLOCKWISE by `angle`
__device__ __forceinline__ Rect make_rect(const float *b, int fmt) {
  Rect r;
  r.cx = b[0]; r.cy = b[1];
  float s, c;
  if (fmt == 7) {
    r.hu = 0.5f * b[3]; r.hv = 0.5f * b[4];
    sincosf(b[6], &s, &c);
    r.ax = c; r.ay = s;
  } else {
    r.hu = 0.5f * b[2]; r.hv = 0.5f * b[3];
    sincosf(b[4], &s, &c);
    r.ax = c; r.ay = -s;
  }
  return r;
}

__device__ __forceinline__ float rect_area(const Rect &r) { return 4.0f * r.hu * r.hv; }

// Area of A n B.  pu / pv: this thread's column of the two [MAXV][IB] LDS planes (ping) and qu / qv (pong).
__device__ inline float overlap_area(const Rect &A, const Rect &B, float *pu, float *pv, float *qu, float *qv) {
  const float dx = A.cx - B.cx, dy = A.cy - B.cy;
  const float ra2 = A.hu * A.hu + A.hv * A.hv, rb2 = B.hu * B.hu + B.hv * B.hv;
  const float rr = ra2 + rb2 + 2.0f * sqrtf(ra2 * rb2);              // (ra + rb)^2
  if (dx * dx + dy * dy > rr) return 0.0f;                            // circumscribed circles apart
  // A's centre and axes in B's frame
  const float cu = dx * B.ax + dy * B.ay, cv = -dx * B.ay + dy * B.ax;
  const float eu = A.ax * B.ax + A.ay * B.ay, ev = -A.ax * B.ay + A.ay * B.ax;     // A's first axis
  const float au = A.hu * eu, av = A.hu * ev, bu = -A.hv * ev, bv = A.hv * eu;      // half edges
  // counter-clockwise corners
  pu[0 * IB] = cu + au + bu; pv[0 * IB] = cv + av + bv;
  pu[1 * IB] = cu - au + bu; pv[1 * IB] = cv - av + bv;
  pu[2 * IB] = cu - au - bu; pv[2 * IB] = cv - av - bv;
  pu[3 * IB] = cu + au - bu; pv[3 * IB] = cv + av - bv;
  int n = 4;
  // four half-planes g(p) >= 0:  hu - u,  hu + u,  hv - v,  hv + v
#pragma unroll
  for (int side = 0; side < 4; side++) {
    const float lim = (side < 2) ? B.hu : B.hv;
    const float sg = (side & 1) ? 1.0f : -1.0f;
    const float *iu = (side & 1) ? qu : pu, *iv = (side & 1) ? qv : pv;
    float *ou = (side & 1) ? pu : qu, *ov = (side & 1) ? pv : qv;
    const float *ic = (side < 2) ? iu : iv;                            // the coordinate this side bounds
    int m = 0;
    float u0 = iu[0], v0 = iv[0];
    float g0 = lim + sg * ic[0];
    for (int k = 0; k < n; k++) {
      const int kn = (k + 1 == n) ? 0 : k + 1;
      const float u1 = iu[kn * IB], v1 = iv[kn * IB];
      const float g1 = lim + sg * ic[kn * IB];
      if (g0 >= 0.0f) { ou[m * IB] = u0; ov[m * IB] = v0; m++; }
      if ((g0 >= 0.0f) != (g1 >= 0.0f)) {                             // the edge crosses the side
        const float t = g0 / (g0 - g1);
        ou[m * IB] = u0 + t * (u1 - u0);
        ov[m * IB] = v0 + t * (v1 - v0);
        m++;
      }
      u0 = u1; v0 = v1; g0 = g1;
    }
    n = m;
    if (n < 3) return 0.0f;
  }
  // after four sides the polygon is back in (pu, pv); shoelace around its first vertex
  const float ox = pu[0], oy = pv[0];
  float twice = 0.0f;
  float x0 = pu[1 * IB] - ox, y0 = pv[1 * IB] - oy;
  for (int k = 2; k < n; k++) {
    const float x1 = pu[k * IB] - ox, y1 = pv[k * IB] - oy;
    twice += x0 * y1 - x1 * y0;
    x0 = x1; y0 = y1;
  }
  return 0.5f * fabsf(twice);
}

// what to make of the overlap: 0 overlap area, 1 IoU, 2 overlap / area(A), 3 overlap / area(B)
__device__ __forceinline__ float criterion_value(float inter, const Rect &A, const Rect &B, int crit) {
  if (crit == 0) return inter;
  const float sa = rect_area(A), sb = rect_area(B);
  if (crit == 1) return inter / fmaxf(sa + sb - inter, 1e-8f);
  return inter / fmaxf(crit == 2 ? sa : sb, 1e-8f);
}

}  // namespace
