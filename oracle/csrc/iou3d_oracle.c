/* iou3d_oracle.c -- CPU side of the f-3 checker (TEST INFRASTRUCTURE ONLY; see oracle/iou3d_oracle.py).
 * The arithmetic is the shared text dfu3d_amd/csrc/iou_common.inc compiled by gcc with glibc's libm. */
#include <math.h>
#include <string.h>
#define IOU_FN static inline
#include "../../dfu3d_amd/csrc/iou_common.inc"

/* mode 0: overlap area (boxes_overlap_bev), 1: BEV IoU (boxes_iou_bev) -- (n,m) row-major */
void orc_boxes_bev(const float *a, int n, const float *b, int m, float *out, int mode) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++)
      out[(long)i * m + j] = mode ? iou_bev(a + 7 * i, b + 7 * j) : iou_box_overlap(a + 7 * i, b + 7 * j);
}

/* greedy NMS over boxes already sorted by score (iou3d_nms.cpp:139-177 with the mask of
 * iou3d_nms_kernel.cu:295-339): box i is kept unless an earlier kept box j has iou_bev(j, i) > thresh */
int orc_nms(const float *boxes, int n, float thresh, long long *keep) {
  int nk = 0;
  char *removed = (char *)__builtin_alloca(n > 0 ? n : 1);
  memset(removed, 0, n > 0 ? n : 1);
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep[nk++] = i;
    for (int j = i + 1; j < n; j++)
      if (!removed[j] && iou_bev(boxes + 7 * i, boxes + 7 * j) > thresh) removed[j] = 1;
  }
  return nk;
}
