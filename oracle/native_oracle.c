/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the two native ops on MEGA's hot path.
 * Never linked into or called by the product package; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * Pinned against: tests/test_nms.py:16-58 and :60-217 of the reference (golden keep sets) and the
 * reference's own compiled CPU ops (oracle/_ref, see build_ref.py) on seeded inputs.
 *
 *   oracle_nms        follows mega_core/csrc/cpu/nms_cpu.cpp:5-65 (stable descending score order,
 *                     +1 area convention, serial greedy suppression, result = ascending original
 *                     indices) with the comparison as a flag: strict_gt = 0 -> IoU >= thr
 *                     (nms_cpu.cpp:60), strict_gt = 1 -> IoU > thr (csrc/cuda/nms.cu:60, devIoU :13-21).
 *   oracle_roi_align  follows mega_core/csrc/cpu/ROIAlign_cpu.cpp:18-111 (pre_calc_for_bilinear_interpolate)
 *                     and :113-219 (ROIAlignForward_cpu_kernel); input NCHW, output [K][C][ph][pw].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int i; } si_t;

static int cmp_desc(const void* a, const void* b) {
  const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i - y->i; /* ties: lower index first (stable) */
}

/* dets [n][4], scores [n]; keep_out [n] receives ascending original indices; returns the count. */
int oracle_nms(const float* dets, const float* scores, int n, float thr, int strict_gt, long long* keep_out) {
  if (n <= 0) return 0;
  si_t* ord = (si_t*)malloc(sizeof(si_t) * n);
  unsigned char* sup = (unsigned char*)calloc(n, 1);
  float* area = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < n; ++i) {
    ord[i].s = scores[i]; ord[i].i = i;
    area[i] = (dets[4 * i + 2] - dets[4 * i + 0] + 1) * (dets[4 * i + 3] - dets[4 * i + 1] + 1);
  }
  qsort(ord, n, sizeof(si_t), cmp_desc);
  for (int _i = 0; _i < n; ++_i) {
    const int i = ord[_i].i;
    if (sup[i]) continue;
    const float ix1 = dets[4 * i], iy1 = dets[4 * i + 1], ix2 = dets[4 * i + 2], iy2 = dets[4 * i + 3];
    const float iarea = area[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      const int j = ord[_j].i;
      if (sup[j]) continue;
      const float xx1 = fmaxf(ix1, dets[4 * j]), yy1 = fmaxf(iy1, dets[4 * j + 1]);
      const float xx2 = fminf(ix2, dets[4 * j + 2]), yy2 = fminf(iy2, dets[4 * j + 3]);
      const float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
      const float inter = w * h;
      const float ovr = inter / (iarea + area[j] - inter);
      if (strict_gt ? (ovr > thr) : (ovr >= thr)) sup[j] = 1;
    }
  }
  int cnt = 0;
  for (int i = 0; i < n; ++i) if (!sup[i]) keep_out[cnt++] = i;
  free(ord); free(sup); free(area);
  return cnt;
}

/* feat [B][C][H][W], rois [K][5] -> out [K][C][ph][pw] */
void oracle_roi_align(const float* feat, const float* rois, float* out, int K, int C, int H, int W,
                      float spatial_scale, int PH, int PW, int sampling_ratio) {
  for (int n = 0; n < K; ++n) {
    const float* r = rois + 5 * n;
    const int b = (int)r[0];
    const float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
    const float roi_end_w = r[3] * spatial_scale, roi_end_h = r[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    const float bin_size_h = roi_height / (float)PH, bin_size_w = roi_width / (float)PW;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / PH);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / PW);
    const float count = (float)(gh * gw);
    const int npc = PH * PW * gh * gw;
    int* pos = (int*)malloc(sizeof(int) * 4 * npc);
    float* wgt = (float*)malloc(sizeof(float) * 4 * npc);
    int idx = 0;
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int iy = 0; iy < gh; ++iy) {
          const float yy = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / (float)gh;
          for (int ix = 0; ix < gw; ++ix, ++idx) {
            const float xx = roi_start_w + pw * bin_size_w + (ix + .5f) * bin_size_w / (float)gw;
            float x = xx, y = yy;
            if (y < -1.0 || y > H || x < -1.0 || x > W) {
              for (int t = 0; t < 4; ++t) { pos[4 * idx + t] = 0; wgt[4 * idx + t] = 0.f; }
              continue;
            }
            if (y <= 0) y = 0;
            if (x <= 0) x = 0;
            int y_low = (int)y, x_low = (int)x, y_high, x_high;
            if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
            if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
            const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
            pos[4 * idx + 0] = y_low * W + x_low;  wgt[4 * idx + 0] = hy * hx;
            pos[4 * idx + 1] = y_low * W + x_high; wgt[4 * idx + 1] = hy * lx;
            pos[4 * idx + 2] = y_high * W + x_low; wgt[4 * idx + 2] = ly * hx;
            pos[4 * idx + 3] = y_high * W + x_high; wgt[4 * idx + 3] = ly * lx;
          }
        }
    for (int c = 0; c < C; ++c) {
      const float* src = feat + ((size_t)b * C + c) * H * W;
      int pc = 0;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          float v = 0.f;
          for (int s = 0; s < gh * gw; ++s, ++pc)
            v += wgt[4 * pc] * src[pos[4 * pc]] + wgt[4 * pc + 1] * src[pos[4 * pc + 1]] +
                 wgt[4 * pc + 2] * src[pos[4 * pc + 2]] + wgt[4 * pc + 3] * src[pos[4 * pc + 3]];
          out[(((size_t)n * C + c) * PH + ph) * PW + pw] = v / count;
        }
    }
    free(pos); free(wgt);
  }
}
