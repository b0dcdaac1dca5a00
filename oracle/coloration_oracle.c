/*
 * coloration_oracle.c -- CPU restatement of the reference's MeshColoration pass.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as tsdf_oracle.c): only tests/ and bench/smoke checkers may use it.
 * PARITY UNPINNED: the reference ships no tests or fixtures for this pass either, and it runs on VTK
 * (vtkTransform, vtkPolyData), which this image lacks.  Citations:
 *   MC.cxx = Coloration/MeshColoration.cxx, RD.cxx = Sources/ReconstructionData.cxx, Helper.h = Sources/Helper.h.
 * Third-party arithmetic on the path: vtkTransform::TransformPoint / TransformVector (VTK, version not pinned by
 * the reference's CMake; the 6.x-8.x implementation is restated): for a vtkTransform whose matrix was set
 * with SetMatrix, TransformPoint evaluates  out[i] = M[i][0]*x + M[i][1]*y + M[i][2]*z + M[i][3]  and
 * TransformVector  out[i] = M[i][0]*x + M[i][1]*y + M[i][2]*z,  left to right in double
 * (vtkLinearTransform.cxx: vtkLinearTransformPoint / vtkLinearTransformVector).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* RD.cxx:169-182 TransformWorldToDepthMapPosition.  No test of the sign of z and no depth test.
 * double -> int of round(): NaN, infinities and |x| >= 2^31 are "outside" (project rule, as tsdf_oracle.c). */
static int world_to_pixel(const double RT[16], const double K4[16], const double w[3], int pix[2])
{
  double c[3], d[3];
  for (int i = 0; i < 3; ++i)
    c[i] = RT[4 * i + 0] * w[0] + RT[4 * i + 1] * w[1] + RT[4 * i + 2] * w[2] + RT[4 * i + 3]; /* TransformPoint, RD.cxx:173 */
  for (int i = 0; i < 3; ++i)
    d[i] = K4[4 * i + 0] * c[0] + K4[4 * i + 1] * c[1] + K4[4 * i + 2] * c[2]; /* TransformVector, RD.cxx:175 */
  d[0] = d[0] / d[2]; /* RD.cxx:177 */
  d[1] = d[1] / d[2]; /* RD.cxx:178 */
  double ru = round(d[0]), rv = round(d[1]); /* RD.cxx:180-181 */
  if (!(ru > -2147483648.0 && ru < 2147483648.0) || !(rv > -2147483648.0 && rv < 2147483648.0))
    return 0;
  pix[0] = (int)ru;
  pix[1] = (int)rv;
  return 1;
}

static int cmp_double(const void *a, const void *b)
{
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* Helper.h:174-187 ComputeMedian */
static double compute_median(double *v, size_t n)
{
  qsort(v, n, sizeof(double), cmp_double);
  size_t mid = n / 2;
  if (n % 2 == 0)
    return (v[mid] + v[mid - 1]) / 2;
  return v[mid];
}

/* MC.cxx:98-199 ProcessColoration.
 *   points [nv][3] f64; colors [n][H][W][3] u8, vtk point order (row 0 = bottom, RD.cxx:106-108);
 *   K4, RT4 [n][16] row-major; mean, median [nv][3] u8; count [nv] i32.  Outputs start at 0 (MC.cxx:113-133). */
void oracle_color_mesh(const double *points, int64_t nv, const uint8_t *colors, const double *K4, const double *RT4,
                       int n, int W, int H, uint8_t *mean, uint8_t *median, int32_t *count)
{
  double *l0 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double *l1 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double *l2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  memset(mean, 0, (size_t)nv * 3);
  memset(median, 0, (size_t)nv * 3);
  memset(count, 0, (size_t)nv * sizeof(int32_t));
  for (int64_t id = 0; id < nv; ++id) { /* MC.cxx:140 */
    size_t cnt = 0;
    for (int m = 0; m < n; ++m) { /* MC.cxx:150 */
      int pix[2];
      if (!world_to_pixel(RT4 + 16 * m, K4 + 16 * m, points + 3 * id, pix))
        continue;
      if (pix[0] < 0 || pix[1] < 0 || pix[0] >= W || pix[1] >= H) /* MC.cxx:158-163 */
        continue;
      /* RD.cxx:92-116 GetColorValue: pixel (x, H-1-y) of the vtk image */
      const uint8_t *c = colors + (((size_t)m * H + (size_t)(H - 1 - pix[1])) * W + (size_t)pix[0]) * 3;
      l0[cnt] = c[0];
      l1[cnt] = c[1];
      l2[cnt] = c[2];
      ++cnt;
    }
    if (cnt != 0) { /* MC.cxx:174 */
      /* std::accumulate(begin, end, 0): the init value is an int, so the running sum is an int (MC.cxx:176-178) */
      int s0 = 0, s1 = 0, s2 = 0;
      for (size_t i = 0; i < cnt; ++i) {
        s0 = (int)(s0 + l0[i]);
        s1 = (int)(s1 + l1[i]);
        s2 = (int)(s2 + l2[i]);
      }
      double nb = (double)cnt;
      /* SetTuple3 on a vtkUnsignedCharArray converts each double with static_cast<unsigned char> (MC.cxx:180) */
      mean[3 * id + 0] = (uint8_t)((double)s0 / nb);
      mean[3 * id + 1] = (uint8_t)((double)s1 / nb);
      mean[3 * id + 2] = (uint8_t)((double)s2 / nb);
      median[3 * id + 0] = (uint8_t)compute_median(l0, cnt); /* MC.cxx:181-185 */
      median[3 * id + 1] = (uint8_t)compute_median(l1, cnt);
      median[3 * id + 2] = (uint8_t)compute_median(l2, cnt);
      count[id] = (int32_t)cnt; /* MC.cxx:186 */
    }
  }
  free(l0);
  free(l1);
  free(l2);
}
