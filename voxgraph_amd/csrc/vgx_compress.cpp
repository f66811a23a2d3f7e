// vgx_reg_compress_normal: 9x9 symmetric eigen-decomposition (cyclic Jacobi) turning a
// constraint's normal block into an equivalent 9-residual block.  Host only.
#include <cmath>
#include <cstring>

#include "voxgraph_amd.h"

extern "C" int vgx_reg_compress_normal(const double normal[45], double residuals9[9],
                                       double jacobian9x8[72]) {
  if (!normal || !residuals9 || !jacobian9x8) return VGX_ERR_INVALID;
  // N = [J^T J, J^T r; r^T J, r^T r]
  double A[9][9], V[9][9];
  int k = 9;
  for (int i = 0; i < 8; ++i)
    for (int j = i; j < 8; ++j) A[i][j] = A[j][i] = normal[k++];
  for (int i = 0; i < 8; ++i) A[i][8] = A[8][i] = normal[1 + i];
  A[8][8] = normal[0];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      if (!std::isfinite(A[i][j])) return VGX_ERR_INVALID;
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < 9; ++i) {
      diag += A[i][i] * A[i][i];
      for (int j = i + 1; j < 9; ++j) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-30 * diag || off == 0) break;
    for (int p = 0; p < 8; ++p)
      for (int q = p + 1; q < 9; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int i = 0; i < 9; ++i) {  // A <- A G
          double aip = A[i][p], aiq = A[i][q];
          A[i][p] = c * aip - s * aiq;
          A[i][q] = s * aip + c * aiq;
        }
        for (int i = 0; i < 9; ++i) {  // A <- G^T A
          double api = A[p][i], aqi = A[q][i];
          A[p][i] = c * api - s * aqi;
          A[q][i] = s * api + c * aqi;
        }
        for (int i = 0; i < 9; ++i) {  // V <- V G
          double vip = V[i][p], viq = V[i][q];
          V[i][p] = c * vip - s * viq;
          V[i][q] = s * vip + c * viq;
        }
      }
  }
  // rows of sqrt(L) V^T
  for (int m = 0; m < 9; ++m) {
    double l = A[m][m] > 0 ? std::sqrt(A[m][m]) : 0.0;
    for (int j = 0; j < 8; ++j) jacobian9x8[8 * m + j] = l * V[j][m];
    residuals9[m] = l * V[8][m];
  }
  return VGX_OK;
}
