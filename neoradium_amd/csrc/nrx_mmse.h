// Per-RE MMSE solve shared by the equaliser kernel and the fused estimate+equalise kernel (gfx950).
#pragma once
#include "nrx_cplx.h"

namespace nrx {

// grid.py:669-688:  Ainv = (H^H H + nv I)^-1,  xhat = Ainv H^H y,  scale = 1 / Re diag(Ainv).
// The reference goes through pinv/SVD of the same Hermitian positive-definite matrix; here it is a Cholesky
// factorisation in float64, everything in registers (NR, NL compile-time).
template <int NR, int NL>
__device__ __forceinline__ void mmse_solve(const cx<double> (&H)[NR][NL], const cx<double> (&y)[NR], double nv,
                                           cx<double> (&xhat)[NL], double (&scale)[NL]) {
  using cd = cx<double>;
  // A = H^H H + nv I (lower triangle), z = H^H y
  cd A[NL][NL], z[NL];
#pragma unroll
  for (int p = 0; p < NL; ++p) {
    cd zz(0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) nrx::cmacc(zz, H[r][p], y[r]);
    z[p] = zz;
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      cd a(0, 0);
#pragma unroll
      for (int r = 0; r < NR; ++r) nrx::cmacc(a, H[r][q], H[r][p]);  // conj(H[r][q]) * H[r][p] = A[q][p]
      A[p][q] = nrx::conj(a);                                         // store A[p][q] = conj(A[q][p])
    }
    A[p][p].re += nv;
  }
  // Cholesky A = L L^H (L lower, real positive diagonal)
  cd Lm[NL][NL];
  double dinv[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    double d = A[j][j].re;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= nrx::norm2(Lm[j][k]);
    const double ljj = sqrt(d);
    dinv[j] = 1.0 / ljj;
    Lm[j][j] = cd(ljj, 0);
#pragma unroll
    for (int r = j + 1; r < NL; ++r) {
      cd s = A[r][j];
#pragma unroll
      for (int k = 0; k < j; ++k) {  // s -= L[r][k] * conj(L[j][k])
        s.re -= Lm[r][k].re * Lm[j][k].re + Lm[r][k].im * Lm[j][k].im;
        s.im -= Lm[r][k].im * Lm[j][k].re - Lm[r][k].re * Lm[j][k].im;
      }
      Lm[r][j] = s * dinv[j];
    }
  }
  // M = L^-1 (lower)
  cd M[NL][NL];
#pragma unroll
  for (int c = 0; c < NL; ++c) {
    M[c][c] = cd(dinv[c], 0);
#pragma unroll
    for (int r = c + 1; r < NL; ++r) {
      cd s(0, 0);
#pragma unroll
      for (int k = c; k < r; ++k) nrx::cmac(s, Lm[r][k], M[k][c]);
      M[r][c] = cd(-s.re * dinv[r], -s.im * dinv[r]);
    }
  }
  // xhat = M^H (M z);  diag(Ainv)_p = sum_{k>=p} |M[k][p]|^2
  cd u[NL];
#pragma unroll
  for (int r = 0; r < NL; ++r) {
    cd s(0, 0);
#pragma unroll
    for (int c = 0; c <= r; ++c) nrx::cmac(s, M[r][c], z[c]);
    u[r] = s;
  }
#pragma unroll
  for (int p = 0; p < NL; ++p) {
    cd s(0, 0);
    double dg = 0;
#pragma unroll
    for (int k = p; k < NL; ++k) {
      nrx::cmacc(s, M[k][p], u[k]);
      dg += nrx::norm2(M[k][p]);
    }
    xhat[p] = s;
    scale[p] = 1.0 / dg;
  }
}

}  // namespace nrx
