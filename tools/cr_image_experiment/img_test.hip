// standalone check of the CrImg staging / reading helpers with index-coded arrays
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../agglomerationmultigrid1d_amd/csrc/cr_kernels.hpp"
using namespace aggmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int M>
__global__ void k(CrLevel l0, CrLevel l1, CrLevel l2, const double* d0, int64_t b0, double* out) {
  extern __shared__ double sh[];
  char* img = reinterpret_cast<char*>(sh);
  const int lane = threadIdx.x & 63;
  CrLevel lv[3] = {l0, l1, l2};
  using G = CrImg<M>;
  cr_glds<G::kDb>(d0, b0, lane, img + G::oD);
  cr_stage_levels<M, true, 0>(lv, b0, lane, img);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // layout of out per lane: d (8M), then per level I: fe rows 0..NB1 (each 2MM), lu rows (MM each) , perm rows (M each as double)
  double* o = out + (size_t)lane * 512;
  int n = 0;
  double v[19 * M];
  cr_img_d<M>(img, lane, v);
  for (int t = 0; t < 8 * M; ++t) o[n++] = v[t];
#define LEVEL(I)                                                              \
  {                                                                           \
    constexpr int NB1 = 4 >> I;                                               \
    for (int jj = 0; jj <= NB1; ++jj) {                                       \
      double ac[2 * M * M];                                                   \
      cr_img_row<M, I>(img, lane, jj, ac);                                    \
      for (int t = 0; t < 2 * M * M; ++t) o[n++] = ac[t];                     \
    }                                                                         \
    for (int jj = 0; jj < NB1; ++jj) {                                        \
      double f[M * M];                                                        \
      int32_t pm[M];                                                          \
      cr_img_lu<M, I>(img, lane, jj, f, pm);                                  \
      for (int t = 0; t < M * M; ++t) o[n++] = f[t];                          \
      for (int t = 0; t < M; ++t) o[n++] = (double)pm[t];                     \
    }                                                                         \
  }
  LEVEL(0) LEVEL(1) LEVEL(2)
  o[511] = n;
}

template <int M>
int run() {
  const int64_t b0 = 128;          // wave's first sub-chunk
  const int64_t nsub = b0 + 64 + 2;
  std::vector<double> d(nsub * 8 * M);
  for (size_t i = 0; i < d.size(); ++i) d[i] = 1e6 + i;
  std::vector<double> fe[3], lu[3];
  std::vector<int32_t> pm[3];
  CrLevel L[3];
  double* dd;
  CK(hipMalloc(&dd, d.size() * 8));
  CK(hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  for (int I = 0; I < 3; ++I) {
    const int64_t rows = nsub * (4 >> I) + 1;
    fe[I].resize(rows * 2 * M * M), lu[I].resize(rows * M * M), pm[I].resize(rows * M);
    for (size_t i = 0; i < fe[I].size(); ++i) fe[I][i] = (I + 1) * 1e7 + i;
    for (size_t i = 0; i < lu[I].size(); ++i) lu[I][i] = -((I + 1) * 1e7 + i);
    for (size_t i = 0; i < pm[I].size(); ++i) pm[I][i] = (int32_t)(i % 1000);
    double *a, *b;
    int32_t* c;
    CK(hipMalloc(&a, fe[I].size() * 8)); CK(hipMalloc(&b, lu[I].size() * 8)); CK(hipMalloc(&c, pm[I].size() * 4));
    CK(hipMemcpy(a, fe[I].data(), fe[I].size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, lu[I].data(), lu[I].size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, pm[I].data(), pm[I].size() * 4, hipMemcpyHostToDevice));
    L[I].fe = a, L[I].fo = a, L[I].lu = b, L[I].perm = c;
    L[I].n = 1 << 30, L[I].n_even = rows, L[I].n_odd = rows;
  }
  double* out;
  CK(hipMalloc(&out, 64 * 512 * 8));
  CK(hipMemset(out, 0, 64 * 512 * 8));
  hipLaunchKernelGGL((k<M>), dim3(1), dim3(64), CrImg<M>::kBytes, 0, L[0], L[1], L[2], dd, b0, out);
  CK(hipDeviceSynchronize());
  std::vector<double> h(64 * 512);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const double* o = h.data() + lane * 512;
    int n = 0;
    const int64_t b = b0 + lane;
    auto chk = [&](double got, double want, const char* what, int I, int jj, int t) {
      if (got != want && bad++ < 12) printf("M=%d lane %d %s I=%d jj=%d t=%d got %.1f want %.1f\n", M, lane, what, I, jj, t, got, want);
    };
    for (int t = 0; t < 8 * M; ++t) chk(o[n++], d[b * 8 * M + t], "d", 0, 0, t);
    for (int I = 0; I < 3; ++I) {
      const int NB1 = 4 >> I;
      for (int jj = 0; jj <= NB1; ++jj)
        for (int t = 0; t < 2 * M * M; ++t) chk(o[n++], fe[I][(b * NB1 + jj) * 2 * M * M + t], "fe", I, jj, t);
      for (int jj = 0; jj < NB1; ++jj) {
        for (int t = 0; t < M * M; ++t) chk(o[n++], lu[I][(b * NB1 + jj) * M * M + t], "lu", I, jj, t);
        for (int t = 0; t < M; ++t) {
          const double want = M == 1 ? 0.0 : (double)pm[I][(b * NB1 + jj) * M + t];
          chk(o[n++], want, "perm", I, jj, t);
        }
      }
    }
  }
  printf("M=%d: %d mismatches, image bytes %d\n", M, bad, CrImg<M>::kBytes);
  return 0;
}

int main() { return run<1>() | run<2>(); }
