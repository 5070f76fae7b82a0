// step 0 forward through the LDS image against the register path, same synthetic system (one chunk, q = 9)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdint>
#include <cmath>
#include "../../agglomerationmultigrid1d_amd/csrc/cr_kernels.hpp"
using namespace aggmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int M, bool IMG>
__global__ __launch_bounds__(256) void k(CrStageArgs A, const double* d0, double* out) {
  extern __shared__ double sh[];
  for (int t = threadIdx.x; t < A.lds_total; t += blockDim.x) sh[t] = 0.0;
  __syncthreads();
  const int64_t c = blockIdx.x;
  const bool wg_shared = ((c + 1) << A.q) <= A.lv[0].n - 1;
  if (IMG) cr_step0_forward_img<M>(A, c, wg_shared, d0, sh, cr_wave_image<M>(A, sh));
  else cr_step_forward<M, 3>(A, 0, c, wg_shared, d0, nullptr, sh);
  __syncthreads();
  const int cnt = ((1 << (A.q - 3)) + 1) * M;
  for (int t = threadIdx.x; t < 2 * cnt; t += blockDim.x) out[c * 2 * cnt + t] = sh[A.lds_off[1] + t];
}

template <int M>
__global__ __launch_bounds__(64) void kdbg(CrStageArgs A, const double* d0, double* out) {
  extern __shared__ double sh[];
  const int lane = threadIdx.x & 63;
  using G = CrImg<M>;
  char* img = cr_wave_image<M>(A, sh);
  const int64_t b0 = 64;
  cr_glds<G::kDb>(d0, b0, lane, img + G::oD);
  cr_stage_levels<M, true, 0>(&A.lv[0], b0, lane, img);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double v[19 * M], w[19 * M];
  cr_img_d<M>(img, lane, v);
  for (int e = 0; e < M; ++e) v[8 * M + e] = 0.0;
  cr_fwd_img<M, 0>(img, lane, v);
  const CrStepGeom g = cr_step_geom<3>(A, 0, 1, true);
  int64_t thi; bool tsh;
  cr_loc_load<M, 3>(A, 0, g, b0 + lane, d0, nullptr, sh, w, thi, tsh, false);
  cr_loc_fwd<M, 3, 0>(&A.lv[0], b0 + lane, w);
  for (int t = 0; t < 19 * M; ++t) { out[(lane * 2 + 0) * 40 + t] = v[t]; out[(lane * 2 + 1) * 40 + t] = w[t]; }
}

template <int M>
int run(int threads) {
  const int q = 9;
  const int64_t nchunks = 3;
  const int64_t n = (nchunks << q) + 1;   // last block shared => every group interior
  CrStageArgs A;
  std::memset(&A, 0, sizeof(A));
  A.q = q;
  A.nsteps = 3;
  A.step_a[0] = 0, A.step_a[1] = 3, A.step_a[2] = 6, A.step_a[3] = 9;
  int o = 0;
  for (int s = 1; s <= 3; ++s) { const int cnt = ((1 << (q - A.step_a[s])) + 1) * M; A.lds_off[s] = o; o += 2 * cnt; }
  for (int s = 1; s <= 3; ++s) { const int cnt = ((1 << (q - A.step_a[s])) + 1) * M; A.lds_xoff[s] = o; o += cnt; }
  A.lds_total = o;
  A.n_out = nchunks + 1;
  int64_t nl = n;
  std::vector<void*> keep;
  for (int l = 0; l < 3; ++l) {
    const int64_t ne = (nl + 1) / 2, no = nl / 2;
    std::vector<double> fe(ne * 2 * M * M), fo(no * 2 * M * M), lu(no * M * M);
    std::vector<int32_t> pm(no * M);
    for (auto& v : fe) v = 0.3 * std::sin(1.0 + (&v - fe.data()) * 0.37 + l);
    for (auto& v : fo) v = 0.3 * std::cos(2.0 + (&v - fo.data()) * 0.21 + l);
    for (int64_t r = 0; r < no; ++r) {
      for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) lu[(r * M + i) * M + j] = (i == j ? 3.0 + 0.1 * std::sin(r + l) : (i > j ? 0.25 : 0.5));
      for (int i = 0; i < M; ++i) pm[r * M + i] = (M == 2 && r % 3 == 0) ? 1 - i : i;
    }
    double *a, *b, *c_;
    int32_t* p;
    CK(hipMalloc(&a, fe.size() * 8)); CK(hipMalloc(&b, std::max<size_t>(fo.size(), 1) * 8)); CK(hipMalloc(&c_, std::max<size_t>(lu.size(), 1) * 8));
    CK(hipMalloc(&p, std::max<size_t>(pm.size(), 1) * 4));
    CK(hipMemcpy(a, fe.data(), fe.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, fo.data(), fo.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c_, lu.data(), lu.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice));
    A.lv[l].fe = a, A.lv[l].fo = b, A.lv[l].lu = c_, A.lv[l].perm = p;
    A.lv[l].n = nl, A.lv[l].n_even = ne, A.lv[l].n_odd = no;
    nl = ne;
  }
  for (int l = 3; l < q; ++l) { A.lv[l] = A.lv[2]; A.lv[l].n = nl; nl = (nl + 1) / 2; }
  std::vector<double> d(n * M);
  for (size_t i = 0; i < d.size(); ++i) d[i] = std::sin(0.001 * i) + 0.5;
  double* dd;
  CK(hipMalloc(&dd, d.size() * 8));
  CK(hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice));
  const int64_t nsub = (n >> 3) + 128;
  double* st0;
  CK(hipMalloc(&st0, 3 * M * nsub * 8));
  A.stack0 = st0, A.stack0_stride = nsub;
  A.img = 3;
  const int cnt = ((1 << (q - 3)) + 1) * M;
  double *o1, *o2;
  CK(hipMalloc(&o1, nchunks * 2 * cnt * 8)); CK(hipMalloc(&o2, nchunks * 2 * cnt * 8));
  size_t lds = (((size_t)A.lds_total * 8 + 1023) & ~(size_t)1023) + (threads / 64) * CrImg<M>::kBytes;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<M, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((k<M, true>), dim3(nchunks), dim3(threads), lds, 0, A, dd, o1);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<M, false>), dim3(nchunks), dim3(threads), (size_t)A.lds_total * 8, 0, A, dd, o2);
  CK(hipDeviceSynchronize());
  if (M == 1 && threads == 64) {
    double* od;
    CK(hipMalloc(&od, 64 * 2 * 40 * 8));
    hipLaunchKernelGGL((kdbg<M>), dim3(1), dim3(64), lds, 0, A, dd, od);
    CK(hipDeviceSynchronize());
    std::vector<double> hd(64 * 2 * 40);
    CK(hipMemcpy(hd.data(), od, hd.size() * 8, hipMemcpyDeviceToHost));
    for (int lane : {0, 1, 63}) {
      printf("lane %d:\n", lane);
      for (int t = 0; t < 19 * M; ++t) printf("   v[%2d] img %.6f reg %.6f %s\n", t, hd[(lane * 2) * 40 + t], hd[(lane * 2 + 1) * 40 + t], hd[(lane * 2) * 40 + t] == hd[(lane * 2 + 1) * 40 + t] ? "" : "<<<");
    }
  }
  std::vector<double> h1(nchunks * 2 * cnt), h2(nchunks * 2 * cnt);
  CK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  double mx = 0;
  for (size_t i = 0; i < h1.size(); ++i) {
    const double e = std::fabs(h1[i] - h2[i]);
    if (!(e <= 1e-13 * (1 + std::fabs(h2[i])))) { if (bad++ < 8) printf("  M=%d i=%zu (chunk %zu pos %zu) img %.17g reg %.17g\n", M, i, i / (2 * cnt), i % (2 * cnt), h1[i], h2[i]); }
    if (e == e) mx = std::max(mx, e);
  }
  printf("M=%d threads=%d lds=%zu: %d of %zu differ, max |diff| %.3e\n", M, threads, lds, bad, h1.size(), mx);
  return 0;
}

int main() { return run<2>(128) | run<1>(64) | run<1>(256); }
