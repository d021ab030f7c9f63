// Developer tool: the single-reduction coarse CG (csrc/evomg_cg1.h) on its own — microseconds per iteration for a given
// grouping of the waves into workgroups, next to the classical register CG (evomg_cg_reg2d), and a bit-for-bit check of
// the kernel against a host restatement of its recurrence and summation order.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ievostencils_amd/csrc -o tools/bench_cg1 tools/bench_cg1.hip
// run:   tools/bench_cg1 [level=7]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "evomg_cg1.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double tree(std::vector<double> v, int width) {
  for (int off = width / 2; off > 0; off >>= 1)
    for (int l = 0; l < off; ++l) v[l] = v[l] + v[l + off];
  return v[0];
}
struct HostCg1 {
  int n, m, pitch, R, E, H;
  double c[5];
  size_t idx(int ix, int iy) const { return (size_t)(ix + 1) + (size_t)pitch * (iy + 1); }
  double dot(const std::vector<double>& a, const std::vector<double>& b) const {
    std::vector<double> waves(32, 0.0);
    const int C = 64 / H, bands = (m + R - 1) / R, nw = (bands + H - 1) / H;
    for (int gw = 0; gw < nw; ++gw) {
      std::vector<double> lanes(64, 0.0);
      for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int i = 0; i < R; ++i)
          for (int e = 0; e < E; ++e) {
            const int ix = E * (l % C) + e, iy = (gw * H + l / C) * R + i;
            if (ix < m && iy < m) s = fma(a[idx(ix, iy)], b[idx(ix, iy)], s);
          }
        lanes[l] = s;
      }
      waves[gw] = tree(lanes, 64);
    }
    return tree(waves, 32);
  }
  void apply(const std::vector<double>& v, std::vector<double>& out) const {
    for (int iy = 0; iy < m; ++iy)
      for (int ix = 0; ix < m; ++ix) {
        const size_t q = idx(ix, iy);
        double a = c[0] * v[q];
        a = fma(c[1], v[q - 1], a);
        a = fma(c[2], v[q + 1], a);
        a = fma(c[3], v[q - pitch], a);
        a = fma(c[4], v[q + pitch], a);
        out[q] = a;
      }
  }
  int solve(std::vector<double>& x, const std::vector<double>& b, std::vector<double>& r, int maxit, double tol) const {
    const size_t N = (size_t)pitch * n;
    std::vector<double> p(N, 0.0), s(N, 0.0), w(N, 0.0), ax(N, 0.0);
    apply(x, ax);
    std::fill(r.begin(), r.end(), 0.0);
    for (int iy = 0; iy < m; ++iy)
      for (int ix = 0; ix < m; ++ix) r[idx(ix, iy)] = b[idx(ix, iy)] - ax[idx(ix, iy)];
    apply(r, w);
    double g = dot(r, r), d = dot(w, r);
    const double res0 = sqrt(g);
    if (res0 == 0.0 || maxit <= 0) return 0;
    double den = d, alpha = g / d, beta = 0.0;
    int it = 0;
    for (;;) {
      for (int iy = 0; iy < m; ++iy)
        for (int ix = 0; ix < m; ++ix) {
          const size_t q = idx(ix, iy);
          p[q] = fma(beta, p[q], r[q]);
          s[q] = fma(beta, s[q], w[q]);
          x[q] = fma(alpha, p[q], x[q]);
          r[q] = fma(-alpha, s[q], r[q]);
        }
      apply(r, w);
      const double inv_g = 1.0 / g;
      const double gn = dot(r, r);
      d = dot(w, r);
      ++it;
      if (!(sqrt(gn) > tol * res0) || it >= maxit) break;
      beta = gn * inv_g;
      den = fma(-(beta * beta), den, d);
      alpha = gn / den;
      g = gn;
    }
    return it;
  }
};

template <int R, int E, int H, int WAVES, bool MULTI, int MASK, bool AUTO = false>
static void run(const char* name, int level, int groups, int stride, bool check) {
  const int n = (1 << level) + 1, m = n - 2;
  const int pitch = (n + 15) / 16 * 16;
  const size_t N = (size_t)pitch * n;
  Geo g{n, pitch, 0};
  Stencil<double> st;
  const double h = 1.0 / (n - 1);
  st.c[0] = 4.0 / (h * h);
  for (int k = 1; k < 5; ++k) st.c[k] = -1.0 / (h * h);
  st.c[5] = st.c[6] = 0.0;
  std::vector<double> hb(N, 0.0), hx(N, 0.0);
  srand(level);
  for (int iy = 0; iy < m; ++iy)
    for (int ix = 0; ix < m; ++ix) hb[(size_t)(ix + 1) + (size_t)pitch * (iy + 1)] = rand() / (double)RAND_MAX - 0.5;
  double *dx, *db, *dr;
  int* dinfo;
  evomg_u64* ll;
  CHECK(hipMalloc(&dx, N * 8));
  CHECK(hipMalloc(&db, N * 8));
  CHECK(hipMalloc(&dr, N * 8));
  CHECK(hipMalloc(&dinfo, 128));
  CHECK(hipMalloc(&ll, EVOMG_CG1_LL_WORDS * 8));
  CHECK(hipMemset(ll, 0, EVOMG_CG1_LL_WORDS * 8));
  CHECK(hipMemset(dinfo, 0, 128));
  CHECK(hipMemset(dr, 0, N * 8));
  CHECK(hipMemcpy(db, hb.data(), N * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned seq = 0;
  float best = 1e9f;
  int info[32] = {};
  const int blocks = MULTI ? groups * stride : 1;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipMemset(dx, 0, N * 8));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((evomg_cg1_band2d<R, E, H, WAVES, MULTI, false, MASK, AUTO>), dim3(blocks), dim3(64 * WAVES), 0, 0, g, st, dx, (const double*)db, dr,
                       (double*)nullptr, 1000, 1e-12, dinfo, ll, seq, groups, stride, 1);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    seq += 1003;
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    CHECK(hipMemcpy(info, dinfo, 128, hipMemcpyDeviceToHost));
  }
  printf("%-34s %d^2: %4d iterations, %8.1f us per solve, %.3f us per iteration, status %d\n", name, n, info[0], best * 1e3, best * 1e3 / (info[0] ? info[0] : 1), info[1]);
#ifdef EVOMG_CG1_PROFILE
  {
    unsigned long long rounds[2], zero[2] = {0, 0};
    CHECK(hipMemcpyFromSymbol(rounds, HIP_SYMBOL(evomg_ll_poll_rounds), sizeof rounds));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(evomg_ll_poll_rounds), zero, sizeof zero));
    printf("    polls %llu, fetch rounds per poll %.3f\n", rounds[0], rounds[0] ? (double)rounds[1] / (double)rounds[0] : 0.0);
    const long long* pr = reinterpret_cast<const long long*>(info + 4);
    long long tot = 0;
    for (int k = 0; k < 6; ++k) tot += pr[k];
    printf("    ticks per iteration (wave 0): compute %.0f | wave sums %.0f | publish+poll %.0f | barrier %.0f | tree+ghosts %.0f | scalars %.0f | total %.0f\n",
           pr[0] / (double)info[0], pr[1] / (double)info[0], pr[2] / (double)info[0], pr[3] / (double)info[0], pr[4] / (double)info[0], pr[5] / (double)info[0], tot / (double)info[0]);
  }
#endif
  if (check) {
    CHECK(hipMemcpy(hx.data(), dx, N * 8, hipMemcpyDeviceToHost));
    std::vector<double> hr(N), cx(N, 0.0), cr(N, 0.0);
    CHECK(hipMemcpy(hr.data(), dr, N * 8, hipMemcpyDeviceToHost));
    HostCg1 hc{n, m, pitch, R, E, H, {st.c[0], st.c[1], st.c[2], st.c[3], st.c[4]}};
    const int its = hc.solve(cx, hb, cr, 1000, 1e-12);
    size_t bad = 0;
    double maxd = 0.0;
    for (size_t q = 0; q < N; ++q) {
      if (memcmp(&cx[q], &hx[q], 8) != 0 || memcmp(&cr[q], &hr[q], 8) != 0) ++bad;
      maxd = fmax(maxd, fabs(cx[q] - hx[q]));
    }
    printf("    host restatement: %d iterations, %zu of %zu values differ (x and r), max |dx| %.3e\n", its, bad, N, maxd);
  }
  CHECK(hipFree(dx)); CHECK(hipFree(db)); CHECK(hipFree(dr)); CHECK(hipFree(dinfo)); CHECK(hipFree(ll));
}

int main(int argc, char** argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 7;
  if (level == 7) {
    run<4, 2, 1, 4, true, 3>("cg1 R4 E2, 8 groups x 4 waves, one XCD", 7, 8, 8, true);
    run<4, 2, 1, 4, true, 3>("cg1 R4 E2, 8 groups x 4 waves, 8 XCDs", 7, 8, 1, false);
    run<4, 2, 1, 4, true, 3, true>("cg1 R4 E2, 8 x 4, one XCD, every wave for itself", 7, 8, 8, true);
    run<4, 2, 1, 2, true, 3, true>("cg1 R4 E2, 16 x 2, one XCD, every wave for itself", 7, 16, 8, true);
    run<4, 2, 1, 4, true, 3, true>("cg1 R4 E2, 8 x 4, 8 XCDs, every wave for itself", 7, 8, 1, false);
    run<4, 2, 1, 8, true, 3>("cg1 R4 E2, 4 groups x 8 waves, one XCD", 7, 4, 8, true);
    run<4, 2, 1, 2, true, 3>("cg1 R4 E2, 16 groups x 2 waves, one XCD", 7, 16, 8, false);
    run<8, 2, 1, 4, true, 7>("cg1 R8 E2, 4 groups x 4 waves, one XCD", 7, 4, 8, true);
    run<8, 2, 1, 2, true, 7>("cg1 R8 E2, 8 groups x 2 waves, one XCD", 7, 8, 8, false);
  } else if (level == 6) {
    run<4, 1, 1, 16, false, 3>("cg1 R4 E1, 1 group x 16 waves", 6, 1, 1, true);
    run<8, 1, 1, 8, false, 7>("cg1 R8 E1, 1 group x 8 waves", 6, 1, 1, true);
    run<16, 1, 1, 4, false, 15>("cg1 R16 E1, 1 group x 4 waves", 6, 1, 1, true);
    run<4, 1, 1, 4, true, 3>("cg1 R4 E1, 4 groups x 4 waves", 6, 4, 8, true);
  } else if (level == 5) {
    run<4, 1, 2, 4, false, 3>("cg1 R4 E1 H2, 4 waves", 5, 1, 1, true);
    run<8, 1, 2, 2, false, 7>("cg1 R8 E1 H2, 2 waves", 5, 1, 1, true);
    run<16, 1, 2, 1, false, 15>("cg1 R16 E1 H2, 1 wave", 5, 1, 1, true);
    run<4, 1, 1, 8, false, 3>("cg1 R4 E1 H1, 8 waves", 5, 1, 1, true);
  } else if (level == 4) {
    run<2, 1, 4, 2, false, 1>("cg1 R2 E1 H4, 2 waves", 4, 1, 1, true);
    run<4, 1, 4, 1, false, 3>("cg1 R4 E1 H4, 1 wave", 4, 1, 1, true);
  } else if (level == 3) {
    run<1, 1, 8, 1, false, 0>("cg1 R1 E1 H8, 1 wave", 3, 1, 1, true);
  } else if (level == 2) {
    run<1, 1, 16, 1, false, 0>("cg1 R1 E1 H16, 1 wave", 2, 1, 1, true);
  }
  return 0;
}
