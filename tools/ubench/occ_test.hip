// Dev tool: how many 384-thread workgroups does a gfx950 CU really hold, as a function of VGPRs and LDS?
// Each workgroup stamps wall_clock64 around a fixed busy loop and records its CU; the host counts overlaps per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <algorithm>

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc; unsigned simd[8]; };

template <int LDS_BYTES, int TOPV, int THREADS>
__global__ void __launch_bounds__(THREADS) occ_kernel(Rec* recs, int spin) {
  __shared__ float lds[LDS_BYTES / 4];
  if (TOPV == 167) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (TOPV == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (TOPV == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  float a = lds[(threadIdx.x + 1) % THREADS];
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (a == 12345.f) lds[0] = a;
  if ((threadIdx.x & 63) == 0) recs[blockIdx.x].simd[threadIdx.x >> 6] = (hwid >> 4) & 3;
  if (threadIdx.x == 0) { recs[blockIdx.x].t0 = t0; recs[blockIdx.x].t1 = t1; recs[blockIdx.x].hwid = hwid; recs[blockIdx.x].xcc = xcc; }
}

template <int LDS_BYTES, int TOPV, int THREADS>
void run(const char* name) {
  const int grid = 2048;
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * grid);
  hipMemset(d, 0, sizeof(Rec) * grid);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, occ_kernel<LDS_BYTES, TOPV, THREADS>, THREADS, 0);
  hipLaunchKernelGGL((occ_kernel<LDS_BYTES, TOPV, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<Rec> h(grid);
  hipMemcpy(h.data(), d, sizeof(Rec) * grid, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
  int simd_hist[5] = {0, 0, 0, 0, 0};
  for (auto& r : h) {
    const unsigned key = (r.xcc & 15) << 16 | (r.hwid & 0xff00);  // cu_id[11:8] sh[12] se[15:13]
    ev[key].push_back({r.t0, +1});
    ev[key].push_back({r.t1, -1});
    int cnt[4] = {0, 0, 0, 0};
    for (int w = 0; w < THREADS / 64; ++w) cnt[r.simd[w] & 3]++;
    simd_hist[*std::max_element(cnt, cnt + 4)]++;
  }
  int mx = 0;
  double avg = 0;
  for (auto& kv : ev) {
    auto v = kv.second;
    std::sort(v.begin(), v.end());
    int cur = 0, m = 0;
    for (auto& e : v) { cur += e.second; m = std::max(m, cur); }
    mx = std::max(mx, m);
    avg += m;
  }
  printf("%-34s API occupancy %d | CUs seen %zu | max concurrent blocks/CU %d (avg of per-CU max %.2f) | max waves of a block on one SIMD: 1:%d 2:%d 3:%d 4:%d\n",
         name, occ, ev.size(), mx, avg / ev.size(), simd_hist[1], simd_hist[2], simd_hist[3], simd_hist[4]);
  hipFree(d);
}

int main() {
  run<79872, 167, 384>("384thr 168vgpr 78KB");
  run<39936, 167, 384>("384thr 168vgpr 39KB");
  run<79872, 127, 384>("384thr 128vgpr 78KB");
  run<39936, 127, 384>("384thr 128vgpr 39KB");
  run<39936, 95, 384>("384thr  96vgpr 39KB");
  run<1024, 167, 384>("384thr 168vgpr  1KB");
  run<1024, 167, 256>("256thr 168vgpr  1KB");
  run<1024, 167, 192>("192thr 168vgpr  1KB");
  run<79872, 167, 256>("256thr 168vgpr 78KB");
  run<79872, 167, 512>("512thr 128vgpr? 78KB");
  return 0;
}
