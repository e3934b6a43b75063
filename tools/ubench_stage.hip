// What does ONE k_pipe stage wave's instruction stream cost, by itself?  Runs alz_casc.hip's section_tile_emit (the
// pinned, interleaved 16-sample tile of one gammatone.slaney section: 112 FP64 operations) in a loop, one to four waves
// per workgroup (one per SIMD), 256 workgroups, with the hand-over's LDS operations switched on one by one:
//   mode 0  arithmetic only (results kept live in registers)
//   mode 1  + the eight ds_write_b128 of the finished pairs
//   mode 2  + the eight ds_read_b128 of the next tile
//   mode 3  + s_waitcnt lgkmcnt(0) and s_barrier after every tile
//   mode 4  the FRONT-LOADED schedule: every LDS operation of an interval is issued in its first six groups -- reads
//           2 per group in groups 0 - 3, the previous tile's pairs 4 - 7 written in groups 0 - 3, this tile's pairs
//           0 - 3 in groups 4 - 5 -- so that the drain before the barrier finds them finished; + drain + barrier
//   mode 5  mode 4 without the barrier
//   mode 6  reads only: the eight ds_read_b128 of the next tile in groups 0 - 7, no writes, drain at the end
//   mode 7  the same bytes as sixteen ds_read_b64
//   mode 8  the same bytes as eight ds_read_b128 issued back to back in group 0
//   mode 9  writes only, as sixteen ds_write_b64
// Reports shader cycles per tile (s_memtime) of wave 0 of workgroup 5.  k_pipe's stage waves measure ~1000 - 1095 per
// tile inside the kernel (profiles/NOTES_r04.md); 112 operations at the 4.7 cycles a lone wave issues them are 526.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I audiolazy_amd/csrc -I include [-DALZ_PIPE_ILV=0]
#include "alz_casc.hip"
#include <cstdio>
namespace alz {   // (the two host helpers alz_casc.hip's launcher refers to; not used here)
int ensure_dynamic_lds(const void *, int) { return 0; }
int fail(int code, const std::string &) { return code; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, bool FMA>
__global__ __launch_bounds__(320) void k_stage(double *out, long long *cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char *ring = smem + wave * 2 * 8192;
  double bc[8] = {0.5, 0.25, 0, 0, 0, 0, 0, 0}, dx[7] = {0, 0, 0, 0, 0, 0, 0};
  double na1 = 1.2 + lane * 1e-6, na2 = -0.5, m1 = 0.1, m2 = 0.2;
  asm volatile("" : "+v"(na1), "+v"(na2), "+v"(bc[0]), "+v"(bc[1]));
  double va[16], vb[16];
  for (int u = 0; u < 16; ++u) { va[u] = 1e-3 * (u + 1) + lane * 1e-7; vb[u] = 2e-3 * (u + 1); }
  for (int i = lane; i < 2 * 8192 / 8; i += 64) reinterpret_cast<double *>(ring)[i] = 1e-3;
  __syncthreads();
  double keep = 0.0;
  auto one = [&](int t, double (&cur)[16], double (&nxt)[16]) {
    char *dst = ring + (t & 1) * 8192 + lane * 16;
    const char *src = ring + ((t + 1) & 1) * 8192 + lane * 16;
    alz::section_tile_emit<2, 3u, 3u, FMA, 1>(cur, bc, na1, na2, dx, m1, m2,
        [&](int j, double a, double b) {
          if constexpr (MODE >= 1) { alz::cdbl2 w; w.x = a; w.y = b; *reinterpret_cast<alz::cdbl2 *>(dst + j * 1024) = w; }
          else { keep += a; keep += b; }
        },
        [&](int j) {
          if constexpr (MODE >= 2) { const alz::cdbl2 w = *reinterpret_cast<const alz::cdbl2 *>(src + j * 1024); nxt[2 * j] = w.x; nxt[2 * j + 1] = w.y; }
        });
    if constexpr (MODE >= 3) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }
  };
  double oa[16], ob[16];
  for (int u = 0; u < 16; ++u) oa[u] = ob[u] = 0.0;
  // front-loaded schedule: `prev` = the previous tile's outputs (its pairs 4 - 7 are still to be written)
  auto two = [&](int t, double (&cur)[16], double (&nxt)[16], double (&o)[16], const double (&prev)[16]) {
    char *dst = ring + (t & 1) * 8192 + lane * 16;
    char *dprev = ring + ((t + 1) & 1) * 8192 + lane * 16;
    const char *src = ring + ((t + 1) & 1) * 8192 + lane * 16;
    auto wr = [&](char *base, int j, double a, double b) { alz::cdbl2 w; w.x = a; w.y = b; *reinterpret_cast<alz::cdbl2 *>(base + j * 1024) = w; };
    auto rd = [&](int j) { const alz::cdbl2 w = *reinterpret_cast<const alz::cdbl2 *>(src + j * 1024); nxt[2 * j] = w.x; nxt[2 * j + 1] = w.y; };
    alz::section_tile_hook<2, 3u, 3u, FMA, 1>(cur, bc, na1, na2, dx, m1, m2, o, [&](int j) {
      if (j < 4) { rd(2 * j); rd(2 * j + 1); wr(dprev, 4 + j, prev[8 + 2 * j], prev[9 + 2 * j]); }
      else if (j < 6) { wr(dst, 2 * (j - 4), o[4 * (j - 4)], o[4 * (j - 4) + 1]); wr(dst, 2 * (j - 4) + 1, o[4 * (j - 4) + 2], o[4 * (j - 4) + 3]); }
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if constexpr (MODE == 4) __builtin_amdgcn_s_barrier();
  };
  auto three = [&](int t, double (&cur)[16], double (&nxt)[16], double (&o)[16]) {
    char *dst = ring + (t & 1) * 8192 + lane * 16;
    const char *src = ring + ((t + 1) & 1) * 8192 + lane * 16;
    auto rd = [&](int j) { const alz::cdbl2 w = *reinterpret_cast<const alz::cdbl2 *>(src + j * 1024); nxt[2 * j] = w.x; nxt[2 * j + 1] = w.y; };
    alz::section_tile_hook<2, 3u, 3u, FMA, 1>(cur, bc, na1, na2, dx, m1, m2, o, [&](int j) {
      if constexpr (MODE == 6) rd(j);
      if constexpr (MODE == 7) {
        nxt[2 * j] = *reinterpret_cast<const double *>(src + j * 1024 - lane * 8);
        nxt[2 * j + 1] = *reinterpret_cast<const double *>(src + j * 1024 + 512 - lane * 8);
      }
      if constexpr (MODE == 8) { if (j == 0) { rd(0); rd(1); rd(2); rd(3); rd(4); rd(5); rd(6); rd(7); } }
      if constexpr (MODE == 9) {
        if (j > 0) {
          *reinterpret_cast<double *>(dst + (j - 1) * 1024 - lane * 8) = o[2 * j - 2];
          *reinterpret_cast<double *>(dst + (j - 1) * 1024 + 512 - lane * 8) = o[2 * j - 1];
        }
      }
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  const long long c0 = __builtin_readcyclecounter();
  if constexpr (MODE >= 6) {
    for (int t = 0; t < tiles; t += 2) {
      three(t, va, vb, oa);
      three(t + 1, vb, va, ob);
    }
    keep += oa[15] + ob[15];
  } else if constexpr (MODE >= 4) {
    for (int t = 0; t < tiles; t += 2) {
      two(t, va, vb, oa, ob);
      two(t + 1, vb, va, ob, oa);
    }
    keep += oa[15] + ob[15];
  } else {
    for (int t = 0; t < tiles; t += 2) {
      one(t, va, vb);
      one(t + 1, vb, va);
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * 320 + threadIdx.x] = m1 + m2 + keep + va[3] + vb[5];
  if (blockIdx.x == 5 && threadIdx.x == 0) cyc[0] = c1 - c0;
}

template <int MODE, bool FMA>
static int run(double *out, long long *cyc, int waves) {
  const int tiles = 4096;
  hipLaunchKernelGGL((k_stage<MODE, FMA>), dim3(256), dim3(64 * waves), 5 * 2 * 8192, 0, out, cyc, 64);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_stage<MODE, FMA>), dim3(256), dim3(64 * waves), 5 * 2 * 8192, 0, out, cyc, tiles);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h; CK(hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  printf("mode %d %s, %d wave(s) per workgroup: %7.1f cycles per 16-sample tile (%.0f ns; %.2f cycles per FP64 operation)\n", MODE,
         FMA ? "fma" : "mul+add", waves, (double)h / tiles, ms * 1e6 / tiles, (double)h / tiles / (FMA ? 64 : 112));
  return 0;
}

int main() {
  double *out; long long *cyc;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipMalloc(&out, 256 * 320 * sizeof(double)));
  CK(hipMalloc(&cyc, 64));
  printf("ALZ_PIPE_ILV = %d\n", ALZ_PIPE_ILV);
  for (int waves : {1, 4}) {
    if (run<0, false>(out, cyc, waves)) return 1;
    if (run<1, false>(out, cyc, waves)) return 1;
    if (run<2, false>(out, cyc, waves)) return 1;
    if (run<3, false>(out, cyc, waves)) return 1;
    if (run<4, false>(out, cyc, waves)) return 1;
    if (run<5, false>(out, cyc, waves)) return 1;
    if (run<6, false>(out, cyc, waves)) return 1;
    if (run<7, false>(out, cyc, waves)) return 1;
    if (run<8, false>(out, cyc, waves)) return 1;
    if (run<9, false>(out, cyc, waves)) return 1;
    if (run<6, true>(out, cyc, waves)) return 1;
    if (run<0, true>(out, cyc, waves)) return 1;
    if (run<3, true>(out, cyc, waves)) return 1;
    if (run<4, true>(out, cyc, waves)) return 1;
  }
  return 0;
}
