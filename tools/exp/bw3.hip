// Bandwidth probe 3 (experiments only, round 3).
//
// (1) Which copy shape reaches the device's read+write ceiling?  MI355X_MICROARCH.md quotes
//     6.29 TB/s for "a float4 copy"; round 2's best shape (256 KiB contiguous chunks per
//     workgroup, non-temporal) measured 5.46-5.6 TB/s on this pool.  The simplest shapes are
//     put next to it here: one / two / four float4 per thread, no loop, a block per 4-16 KiB.
// (2) What does the forward predictor's access shape cost on the memory side alone?  Persistent
//     waves pulling strips, 32 B of pixels in and 16 + 16 B of planes out per lane and step,
//     with and without the delta planes' 16 + 16 B (L2 hits), and with ONE wave working on
//     the same strip of F = 1 / 2 / 4 consecutive frames so that the delta registers are
//     shared (the restructuring VERDICT r02 asks for), at several depths and waves per CU.
//     No predictor arithmetic: split by byte permutes, SWAR subtraction, stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- simple copies ----
template <int PER, int T, bool NT>
__global__ __launch_bounds__(T) void k_simple(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t base = (size_t)blockIdx.x * (T * PER) + threadIdx.x;
  u32x4 v[PER];
#pragma unroll
  for (int u = 0; u < PER; u++)
    if (base + u * T < n16) v[u] = NT ? __builtin_nontemporal_load(src + base + u * T) : src[base + u * T];
#pragma unroll
  for (int u = 0; u < PER; u++)
    if (base + u * T < n16) {
      if (NT) __builtin_nontemporal_store(v[u], dst + base + u * T);
      else dst[base + u * T] = v[u];
    }
}

// float4 copy as a textbook grid-stride loop (what "float4 copy" most likely means)
template <int T>
__global__ __launch_bounds__(T) void k_gs_f4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n16; i += (size_t)gridDim.x * T) dst[i] = src[i];
}

// round 2's best: contiguous chunk per workgroup
template <int UNROLL, int T, bool NT>
__global__ __launch_bounds__(T) void k_chunk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, size_t chunk16) {
  for (size_t c = blockIdx.x; c * chunk16 < n16; c += gridDim.x) {
    const size_t b = c * chunk16, e = b + chunk16 < n16 ? b + chunk16 : n16;
    size_t i = b + threadIdx.x;
    for (; i + (UNROLL - 1) * T < e; i += UNROLL * T) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * T) : src[i + u * T];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        if (NT) __builtin_nontemporal_store(v[u], dst + i + u * T);
        else dst[i + u * T] = v[u];
      }
    }
  }
}

// ---- predictor-shaped streams ----
__device__ __forceinline__ unsigned sub4(unsigned a, unsigned b) {
  const unsigned H = 0x80808080u;
  return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}

struct SArgs {
  const unsigned short* frames;
  const unsigned char *dh, *dl;
  unsigned char *high, *low;
  unsigned* ctr;
  unsigned npix, strip_px, strips, nframes;
};

// One wave = one strip of F consecutive frames at a time.  DEPTH steps of loads in flight.
template <int F, int DEPTH, bool DELTA, bool NTD>
__global__ __launch_bounds__(64) void k_stream_shape(SArgs g) {
  const unsigned lane = threadIdx.x;
  const unsigned ngroups = (g.nframes + F - 1) / F;
  const unsigned ntasks = ngroups * g.strips;
  unsigned task = blockIdx.x;
  while (task < ntasks) {
    const unsigned strip = task / ngroups, group = task - strip * ngroups;
    const unsigned s = strip * g.strip_px;
    const unsigned niter = g.strip_px / 1024;
    const unsigned short* img[F];
    unsigned char *hi[F], *lo[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
      const unsigned fr = min(group * F + f, g.nframes - 1);
      img[f] = g.frames + (size_t)fr * g.npix + s;
      hi[f] = g.high + (size_t)fr * g.npix + s;
      lo[f] = g.low + (size_t)fr * g.npix + s;
    }
    u32x4 a[DEPTH][F], b[DEPTH][F], dh[DEPTH], dl[DEPTH];
    auto load = [&](int d, unsigned it) {
      const unsigned o = min(it, niter - 1) * 1024 + lane * 16;
#pragma unroll
      for (int f = 0; f < F; f++) {
        a[d][f] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(img[f] + o));
        b[d][f] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(img[f] + o + 8));
      }
      if (DELTA) {
        if (NTD) {
          dh[d] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g.dh + s + o));
          dl[d] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g.dl + s + o));
        } else {
          dh[d] = *reinterpret_cast<const u32x4*>(g.dh + s + o);
          dl[d] = *reinterpret_cast<const u32x4*>(g.dl + s + o);
        }
      }
    };
    auto step = [&](int d, unsigned it) {
      const unsigned o = it * 1024 + lane * 16;
#pragma unroll
      for (int f = 0; f < F; f++) {
        u32x4 h, l;
        const u32x4 A = a[d][f], B = b[d][f];
        h.x = __builtin_amdgcn_perm(A.y, A.x, 0x07050301u); l.x = __builtin_amdgcn_perm(A.y, A.x, 0x06040200u);
        h.y = __builtin_amdgcn_perm(A.w, A.z, 0x07050301u); l.y = __builtin_amdgcn_perm(A.w, A.z, 0x06040200u);
        h.z = __builtin_amdgcn_perm(B.y, B.x, 0x07050301u); l.z = __builtin_amdgcn_perm(B.y, B.x, 0x06040200u);
        h.w = __builtin_amdgcn_perm(B.w, B.z, 0x07050301u); l.w = __builtin_amdgcn_perm(B.w, B.z, 0x06040200u);
        if (DELTA) {
          h.x = sub4(h.x, dh[d].x); h.y = sub4(h.y, dh[d].y); h.z = sub4(h.z, dh[d].z); h.w = sub4(h.w, dh[d].w);
          l.x = sub4(l.x, dl[d].x); l.y = sub4(l.y, dl[d].y); l.z = sub4(l.z, dl[d].z); l.w = sub4(l.w, dl[d].w);
        }
        __builtin_nontemporal_store(l, reinterpret_cast<u32x4*>(lo[f] + o));
        __builtin_nontemporal_store(h, reinterpret_cast<u32x4*>(hi[f] + o));
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(d, d);
    const unsigned nd = niter - niter % DEPTH;
    for (unsigned it = 0; it < nd; it += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        step(d, it + d);
        load(d, it + d + DEPTH);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++)
      if (nd + d < niter) step(d, nd + d);
    unsigned next = 0;
    if (lane == 0) next = atomicAdd(g.ctr, 1u);
    task = gridDim.x + (unsigned)__builtin_amdgcn_readfirstlane((int)next);
  }
}

static hipEvent_t e0, e1;
static double g_best = 0;
template <typename L>
static double time_it(const char* name, double bytes, int reps, L launch) {
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < 3; r++) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; k++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double gbps = bytes / best / 1e6;
  printf("%-64s %8.4f ms (mean %8.4f) %7.0f GB/s\n", name, best, sum / 3, gbps);
  fflush(stdout);
  return gbps;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  char nm[160];
  if (!strcmp(what, "all") || !strcmp(what, "copy")) {
    for (size_t mib : {1024, 2048}) {
      const size_t bytes = mib << 20, n16 = bytes / 16;
      u32x4 *a, *b;
      CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
      CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
      printf("--- copies, %zu MiB each way ---\n", mib);
      const int R = 10;
      double best = 0;
      auto upd = [&](double v) { if (v > best) best = v; };
      upd(time_it("hipMemcpyDtoD", 2.0 * bytes, R, [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }));
#define SIMPLE(PER, T, NT)                                                                         \
  snprintf(nm, sizeof nm, "simple %d x float4 per thread, T=%d, %s", PER, T, NT ? "nt" : "plain"); \
  upd(time_it(nm, 2.0 * bytes, R, [&] {                                                            \
    k_simple<PER, T, NT><<<(unsigned)((n16 + (size_t)T * PER - 1) / ((size_t)T * PER)), T>>>(a, b, n16); }));
      SIMPLE(1, 256, false) SIMPLE(1, 256, true) SIMPLE(1, 512, false) SIMPLE(1, 1024, false) SIMPLE(1, 1024, true)
      SIMPLE(2, 256, false) SIMPLE(2, 256, true) SIMPLE(4, 256, false) SIMPLE(4, 256, true)
      SIMPLE(4, 512, true) SIMPLE(8, 256, true) SIMPLE(4, 64, true) SIMPLE(8, 64, true) SIMPLE(16, 64, true)
      for (int blocks : {cus * 4, cus * 8, cus * 16, cus * 32}) {
        snprintf(nm, sizeof nm, "grid-stride float4 T=256 blocks=%d", blocks);
        upd(time_it(nm, 2.0 * bytes, R, [&] { k_gs_f4<256><<<blocks, 256>>>((const float4*)a, (float4*)b, n16); }));
        snprintf(nm, sizeof nm, "grid-stride float4 T=1024 blocks=%d", blocks / 4);
        upd(time_it(nm, 2.0 * bytes, R, [&] { k_gs_f4<1024><<<blocks / 4, 1024>>>((const float4*)a, (float4*)b, n16); }));
      }
      for (size_t chunkKB : {64, 256}) {
        snprintf(nm, sizeof nm, "chunk copy nt u4 T256 chunk=%zuK blocks=%d (round 2's k_copy)", chunkKB, cus * 32);
        upd(time_it(nm, 2.0 * bytes, R, [&] { k_chunk<4, 256, true><<<cus * 32, 256>>>(a, b, n16, chunkKB * 1024 / 16); }));
        snprintf(nm, sizeof nm, "chunk copy nt u8 T256 chunk=%zuK blocks=%d", chunkKB, cus * 8);
        upd(time_it(nm, 2.0 * bytes, R, [&] { k_chunk<8, 256, true><<<cus * 8, 256>>>(a, b, n16, chunkKB * 1024 / 16); }));
      }
      printf("BEST COPY %zu MiB: %.0f GB/s\n", mib, best);
      if (best > g_best) g_best = best;
      CK(hipFree(a)); CK(hipFree(b));
    }
  }
  if (!strcmp(what, "all") || !strcmp(what, "stream")) {
    const unsigned xs = 1280, ys = 800, npix = xs * ys, N = 1000;
    SArgs g;
    unsigned short* frames; unsigned char *dh, *dl, *high, *low; unsigned* ctr;
    CK(hipMalloc(&frames, (size_t)N * npix * 2)); CK(hipMalloc(&high, (size_t)N * npix)); CK(hipMalloc(&low, (size_t)N * npix));
    CK(hipMalloc(&dh, npix)); CK(hipMalloc(&dl, npix)); CK(hipMalloc(&ctr, 256));
    CK(hipMemset(frames, 3, (size_t)N * npix * 2)); CK(hipMemset(dh, 1, npix)); CK(hipMemset(dl, 2, npix));
    g.frames = frames; g.dh = dh; g.dl = dl; g.high = high; g.low = low; g.ctr = ctr;
    g.npix = npix; g.nframes = N;
    const double bytes = (double)N * npix * 4.0;
    printf("--- predictor-shaped streams: 1000 x 1280x800, %0.3f GB algorithmic per pass (preview not counted) ---\n", bytes / 1e9);
    for (unsigned srows : {160u, 80u, 32u}) {
      g.strip_px = srows * xs; g.strips = ys / srows;
#define SHAPE(F, D, DELTA, NTD, WPC)                                                                                       \
  snprintf(nm, sizeof nm, "strip %u rows F=%d depth=%d delta=%d%s waves/CU=%d", srows, F, D, DELTA, NTD ? "(nt)" : "", WPC); \
  time_it(nm, bytes, 5, [&] {                                                                                              \
    hipMemsetAsync(ctr, 0, 4, 0);                                                                                          \
    k_stream_shape<F, D, DELTA, NTD><<<cus * WPC, 64>>>(g); });
      SHAPE(1, 3, false, false, 8) SHAPE(1, 3, true, false, 8) SHAPE(1, 3, true, true, 8)
      SHAPE(1, 2, true, false, 8) SHAPE(1, 4, true, false, 8) SHAPE(1, 3, true, false, 4) SHAPE(1, 3, true, false, 12)
      SHAPE(1, 3, true, false, 16)
      SHAPE(2, 2, true, false, 8) SHAPE(2, 2, true, false, 4) SHAPE(2, 2, true, false, 6) SHAPE(2, 3, true, false, 4)
      SHAPE(2, 1, true, false, 8) SHAPE(2, 3, true, false, 8) SHAPE(2, 2, false, false, 8)
      SHAPE(4, 1, true, false, 4) SHAPE(4, 1, true, false, 8) SHAPE(4, 2, true, false, 4) SHAPE(4, 2, true, false, 2)
      SHAPE(4, 2, true, false, 8)
    }
  }
  return 0;
}
