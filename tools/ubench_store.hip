// Store-stream floor for the N x N float matrix of k_pairwise (N = 10 000, 400 MB): which tile shape
// writes fastest when there is no arithmetic at all?  Every kernel writes each element once with
// 16-byte stores; shapes differ in how many columns a thread / wave covers and how many rows a
// workgroup walks.  Also: hipMemsetAsync of the same buffer as the fill reference.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench_store.hip -o tools/ubench_store.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// thread: VEC float4 stores per row (16 * VEC contiguous bytes at stride 256 * 16 bytes when STRIDED, else adjacent);
// block: 256 threads x ROWS rows
template <int VEC, int ROWS, bool ADJ>
__global__ __launch_bounds__(256) void k_store(float *out, int n, float v) {
  const int cols_per_block = 256 * 4 * VEC;
  const int c0 = blockIdx.x * cols_per_block;
  const int r0 = blockIdx.y * ROWS;
  for (int row = r0; row < r0 + ROWS && row < n; ++row) {
    float *orow = out + (size_t)row * n;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int j = ADJ ? c0 + (threadIdx.x * VEC + k) * 4 : c0 + (k * 256 + threadIdx.x) * 4;
      if (j + 4 <= n) *reinterpret_cast<float4 *>(orow + j) = make_float4(v, v + row, v + k, v);
    }
  }
}

// one wave per row segment: a workgroup of 4 waves writes 4 rows at a time, each wave a contiguous run of SEG KB
template <int SEGF4>  // float4 per lane per row
__global__ __launch_bounds__(256) void k_store_rowwave(float *out, int n, float v, int rows_per_block) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cols_per_block = 64 * 4 * SEGF4;
  const int c0 = blockIdx.x * cols_per_block;
  const int r0 = blockIdx.y * rows_per_block;
  for (int row = r0 + wave; row < r0 + rows_per_block && row < n; row += 4) {
    float *orow = out + (size_t)row * n;
#pragma unroll
    for (int k = 0; k < SEGF4; ++k) {
      const int j = c0 + (k * 64 + lane) * 4;
      if (j + 4 <= n) *reinterpret_cast<float4 *>(orow + j) = make_float4(v, v + row, v + k, v);
    }
  }
}

// linear fill: grid-stride over float4 elements, UNR stores in flight per thread
template <int UNR> __global__ __launch_bounds__(256) void k_linear(float4 *out, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
#pragma unroll
    for (int k = 0; k < UNR; ++k) out[i + k * stride] = make_float4(v, v, v + k, v);
  }
  for (; i < n4; i += stride) out[i] = make_float4(v, v, v, v);
}
// chunked linear fill: block b owns the contiguous chunk [b * chunk4, (b + 1) * chunk4) of float4 elements
__global__ __launch_bounds__(256) void k_chunk(float4 *out, size_t n4, size_t chunk4, float v) {
  const size_t b0 = (size_t)blockIdx.x * chunk4;
  for (size_t i = b0 + threadIdx.x; i < b0 + chunk4 && i < n4; i += 256) out[i] = make_float4(v, v, v, v);
}
// rows: block (x = row group of ROWS rows) walks ALL column tiles itself, tile by tile (column-tile outer, rows inner)
template <int ROWS> __global__ __launch_bounds__(256) void k_rowgroup(float *out, int n, float v) {
  const int r0 = blockIdx.x * ROWS;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int j = c0 + threadIdx.x * 4;
    for (int row = r0; row < r0 + ROWS && row < n; ++row)
      if (j + 4 <= n) *reinterpret_cast<float4 *>(out + (size_t)row * n + j) = make_float4(v, v + row, v, v);
  }
}
// the same, rows outer (every row written start to end before the next)
template <int ROWS> __global__ __launch_bounds__(256) void k_rowgroup_rowmajor(float *out, int n, float v) {
  const int r0 = blockIdx.x * ROWS;
  for (int row = r0; row < r0 + ROWS && row < n; ++row)
    for (int c0 = 0; c0 < n; c0 += 1024) {
      const int j = c0 + threadIdx.x * 4;
      if (j + 4 <= n) *reinterpret_cast<float4 *>(out + (size_t)row * n + j) = make_float4(v, v + row, v, v);
    }
}

// persistent form of the pairwise tile: WGs loop over (row group, column tile) tiles with a grid stride
template <int ROWS> __global__ __launch_bounds__(256) void k_tiles_persistent(float *out, int n, float v) {
  const int ct = (n + 1023) / 1024, rt = (n + ROWS - 1) / ROWS;
  for (int t = blockIdx.x; t < ct * rt; t += gridDim.x) {
    const int c0 = (t % ct) * 1024, r0 = (t / ct) * ROWS;
    const int j = c0 + threadIdx.x * 4;
    for (int row = r0; row < r0 + ROWS && row < n; ++row)
      if (j + 4 <= n) *reinterpret_cast<float4 *>(out + (size_t)row * n + j) = make_float4(v, v + row, v, v);
  }
}
// column-tile-major order of the same tiles (a WG's successive tiles walk DOWN a column strip)
template <int ROWS> __global__ __launch_bounds__(256) void k_tiles_colmajor(float *out, int n, float v) {
  const int ct = (n + 1023) / 1024, rt = (n + ROWS - 1) / ROWS;
  for (int t = blockIdx.x; t < ct * rt; t += gridDim.x) {
    const int c0 = (t / rt) * 1024, r0 = (t % rt) * ROWS;
    const int j = c0 + threadIdx.x * 4;
    for (int row = r0; row < r0 + ROWS && row < n; ++row)
      if (j + 4 <= n) *reinterpret_cast<float4 *>(out + (size_t)row * n + j) = make_float4(v, v + row, v, v);
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  std::vector<float> t;
  for (int r = 0; r < 15; ++r) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const int n = 10000;
  float *out; hipMalloc(&out, sizeof(float) * (size_t)n * n);
  const double bytes = 4.0 * n * n;
  auto rep = [&](const char *name, float us) { printf("%-44s %7.1f us  %5.2f TB/s\n", name, us, bytes / us / 1e6); };
  rep("hipMemsetAsync", timeit([&] { hipMemsetAsync(out, 0, sizeof(float) * (size_t)n * n, 0); }));
#define RUN(VEC, ROWS, ADJ)                                                                           \
  rep("thread " #VEC " x float4, " #ROWS " rows/WG, adj=" #ADJ,                                          \
      timeit([&] { hipLaunchKernelGGL((k_store<VEC, ROWS, ADJ>), dim3((n + 1024 * VEC - 1) / (1024 * VEC), (n + ROWS - 1) / ROWS), \
                                      dim3(256), 0, 0, out, n, 1.0f); }))
  RUN(1, 16, false); RUN(1, 8, false); RUN(1, 32, false); RUN(1, 64, false);
  RUN(2, 16, false); RUN(2, 16, true); RUN(2, 32, false); RUN(4, 16, false); RUN(4, 8, false); RUN(4, 16, true);
#define RUNW(SEG, RPB)                                                                                 \
  rep("wave-per-row " #SEG " float4/lane, " #RPB " rows/WG",                                             \
      timeit([&] { hipLaunchKernelGGL((k_store_rowwave<SEG>), dim3((n + 256 * SEG - 1) / (256 * SEG), (n + RPB - 1) / RPB), \
                                      dim3(256), 0, 0, out, n, 1.0f, RPB); }))
  RUNW(1, 16); RUNW(2, 16); RUNW(4, 16); RUNW(4, 32); RUNW(8, 16); RUNW(10, 16); RUNW(10, 64);
  const size_t n4 = (size_t)n * n / 4;
  rep("linear grid-stride, 2048 WGs, 1 in flight", timeit([&] { hipLaunchKernelGGL(k_linear<1>, dim3(2048), dim3(256), 0, 0, (float4 *)out, n4, 1.f); }));
  rep("linear grid-stride, 2048 WGs, 4 in flight", timeit([&] { hipLaunchKernelGGL(k_linear<4>, dim3(2048), dim3(256), 0, 0, (float4 *)out, n4, 1.f); }));
  rep("linear grid-stride, 8192 WGs, 4 in flight", timeit([&] { hipLaunchKernelGGL(k_linear<4>, dim3(8192), dim3(256), 0, 0, (float4 *)out, n4, 1.f); }));
  rep("linear grid-stride, 1024 WGs, 8 in flight", timeit([&] { hipLaunchKernelGGL(k_linear<8>, dim3(1024), dim3(256), 0, 0, (float4 *)out, n4, 1.f); }));
  for (size_t kb : {64, 256, 640, 2560}) {
    const size_t chunk4 = kb * 1024 / 16;
    char nm[64]; snprintf(nm, sizeof nm, "chunked linear, %zu KB per WG", kb);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunk, dim3((unsigned)((n4 + chunk4 - 1) / chunk4)), dim3(256), 0, 0, (float4 *)out, n4, chunk4, 1.f); }));
  }
  rep("row group 16, column tiles outer", timeit([&] { hipLaunchKernelGGL(k_rowgroup<16>, dim3((n + 15) / 16), dim3(256), 0, 0, out, n, 1.f); }));
  rep("row group 4, column tiles outer", timeit([&] { hipLaunchKernelGGL(k_rowgroup<4>, dim3((n + 3) / 4), dim3(256), 0, 0, out, n, 1.f); }));
  rep("row group 16, rows outer", timeit([&] { hipLaunchKernelGGL(k_rowgroup_rowmajor<16>, dim3((n + 15) / 16), dim3(256), 0, 0, out, n, 1.f); }));
  rep("row group 4, rows outer", timeit([&] { hipLaunchKernelGGL(k_rowgroup_rowmajor<4>, dim3((n + 3) / 4), dim3(256), 0, 0, out, n, 1.f); }));
  rep("row group 1, rows outer", timeit([&] { hipLaunchKernelGGL(k_rowgroup_rowmajor<1>, dim3(n), dim3(256), 0, 0, out, n, 1.f); }));
  for (int wgs : {1024, 1536, 2048, 3072, 4096}) {
    char nm[80];
    snprintf(nm, sizeof nm, "persistent 16x1024 tiles, %d WGs", wgs);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_tiles_persistent<16>, dim3(wgs), dim3(256), 0, 0, out, n, 1.f); }));
    snprintf(nm, sizeof nm, "persistent 8x1024 tiles, %d WGs", wgs);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_tiles_persistent<8>, dim3(wgs), dim3(256), 0, 0, out, n, 1.f); }));
    snprintf(nm, sizeof nm, "persistent 16x1024 tiles col-major, %d WGs", wgs);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_tiles_colmajor<16>, dim3(wgs), dim3(256), 0, 0, out, n, 1.f); }));
  }
  for (size_t kb : {128, 200, 256, 320, 400}) {
    const size_t chunk4 = kb * 1024 / 16;
    char nm[64]; snprintf(nm, sizeof nm, "chunked linear, %zu KB per WG", kb);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_chunk, dim3((unsigned)((n4 + chunk4 - 1) / chunk4)), dim3(256), 0, 0, (float4 *)out, n4, chunk4, 1.f); }));
  }
  hipFree(out);
  return 0;
}
