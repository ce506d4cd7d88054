// write_calib.hip — what rocprofv3's WRITE_SIZE (and FETCH_SIZE) report on gfx950 for the store widths the match kernels use.
// MI355X_MICROARCH.md calibrates only 16-B-per-lane streaming stores ("other access widths are uncalibrated: calibrate on
// a known byte count in your own access pattern").  Each kernel writes exactly `bytes` bytes; run under
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- tools/write_calib
// and compare the counter (KB) with the byte count.  Build: hipcc -O2 --offload-arch=gfx950 tools/write_calib.hip -o tools/write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void store16(uint4* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(1, 2, 3, 4); }
__global__ void store8(uint2* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint2(1, 2); }
__global__ void store4(uint32_t* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 1u; }
// the final-set rows: one wavefront per row of `row_words` dwords (even), 8-byte stores, rows back to back (1 192 B: not line-aligned)
__global__ void rows8(uint32_t* p, uint32_t n_rows, uint32_t row_words) {
  const uint32_t row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (row >= n_rows) return;
  uint2* r = reinterpret_cast<uint2*>(p + (size_t)row * row_words);
  for (uint32_t w = lane; w < row_words / 2u; w += 64u) r[w] = make_uint2(w, row);
}
// the any-match words: lane = stream, one dword per stream and call, `stride` dwords apart; `words` calls one after the other
__global__ void scattered4(uint32_t* p, uint32_t n_streams, uint32_t stride, uint32_t word) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_streams) p[(size_t)s * stride + word] = s;
}
// accept events: 12 bytes per lane, back to back
struct Ev { uint32_t a, b, c; };
__global__ void store12(Ev* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { Ev e{1, 2, 3}; p[i] = e; } }

// the input windows: lane = 4 * slot + part reads bytes [64 * chunk + 16 * part, + 16) of stream `slot` (rows `pitch` bytes apart),
// 16 streams per wave-load, as rx_sym_pack_kernel's load_win does; every byte of n_streams x pitch is read once
__global__ void read_win(const uint8_t* p, uint32_t n_streams, uint32_t pitch, uint32_t* sink) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  const uint32_t slot = wave * 16u + (lane >> 2), part = lane & 3u;
  if (slot >= n_streams) return;
  uint32_t acc = 0;
  for (uint32_t chunk = 0; chunk < pitch / 64u; chunk++) {
    const uint4 q = *reinterpret_cast<const uint4*>(p + (size_t)slot * pitch + chunk * 64u + part * 16u);
    acc ^= q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void read16(const uint4* p, size_t n, uint32_t* sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) { const uint4 q = p[i]; if ((q.x ^ q.y ^ q.z ^ q.w) == 0x12345678u) sink[0] = 1; }
}

int main() {
  const size_t bytes = 64u << 20;
  void* d;
  CHECK(hipMalloc(&d, 256u << 20));
  CHECK(hipMemset(d, 0, 256u << 20));
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(store16, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, (uint4*)d, bytes / 16);
  hipLaunchKernelGGL(store8, dim3((unsigned)(bytes / 8 / 256)), dim3(256), 0, 0, (uint2*)d, bytes / 8);
  hipLaunchKernelGGL(store4, dim3((unsigned)(bytes / 4 / 256)), dim3(256), 0, 0, (uint32_t*)d, bytes / 4);
  hipLaunchKernelGGL(store12, dim3((unsigned)(bytes / 12 / 256 + 1)), dim3(256), 0, 0, (Ev*)d, bytes / 12);
  const uint32_t n_rows = 65536, row_words = 298;  // snort_16: 2 * ceil(9514 / 64)
  hipLaunchKernelGGL(rows8, dim3(n_rows / 4), dim3(256), 0, 0, (uint32_t*)d, n_rows, row_words);
  for (uint32_t w = 0; w < 33; w++)  // 65 536 streams x 33 any-match words = 8.65 MB in all
    hipLaunchKernelGGL(scattered4, dim3(65536 / 256), dim3(256), 0, 0, (uint32_t*)d, 65536u, 33u, w);
  // reads (run with --pmc FETCH_SIZE): 64 MiB as the pack kernel's input windows, then as a plain 16-B-per-lane stream; the
  // buffer was last written by the kernels above and is far larger than L2, the 256 MiB memset in between evicts the rest
  CHECK(hipMemset((char*)d + (128u << 20), 0, 128u << 20));
  hipLaunchKernelGGL(read_win, dim3(65536 / 16 / 4), dim3(256), 0, 0, (const uint8_t*)d, 65536u, 1024u, (uint32_t*)d + (255u << 18));
  CHECK(hipMemset((char*)d + (128u << 20), 1, 128u << 20));
  hipLaunchKernelGGL(read16, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, (const uint4*)d, bytes / 16, (uint32_t*)d + (255u << 18));
  CHECK(hipDeviceSynchronize());
  printf("bytes written: store16/8/4 %zu each, store12 %zu, rows8 %zu, scattered4 %u per launch x 33 launches\n", bytes, bytes / 12 * 12,
         (size_t)n_rows * row_words * 4, 65536u * 4u);
  return 0;
}
