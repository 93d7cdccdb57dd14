// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in THIS repo's access
// patterns (MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE under-reports wide coalesced reads and
// other widths are uncalibrated): a streaming copy with 8-byte-per-lane and with 4-byte-per-lane
// accesses over a buffer far larger than the 256 MiB Infinity Cache.  Known bytes: n * sizeof(T)
// read and written.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <class T> __global__ void k_copy(const T* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

extern "C" int calib_copy(const void* src, void* dst, int64_t n, int elem_bytes, void* stream) {
  if (elem_bytes == 8)
    hipLaunchKernelGGL((k_copy<double>), dim3(4096), dim3(256), 0, (hipStream_t)stream,
                       (const double*)src, (double*)dst, n);
  else
    hipLaunchKernelGGL((k_copy<float>), dim3(4096), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, (float*)dst, n);
  return (int)hipGetLastError();
}
