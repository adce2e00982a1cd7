// Thread-local error string + ABI version + the library's own fill (see sf_fill_async below).
#include <stdarg.h>

#include "sf_common.h"

static thread_local char g_err[512] = "";

void sf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Byte fill as a KERNEL, never hipMemsetAsync: a memset node of a captured hipGraph wrote its value correctly on the first launch of the
// graph and a DIFFERENT pattern on every later one (ROCm 7.2, gfx950: 256 zero bytes came back as 0x3f800000 words from the second replay
// on - tests/test_graph_fill_gpu.py keeps the reproduction), so every scratch reset on a path that may be captured goes through here.
namespace {
__global__ void fill_kernel(unsigned* __restrict__ p, unsigned word, size_t nwords, unsigned char* tail, unsigned char byte, int ntail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) p[i] = word;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = byte;
}
}  // namespace

hipError_t sf_fill_async(void* ptr, int value, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  unsigned char* b = (unsigned char*)ptr;
  const unsigned char byte = (unsigned char)value;
  const unsigned word = 0x01010101u * byte;
  // head bytes up to a 4-byte boundary are handled as a "tail" of their own launch only when the pointer is unaligned (never, for this library's buffers)
  if ((uintptr_t)b & 3) {
    const size_t head = 4 - ((uintptr_t)b & 3) < bytes ? 4 - ((uintptr_t)b & 3) : bytes;
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, (unsigned*)nullptr, 0u, (size_t)0, b, byte, (int)head);
    b += head; bytes -= head;
    if (bytes == 0) return hipGetLastError();
  }
  const size_t nwords = bytes / 4;
  const int ntail = (int)(bytes & 3);
  const size_t want = (nwords + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, st, (unsigned*)b, word, nwords, b + nwords * 4, byte, ntail);
  return hipGetLastError();
}

extern "C" {
int sf_abi_version(void) { return SF_ABI_VERSION; }
const char* sf_last_error_string(void) { return g_err; }
}
