// Regrouping of parameters in one launch.  The recurrent cells and the attention layers of the reference keep their weights as several
// nn.Parameters that the fused kernels want as ONE matrix (ConvGRUCell: conv_zr over [x ; h] + conv_h1 + conv_h2 -> an x-part and an h-part with
// rows [z | r | n]; axial attention: to_q / to_kv of both axes -> one projection).  Built with torch.cat / slices that is 9 small launches per
// cell and step forward and as many backward (slice gradients are zero-filled full-size tensors that autograd then adds); here it is one launch
// each way: up to SF_MAX_BLOCKS two-dimensional fp32 block copies (or zero fills) described by a table passed by value.
#include "sf_common.h"

namespace {

struct BlockTable { sfBlock b[SF_MAX_BLOCKS]; };

__global__ __launch_bounds__(256) void copy_blocks_kernel(const BlockTable t) {
  const sfBlock& b = t.b[blockIdx.y];
  const long long n = (long long)b.rows * b.cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long r = i / b.cols, c = i - r * b.cols;
    const float v = b.src ? b.src[r * b.src_stride + c] : 0.0f;
    if (b.transpose) b.dst[c * b.dst_stride + r] = v;   // (block-uniform; reads stay coalesced, the strided writes are a few KB of weights)
    else b.dst[r * b.dst_stride + c] = v;
  }
}

}  // namespace

extern "C" int sf_copy_blocks(const sfBlock* blocks, int32_t n, sfStream stream) {
  SF_REQUIRE(n >= 0 && (n == 0 || blocks), "sf_copy_blocks: null table");
  for (int i0 = 0; i0 < n; i0 += SF_MAX_BLOCKS) {
    BlockTable t{};
    const int m = n - i0 < SF_MAX_BLOCKS ? n - i0 : SF_MAX_BLOCKS;
    long long biggest = 0;
    for (int i = 0; i < m; ++i) {
      const sfBlock& b = blocks[i0 + i];
      SF_REQUIRE(b.dst && b.rows >= 0 && b.cols >= 0 && b.dst_stride >= (b.transpose ? b.rows : b.cols) && (!b.src || b.src_stride >= b.cols),
                 "sf_copy_blocks: block %d: rows %lld, cols %lld, strides %lld / %lld", i0 + i, (long long)b.rows, (long long)b.cols,
                 (long long)b.src_stride, (long long)b.dst_stride);
      t.b[i] = b;
      const long long e = (long long)b.rows * b.cols;
      biggest = e > biggest ? e : biggest;
    }
    if (biggest == 0) continue;
    const long long want = (biggest + 1023) / 1024;  // ~4 elements per thread
    const unsigned gx = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, t);
    SF_CHECK_LAUNCH("copy_blocks");
  }
  return 0;
}
