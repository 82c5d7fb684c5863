// Batch ingestion on the device: one launch gathers the sampled pixels of a whole batch of frames from the
// HBM-resident frame cache.  Contract and reference citations: include/lab4d_ingest.h; arithmetic: ingest_math.hpp.
// Latency-bound gather (a training batch is 256 frames x 16 pixels: ~50 scattered reads per pixel, 0.4 MB out), so the kernel
// is shaped for parallelism, not bandwidth: one thread per (frame, pixel, feature channel); the channel-0..4 threads of a pixel
// also move one of the other modalities each, so no thread serialises more than four dependent loads.
#include "common.hpp"
#include "ingest_math.hpp"

namespace lab4d {
using namespace lab4d_ingest;

template <bool IMG16, bool FLOW16, bool FEAT16>
__global__ void __launch_bounds__(256) k_ingest_gather(const int64_t* __restrict__ frame_ptrs, const int* __restrict__ xy, int M, int N, int H, int W,
                                                       int FR, int FC, void* __restrict__ rgb, uint8_t* __restrict__ mask, uint8_t* __restrict__ vis2d,
                                                       void* __restrict__ depth, float* __restrict__ flow, float* __restrict__ flow_uct,
                                                       float* __restrict__ feature, float* __restrict__ hxy) {
  const long total = (long)M * N * FC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / FC;  // pixel index m*N + n
    const int c = (int)(i - p * FC);
    const int m = (int)(p / N);
    const int px = xy[2 * p], py = xy[2 * p + 1];
    const int64_t* fp = frame_ptrs + 5 * (long)m;
    const long pix = (long)py * W + px;
    feature[i] = bilinear_channel<FEAT16>((const void*)fp[4], FR, FC, c, px, py, H);
    // the other modalities ride on the first channel threads of the pixel (FC >= 5 is checked by the host)
    if (c == 0) {
      if (IMG16) {
        const uint16_t* s = (const uint16_t*)fp[0] + pix * 3;
        uint16_t* d = (uint16_t*)rgb + p * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
      } else {
        const float* s = (const float*)fp[0] + pix * 3;
        float* d = (float*)rgb + p * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
      }
    } else if (c == 1) {
      const uint8_t* s = (const uint8_t*)fp[1] + pix * 2;
      mask[p] = s[0];
      vis2d[p] = s[1];
    } else if (c == 2) {
      if (IMG16) ((uint16_t*)depth)[p] = ((const uint16_t*)fp[2])[pix];
      else ((float*)depth)[p] = ((const float*)fp[2])[pix];
    } else if (c == 3) {
      float f[3];
      if (FLOW16) {
        const uint16_t* s = (const uint16_t*)fp[3] + pix * 3;
        f[0] = half_to_float(s[0]); f[1] = half_to_float(s[1]); f[2] = half_to_float(s[2]);
      } else {
        const float* s = (const float*)fp[3] + pix * 3;
        f[0] = s[0]; f[1] = s[1]; f[2] = s[2];
      }
      flow[2 * p] = f[0]; flow[2 * p + 1] = f[1];
      flow_uct[p] = f[2];
    } else if (c == 4) {
      hxy[3 * p] = (float)px; hxy[3 * p + 1] = (float)py; hxy[3 * p + 2] = 1.0f;
    }
  }
}

}  // namespace lab4d

using namespace lab4d;

extern "C" int lab4d_ingest_gather(const int64_t* frame_ptrs, const int32_t* xy, int M, int N, int H, int W, int FR, int FC, int img_dtype,
                                   int flow_dtype, int feat_dtype, void* rgb, uint8_t* mask, uint8_t* vis2d, void* depth, float* flow,
                                   float* flow_uct, float* feature, float* hxy, void* stream) {
  LAB4D_REQUIRE(frame_ptrs && xy && rgb && mask && vis2d && depth && flow && flow_uct && feature && hxy, "ingest_gather: null pointer");
  LAB4D_REQUIRE(M >= 0 && N >= 0 && H > 0 && W > 0 && FR >= 2 && FC >= 5 && FC <= 1024, "ingest_gather: bad sizes M=%d N=%d H=%d W=%d FR=%d FC=%d", M, N, H, W, FR, FC);
  auto ok = [](int d) { return d == LAB4D_F16 || d == LAB4D_F32; };
  LAB4D_REQUIRE(ok(img_dtype) && ok(flow_dtype) && ok(feat_dtype), "ingest_gather: dtypes must be LAB4D_F16 or LAB4D_F32");
  if (M == 0 || N == 0) return LAB4D_OK;
  const long total = (long)M * N * FC;
  long g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  hipStream_t st = (hipStream_t)stream;
#define GO(A, B, C) hipLaunchKernelGGL((k_ingest_gather<A, B, C>), dim3((int)g), dim3(256), 0, st, frame_ptrs, xy, M, N, H, W, FR, FC, rgb, mask, vis2d, depth, flow, flow_uct, feature, hxy)
  const int key = (img_dtype == LAB4D_F16 ? 4 : 0) | (flow_dtype == LAB4D_F16 ? 2 : 0) | (feat_dtype == LAB4D_F16 ? 1 : 0);
  switch (key) {
    case 7: GO(true, true, true); break;
    case 6: GO(true, true, false); break;
    case 5: GO(true, false, true); break;
    case 4: GO(true, false, false); break;
    case 3: GO(false, true, true); break;
    case 2: GO(false, true, false); break;
    case 1: GO(false, false, true); break;
    default: GO(false, false, false); break;
  }
#undef GO
  return check_launch("ingest_gather");
}
