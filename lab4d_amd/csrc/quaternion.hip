// gfx950 replacement for the dqtorch CUDA extension (lab4d/third_party/quaternion/src/*.cu).
//
// All of these are HBM-bound element-wise kernels (32-48 B per row): one thread per row, whole
// rows moved as 16-byte vectors when the row is 4 x fp32, grid-stride over rows, launched on the
// caller's stream.  Semantics follow quaternion.cu:29-217 and matinv.cu:42-240; accumulation is
// done in the storage type for f32/f64 and in f32 for f16.
#include "common.hpp"

namespace lab4d {

template <typename T> struct Acc { using type = T; };
template <> struct Acc<__half> { using type = float; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T* p) { return (typename Acc<T>::type)(*p); }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <typename T, typename A> __device__ __forceinline__ void st(T* p, A v) { *p = (T)v; }
template <> __device__ __forceinline__ void st<__half, float>(__half* p, float v) { *p = __float2half(v); }

template <typename T, typename A>
__device__ __forceinline__ void load_q(const T* p, uint32_t D, A& w, A& x, A& y, A& z) {
  if (D == 3) {
    w = A(0); x = ld(p); y = ld(p + 1); z = ld(p + 2);
  } else {
    if constexpr (sizeof(T) == 4) {  // one 16-byte load
      const float4 v = *reinterpret_cast<const float4*>(p);
      w = v.x; x = v.y; y = v.z; z = v.w;
    } else {
      w = ld(p); x = ld(p + 1); y = ld(p + 2); z = ld(p + 3);
    }
  }
}
template <typename T, typename A>
__device__ __forceinline__ void store_q(T* p, uint32_t D, A w, A x, A y, A z) {
  if (D == 3) {
    st(p, x); st(p + 1, y); st(p + 2, z);
  } else {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(w, x, y, z);
    } else {
      st(p, w); st(p + 1, x); st(p + 2, y); st(p + 3, z);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_qmul_fwd(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o,
                                                    uint32_t B, uint32_t D1, uint32_t D2) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A aw, ax, ay, az, bw, bx, by, bz;
    load_q<T, A>(a + (size_t)r * D1, D1, aw, ax, ay, az);
    load_q<T, A>(b + (size_t)r * D2, D2, bw, bx, by, bz);
    store_q<T, A>(o + (size_t)r * 4, 4,
                  aw * bw - ax * bx - ay * by - az * bz,
                  aw * bx + ax * bw + ay * bz - az * by,
                  aw * by - ax * bz + ay * bw + az * bx,
                  aw * bz + ax * by - ay * bx + az * bw);
  }
}

// d(out)/d(a), d(out)/d(b) contracted with grad: ga = grad (x) conj(b), gb = conj(a) (x) grad
template <typename T>
__global__ void __launch_bounds__(256) k_qmul_bwd(const T* __restrict__ g, uint32_t B, uint32_t D1, uint32_t D2,
                                                    const T* __restrict__ a, const T* __restrict__ b,
                                                    T* __restrict__ ga, T* __restrict__ gb) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A aw, ax, ay, az, bw, bx, by, bz, g0, g1, g2, g3;
    load_q<T, A>(a + (size_t)r * D1, D1, aw, ax, ay, az);
    load_q<T, A>(b + (size_t)r * D2, D2, bw, bx, by, bz);
    load_q<T, A>(g + (size_t)r * 4, 4, g0, g1, g2, g3);
    store_q<T, A>(ga + (size_t)r * D1, D1,
                  g0 * bw + g1 * bx + g2 * by + g3 * bz,
                  -g0 * bx + g1 * bw - g2 * bz + g3 * by,
                  -g0 * by + g1 * bz + g2 * bw - g3 * bx,
                  -g0 * bz - g1 * by + g2 * bx + g3 * bw);
    store_q<T, A>(gb + (size_t)r * D2, D2,
                  g0 * aw + g1 * ax + g2 * ay + g3 * az,
                  -g0 * ax + g1 * aw + g2 * az - g3 * ay,
                  -g0 * ay - g1 * az + g2 * aw + g3 * ax,
                  -g0 * az + g1 * ay - g2 * ax + g3 * aw);
  }
}

// Second order (quaternion.cu:128-199).  With da = grad_out_1, db = grad_out_2 (cotangents of
// grad_in1/grad_in2):  grad_grad = da (x) b + a (x) db ;  gg_a = conj-products of (db, grad).
template <typename T>
__global__ void __launch_bounds__(256) k_qmul_bwd_bwd(const T* __restrict__ go1, const T* __restrict__ go2, uint32_t B,
                                                        uint32_t D1, uint32_t D2, const T* __restrict__ g,
                                                        const T* __restrict__ a, const T* __restrict__ b,
                                                        T* __restrict__ gg, T* __restrict__ gga, T* __restrict__ ggb) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A aw, ax, ay, az, bw, bx, by, bz, g0, g1, g2, g3, daw, dax, day, daz, dbw, dbx, dby, dbz;
    load_q<T, A>(a + (size_t)r * D1, D1, aw, ax, ay, az);
    load_q<T, A>(b + (size_t)r * D2, D2, bw, bx, by, bz);
    load_q<T, A>(go1 + (size_t)r * D1, D1, daw, dax, day, daz);
    load_q<T, A>(go2 + (size_t)r * D2, D2, dbw, dbx, dby, dbz);
    load_q<T, A>(g + (size_t)r * 4, 4, g0, g1, g2, g3);
    store_q<T, A>(gga + (size_t)r * D1, D1,
                  dbw * g0 + dbx * g1 + dby * g2 + dbz * g3,
                  dbw * g1 - dbx * g0 + dby * g3 - dbz * g2,
                  dbw * g2 - dbx * g3 - dby * g0 + dbz * g1,
                  dbw * g3 + dbx * g2 - dby * g1 - dbz * g0);
    store_q<T, A>(ggb + (size_t)r * D2, D2,
                  daw * g0 + dax * g1 + day * g2 + daz * g3,
                  daw * g1 - dax * g0 - day * g3 + daz * g2,
                  daw * g2 + dax * g3 - day * g0 - daz * g1,
                  daw * g3 - dax * g2 + day * g1 - daz * g0);
    store_q<T, A>(gg + (size_t)r * 4, 4,
                  daw * bw + dbw * aw - dax * bx - dbx * ax - day * by - dby * ay - daz * bz - dbz * az,
                  daw * bx + dbw * ax + dax * bw + dbx * aw + day * bz - dby * az - daz * by + dbz * ay,
                  daw * by + dbw * ay - dax * bz + dbx * az + day * bw + dby * aw + daz * bx - dbz * ax,
                  daw * bz + dbw * az + dax * by - dbx * ay - day * bx + dby * ax + daz * bw + dbz * aw);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_qconj(const T* __restrict__ in, uint32_t B, T* __restrict__ out) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A w, x, y, z;
    load_q<T, A>(in + (size_t)r * 4, 4, w, x, y, z);
    store_q<T, A>(out + (size_t)r * 4, 4, w, -x, -y, -z);
  }
}

// ---- 3x3 ----
template <typename T, typename A> __device__ __forceinline__ void load9(const T* p, A* m) {
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = ld(p + i);
}
template <typename A> __device__ __forceinline__ A det9(const A* m) {
  return m[0] * m[4] * m[8] + m[3] * m[7] * m[2] + m[6] * m[5] * m[1] - m[2] * m[4] * m[6] - m[5] * m[7] * m[0] -
         m[8] * m[3] * m[1];
}
template <typename A> __device__ __forceinline__ void adj9(const A* m, A s, A* o) {
  o[0] = s * (m[4] * m[8] - m[5] * m[7]);
  o[1] = s * (m[2] * m[7] - m[1] * m[8]);
  o[2] = s * (m[1] * m[5] - m[2] * m[4]);
  o[3] = s * (m[5] * m[6] - m[3] * m[8]);
  o[4] = s * (m[0] * m[8] - m[2] * m[6]);
  o[5] = s * (m[2] * m[3] - m[0] * m[5]);
  o[6] = s * (m[3] * m[7] - m[4] * m[6]);
  o[7] = s * (m[1] * m[6] - m[0] * m[7]);
  o[8] = s * (m[0] * m[4] - m[1] * m[3]);
}

// mode 0: det -> out (B); mode 1: adjugate/scales -> out (B,9); mode 2: inverse -> out (B,9), scales (B)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_mat3(const T* __restrict__ in, const T* __restrict__ scales_in,
                                               T* __restrict__ out, T* __restrict__ scales_out, uint32_t B) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A m[9], o[9];
    load9<T, A>(in + (size_t)r * 9, m);
    if constexpr (MODE == 0) {
      st(out + r, det9(m));
    } else {
      A d;
      if constexpr (MODE == 1) d = ld(scales_in + r);
      else { d = det9(m); st(scales_out + r, d); }
      adj9(m, A(1) / d, o);
#pragma unroll
      for (int i = 0; i < 9; ++i) st(out + (size_t)r * 9 + i, o[i]);
    }
  }
}

// grad_in = -(H^T G H^T) with H = inverse (matinv.cu:119-240)
template <typename T>
__global__ void __launch_bounds__(256) k_mat3_inv_bwd(const T* __restrict__ grad, const T* __restrict__ inv,
                                                       T* __restrict__ gin, uint32_t B) {
  using A = typename Acc<T>::type;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gridDim.x * blockDim.x) {
    A g[9], h[9];
    load9<T, A>(grad + (size_t)r * 9, g);
    load9<T, A>(inv + (size_t)r * 9, h);
    // t = H^T G   (t[i][j] = sum_k h[k][i] g[k][j])
    A t[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) t[i * 3 + j] = h[0 * 3 + i] * g[0 * 3 + j] + h[1 * 3 + i] * g[1 * 3 + j] + h[2 * 3 + i] * g[2 * 3 + j];
    // out = -(t H^T)  (out[i][j] = -sum_k t[i][k] h[j][k])
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        st(gin + (size_t)r * 9 + i * 3 + j, -(t[i * 3 + 0] * h[j * 3 + 0] + t[i * 3 + 1] * h[j * 3 + 1] + t[i * 3 + 2] * h[j * 3 + 2]));
  }
}

static inline int rows_grid(uint32_t B) {
  int g = div_up(B, 256);
  return g < 1 ? 1 : (g > 4096 ? 4096 : g);  // 256 CUs x 16 blocks, grid-stride beyond that
}

#define LAB4D_DISPATCH(dtype, ...)                                      \
  switch (dtype) {                                                      \
    case LAB4D_F32: { using T = float; __VA_ARGS__; break; }            \
    case LAB4D_F16: { using T = __half; __VA_ARGS__; break; }           \
    case LAB4D_F64: { using T = double; __VA_ARGS__; break; }           \
    default: set_error("unsupported dtype code %d", dtype); return LAB4D_EINVAL; \
  }

}  // namespace lab4d

using namespace lab4d;

extern "C" int lab4d_quaternion_mul_forward(const void* in1, const void* in2, void* out, uint32_t B, uint32_t D1,
                                            uint32_t D2, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(in1 && in2 && out, "quaternion_mul_forward: null pointer");
  LAB4D_REQUIRE((D1 == 3 || D1 == 4) && (D2 == 3 || D2 == 4), "quaternion_mul_forward: D1,D2 must be 3 or 4 (got %u,%u)", D1, D2);
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_qmul_fwd<T>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)in1, (const T*)in2, (T*)out, B, D1, D2));
  return check_launch("quaternion_mul_forward");
}

extern "C" int lab4d_quaternion_mul_backward(const void* grad, uint32_t B, uint32_t D1, uint32_t D2, const void* in1,
                                             const void* in2, void* g1, void* g2, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(grad && in1 && in2 && g1 && g2, "quaternion_mul_backward: null pointer");
  LAB4D_REQUIRE((D1 == 3 || D1 == 4) && (D2 == 3 || D2 == 4), "quaternion_mul_backward: D1,D2 must be 3 or 4");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_qmul_bwd<T>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)grad, B, D1, D2, (const T*)in1, (const T*)in2, (T*)g1, (T*)g2));
  return check_launch("quaternion_mul_backward");
}

extern "C" int lab4d_quaternion_mul_backward_backward(const void* go1, const void* go2, uint32_t B, uint32_t D1,
                                                      uint32_t D2, const void* grad, const void* in1, const void* in2,
                                                      void* gg, void* gga, void* ggb, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(go1 && go2 && grad && in1 && in2 && gg && gga && ggb, "quaternion_mul_backward_backward: null pointer");
  LAB4D_REQUIRE((D1 == 3 || D1 == 4) && (D2 == 3 || D2 == 4), "quaternion_mul_backward_backward: D1,D2 must be 3 or 4");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_qmul_bwd_bwd<T>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)go1, (const T*)go2, B, D1, D2, (const T*)grad, (const T*)in1,
                                           (const T*)in2, (T*)gg, (T*)gga, (T*)ggb));
  return check_launch("quaternion_mul_backward_backward");
}

extern "C" int lab4d_quaternion_conjugate(const void* in, uint32_t B, void* out, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(in && out, "quaternion_conjugate: null pointer");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_qconj<T>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)in, B, (T*)out));
  return check_launch("quaternion_conjugate");
}

extern "C" int lab4d_mat3x3_det_forward(const void* in, void* out, uint32_t B, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(in && out, "mat3x3_det_forward: null pointer");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_mat3<T, 0>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)in, (const T*)nullptr, (T*)out, (T*)nullptr, B));
  return check_launch("mat3x3_det_forward");
}
extern "C" int lab4d_mat3x3_scale_adjoint_forward(const void* in, const void* scales, void* out, uint32_t B, int dtype,
                                                  void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(in && scales && out, "mat3x3_scale_adjoint_forward: null pointer");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_mat3<T, 1>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)in, (const T*)scales, (T*)out, (T*)nullptr, B));
  return check_launch("mat3x3_scale_adjoint_forward");
}
extern "C" int lab4d_mat3x3_inv_forward(const void* in, void* out, void* out_scales, uint32_t B, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(in && out && out_scales, "mat3x3_inv_forward: null pointer");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_mat3<T, 2>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)in, (const T*)nullptr, (T*)out, (T*)out_scales, B));
  return check_launch("mat3x3_inv_forward");
}
extern "C" int lab4d_mat3x3_inv_backward(const void* grad, const void* inv, void* gin, uint32_t B, int dtype, void* stream) {
  if (B == 0) return LAB4D_OK;  // empty batch: nothing to do (pointers may be null)
  LAB4D_REQUIRE(grad && inv && gin, "mat3x3_inv_backward: null pointer");
  LAB4D_DISPATCH(dtype, hipLaunchKernelGGL((k_mat3_inv_bwd<T>), dim3(rows_grid(B)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)grad, (const T*)inv, (T*)gin, B));
  return check_launch("mat3x3_inv_backward");
}
