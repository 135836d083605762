// Input side of the path for gfx950: uint8 HWC image -> square resize -> ToTensor ->
// Normalize -> model dtype, NCHW, ready for the patch-embedding GEMM. Replaces, per image, the reference's host-side
// SquareResize (datasets/transforms.py:223-231 = PIL.Image.resize((S, S), BILINEAR)), ToTensor and Normalize
// (datasets/transforms.py:437-443, datasets/coco.py:127-130, deploy/benchmark.py:273-281).
//
// The resize is Pillow's, bit for bit (src/libImaging/Resample.c): separable triangle filter whose support grows with the
// down-scale factor (antialiasing), coefficients in 22-bit fixed point, accumulator seeded with 1 << 21, result
// clip8(acc >> 22), horizontal pass first into a uint8 intermediate, then the vertical pass. The coefficient tables
// (bounds + fixed-point taps per output column / row) are computed on the host in double exactly as precompute_coeffs does
// and shared by all images of one size. ToTensor + Normalize is a 3 x 256 float table (built with the reference's own f32
// operations) looked up by the final uint8 value: exact by construction. Pure byte / integer work, HBM-bound:
// one lane per output pixel, x fastest, so the vertical pass reads and the planar stores are fully coalesced.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[y][x][c] (H x S x 3 uint8) from src[y][.][c]
__global__ __launch_bounds__(256) void resize_h_kernel(const lwdetr_resize_image* __restrict__ imgs, const int32_t* __restrict__ tab,
                                                       uint8_t* __restrict__ tmp, int S) {
    const lwdetr_resize_image im = imgs[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= S || y >= im.height || im.width == S) return;      // width == S: no horizontal pass (read in place by the vertical pass)
    const int xmin = tab[im.xbounds_off + 2 * x], n = tab[im.xbounds_off + 2 * x + 1];
    const int32_t* k = tab + im.xcoef_off + (long)x * im.xksize;
    const uint8_t* row = im.src + (long)y * im.row_stride + (long)xmin * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        a0 += row[3 * t] * c; a1 += row[3 * t + 1] * c; a2 += row[3 * t + 2] * c;
    }
    uint8_t* o = tmp + im.tmp_off + ((long)y * S + x) * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// vertical pass + ToTensor + Normalize: out[b][c][y][x]. One lane = 4 consecutive output pixels (12 source bytes per tap
// row as three aligned 32-bit loads when the row start allows, 8-byte stores per plane). An image whose width already is S
// skips the horizontal pass (as Pillow does; the identity tap would reproduce the bytes) and is read in place.
template <typename T>
__global__ __launch_bounds__(256) void resize_v_kernel(const lwdetr_resize_image* __restrict__ imgs, const int32_t* __restrict__ tab,
                                                       const uint8_t* __restrict__ tmp, const float* __restrict__ lut,
                                                       T* __restrict__ out, int S) {
    __shared__ float slut[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) slut[i] = lut[i];
    __syncthreads();
    const lwdetr_resize_image im = imgs[blockIdx.z];
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= S) return;
    const int ymin = tab[im.ybounds_off + 2 * y], n = tab[im.ybounds_off + 2 * y + 1];
    const int32_t* k = tab + im.ycoef_off + (long)y * im.yksize;
    const bool direct = im.width == S;
    const long pitch = direct ? im.row_stride : (long)S * 3;
    const uint8_t* col = (direct ? im.src : tmp + im.tmp_off) + (long)ymin * pitch + (long)x0 * 3;
    const int npx = S - x0 < 4 ? S - x0 : 4;
    int acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 1 << (PRECISION_BITS - 1);
    const bool wide = npx == 4 && (((uintptr_t)col | (uintptr_t)pitch) & 3) == 0;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        const uint8_t* px = col + (long)t * pitch;
        if (wide) {
            const uint32_t* p4 = (const uint32_t*)px;
            const uint32_t w0 = p4[0], w1 = p4[1], w2 = p4[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += (int)((w0 >> (8 * i)) & 255) * c;
                acc[4 + i] += (int)((w1 >> (8 * i)) & 255) * c;
                acc[8 + i] += (int)((w2 >> (8 * i)) & 255) * c;
            }
        } else {
            for (int i = 0; i < 3 * npx; ++i) acc[i] += px[i] * c;
        }
    }
    T* o = out + (long)blockIdx.z * 3 * S * S + (long)y * S + x0;
    typedef T V4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        T v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = from_f32<T>(slut[ch * 256 + clip8(acc[3 * i + ch])]);
        T* oc = o + (long)ch * S * S;
        if (npx == 4 && (((uintptr_t)oc) & (4 * sizeof(T) - 1)) == 0) *(V4*)oc = V4{v[0], v[1], v[2], v[3]};
        else for (int i = 0; i < npx; ++i) oc[i] = v[i];
    }
}

}  // namespace

extern "C" int lwdetr_resize_normalize(const lwdetr_resize_image* images, int B, int max_height, const int32_t* tables,
                                       uint8_t* tmp, const float* lut, void* out, int S, int dtype, void* hip_stream) {
    if (!images || !tables || !tmp || !lut || !out || B < 0 || S <= 0 || max_height <= 0 || max_height > 65535 || S > 65535)
        return LWDETR_ERR_BAD_ARG;
    if (B == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    const unsigned gx = (unsigned)((S + 255) / 256), gx4 = (unsigned)((S + 1023) / 1024);
    ProfScope ps(KID_ELTWISE, 0.0, 0.0, st);
    hipLaunchKernelGGL(resize_h_kernel, dim3(gx, (unsigned)max_height, (unsigned)B), dim3(256), 0, st, images, tables, tmp, S);
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL((resize_v_kernel<float>), dim3(gx4, (unsigned)S, (unsigned)B), dim3(256), 0, st, images, tables, tmp, lut, (float*)out, S); break;
        case DT_F16: hipLaunchKernelGGL((resize_v_kernel<f16>), dim3(gx4, (unsigned)S, (unsigned)B), dim3(256), 0, st, images, tables, tmp, lut, (f16*)out, S); break;
        case DT_BF16: hipLaunchKernelGGL((resize_v_kernel<bf16>), dim3(gx4, (unsigned)S, (unsigned)B), dim3(256), 0, st, images, tables, tmp, lut, (bf16*)out, S); break;
        default: return LWDETR_ERR_UNSUPPORTED;
    }
    return lwdetr_check_launch();
}
