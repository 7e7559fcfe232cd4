// bf16 helpers of the matrix-core kernels (maf_forward_bf16.hip, maf_train_bf16.hip): round-to-nearest-even conversion
// and v_mfma_f32_16x16x32_bf16 -- A: lane l holds row l & 15, k = 8 (l >> 4) .. + 7; B: column l & 15, the same k;
// C: column l & 15, rows 4 (l >> 4) + r.
#ifndef PMC_BF16_H
#define PMC_BF16_H

#include "maf_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace fbf {

__device__ __forceinline__ unsigned short to_bf16(float v) {           // round to nearest even
    const unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float from_bf16(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ f32x4 mfma_bf(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b),
                                                   c, 0, 0, 0);
}

}  // namespace fbf

#endif
