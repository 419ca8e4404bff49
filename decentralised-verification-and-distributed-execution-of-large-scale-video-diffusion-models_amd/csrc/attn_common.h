// attn_common.h — index helpers shared by the attention kernels (flash.hip, attn.hip).
#pragma once
#include "vdx_common.h"

#define NEG_BIG (-1.0e30f)

// key (0..31) held by accumulator register `reg` of lane-half `h` when K rows are fed through pi()
__device__ __forceinline__ int acc_key(int reg, int h) {
    return 16 * (reg >> 3) + 8 * h + 4 * ((reg >> 2) & 1) + (reg & 3);
}
// A-operand row i must carry key pi(i) = i with bits 2 and 3 swapped
__device__ __forceinline__ int pi_row(int i) {
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

