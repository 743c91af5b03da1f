// Latency-bound kernels of the path rebuilt on the register-resident MFMA block (ltg_rgemm.h): one memory round trip per
// workgroup instead of one per K tile.  Included by ltg_kernels.hip inside its anonymous namespace (after AdamC, PairView,
// DropView, DLayout, dact, Workspace, Probe).  Reference citations are relative to /root/reference/.
//
//   generator middle layers   fk_enc1, fk_dec0 (reparameterisation + KL folded into its operand loader), fk_dz, fk_dh1
//   encoder layer 0           fk_enc0_fwd / fk_enc0_grad: column-blocked, 8 gathered rows in flight per wave
//   discriminator             fk_d_l1, fk_d_l2 (output unit folded in: per-tile partial dot products), fk_d_y,
//                             fk_d_bwd1 / fk_d_bwd2 (gradient slabs), fk_d_adam (flat float4 sweep)
//   small item counts         fk_dec1, fk_row_dlogits (softmax statistics + losses + dlogits of a row in one pass),
//                             fk_dh2, fk_g_tail (the Adam updates of the generator as jobs of ONE launch)
#pragma once

#include "ltg_rgemm.h"

// Workgroups b, b + 8, b + 16 ... share an XCD (and its L2) under the observed round-robin placement: give each XCD a CONTIGUOUS
// run of the n tile ids, so that tiles of neighbouring rows (same A rows, all of B) meet in one L2 instead of all eight.
// Speed only: any placement computes the same result.  Bijective for every n.
__device__ __forceinline__ int xcd_chunk(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// The same for a 2-D grid of (column tile, row tile) workgroups: blockIdx -> tile ids with the ROW tiles of one column tile -- which share
// the column block of the weight matrix, the large operand of these latency-bound products -- consecutive, so that they meet in one XCD's L2.
// Dealt round-robin every row tile of a column fetched that block into another L2: fk_dec1 moved 21 MB per launch for 3 MB of operands,
// fk_enc1 10 MB for 1.2 (profiles/r4_pmc_traffic.json).  Round 5, Askubuntu_Sample G step with this order in fk_g_tail, fk_dec1, fk_dh2:
// 81.8 -> 78.9 ms per epoch on one box.
struct LtgTile2 {
    int x, y;   // column tile, row tile
};
__device__ __forceinline__ LtgTile2 xcd_tile2() {
    const int t = xcd_chunk((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    return LtgTile2{t / (int)gridDim.y, t % (int)gridDim.y};
}

// a * b rounded to fp32 on its own: never contracted into an fma with a following addition (HIP compiles with
// -ffp-contract=fast-honor-pragmas; __fmul_rn is a plain product there)
__device__ __forceinline__ float ltg_mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

typedef LtgRg<1, 1, 1, 1, 4> Rg16;    // 16 x 16 tile, four K slices
typedef LtgRg<2, 2, 1, 1, 4> Rg32k;   // 32 x 32 tile, four K slices (each wave the whole tile)
typedef LtgRg<2, 2, 1, 1, 8> Rg32k8;  // the same over EIGHT K slices (512 threads): fk_d_l2
typedef LtgRg<1, 1, 2, 2, 1> Rg32;    // 32 x 32 tile, one 16 x 16 per wave over the whole K

#include "ltg_fast_gen.h"
#include "ltg_fast_disc.h"
#include "ltg_fast_small.h"
#include "ltg_fast_fp8.h"
