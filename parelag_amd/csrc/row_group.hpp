// The row-group products' lane arithmetic, shared by the SpMV kernels (kernels_spmv.hip) and the kernels that must reproduce a row-group
// product bit for bit inside another launch (krylov.hip: cg_dir_amul_dot_kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace elag {
namespace k {
namespace rowgroup {

// the sum of a group's LPR values on lane 0 of the group
// (the others end with partial sums): the value of a butterfly of exchanges there bit for bit -- at stride o lane L adds
// lane L + o, which is its butterfly partner wherever the result still matters -- the strides inside a row of 16 lanes by DPP row shifts, the
// two across rows by gfx950's permlane swaps (~20 cycles a step where a ds_bpermute exchange of a double is ~140; these reductions end the
// dependent chain of every launch-bound product)
template <int CTRL>
__device__ __forceinline__ double dpp_shifted(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane L's partner 32 / 16 lanes up, for the lanes whose sum still matters (L < 32; L with bit 4 clear): v_permlane32_swap / v_permlane16_swap
// of a register with itself leave the other half (row pair) in the second result -- gfx950's cross-row moves, no LDS round trip
// (tools/scratch/permlane_probe.hip prints what they return)
template <int STRIDE>
__device__ __forceinline__ double lanes_up(double v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const v2u a = STRIDE == 32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const v2u b = STRIDE == 32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b.y, (int)a.y);
}
template <int LPR>
__device__ __forceinline__ double group_reduce(double v)
{
    if (LPR >= 64) v += lanes_up<32>(v);
    if (LPR >= 32) v += lanes_up<16>(v);
    if (LPR >= 16) v += dpp_shifted<0x108>(v);     // row_shl:8
    if (LPR >= 8) v += dpp_shifted<0x104>(v);      // row_shl:4
    if (LPR >= 4) v += dpp_shifted<0x102>(v);      // row_shl:2
    if (LPR >= 2) v += dpp_shifted<0x101>(v);      // row_shl:1
    return v;
}

constexpr int kRowU = 4;      // entries per lane and pass of the row-group kernels (see spmv_fused_row)

}  // namespace rowgroup
}  // namespace k
}  // namespace elag
