/* Subset_kernels_cuda -- /root/reference/include_kernels_cuda/subset_kernels_cuda.h:33-56 (five overloads) */
#ifndef SUBSET_KERNELS_CUDA_H
#define SUBSET_KERNELS_CUDA_H
#include "rrx_forward.h"

namespace Subset_kernels_cuda
{
    inline void scatter_(const int ncol, const int nlay, const int nbnd, const int ncol_in, const int col_s_in, const int n,
                         Float* const* full, const Float* const* sub)
    { RRX_CALL(rrx_get_from_subset, ncol, nlay, nbnd, ncol_in, col_s_in, n, full, sub); }

    inline void get_from_subset(const int ncol, const int nbnd, const int ncol_in, const int col_s_in, Float* var_full, const Float* var_sub)
    { Float* f[1] = {var_full}; const Float* s[1] = {var_sub}; scatter_(ncol, 1, nbnd, ncol_in, col_s_in, 1, f, s); }

    inline void get_from_subset(const int ncol, const int nlay, const int ncol_in, const int col_s_in,
            Float* var1_full, Float* var2_full, Float* var3_full, Float* var4_full,
            const Float* var1_sub, const Float* var2_sub, const Float* var3_sub, const Float* var4_sub)
    { Float* f[4] = {var1_full, var2_full, var3_full, var4_full}; const Float* s[4] = {var1_sub, var2_sub, var3_sub, var4_sub};
      scatter_(ncol, nlay, 1, ncol_in, col_s_in, 4, f, s); }

    inline void get_from_subset(const int ncol, const int nlay, const int ncol_in, const int col_s_in,
            Float* var1_full, Float* var2_full, Float* var3_full,
            const Float* var1_sub, const Float* var2_sub, const Float* var3_sub)
    { Float* f[3] = {var1_full, var2_full, var3_full}; const Float* s[3] = {var1_sub, var2_sub, var3_sub};
      scatter_(ncol, nlay, 1, ncol_in, col_s_in, 3, f, s); }

    inline void get_from_subset(const int ncol, const int nlay, const int nbnd, const int ncol_in, const int col_s_in,
            Float* var1_full, Float* var2_full, Float* var3_full, Float* var4_full,
            const Float* var1_sub, const Float* var2_sub, const Float* var3_sub, const Float* var4_sub)
    { Float* f[4] = {var1_full, var2_full, var3_full, var4_full}; const Float* s[4] = {var1_sub, var2_sub, var3_sub, var4_sub};
      scatter_(ncol, nlay, nbnd, ncol_in, col_s_in, 4, f, s); }

    inline void get_from_subset(const int ncol, const int nlay, const int nbnd, const int ncol_in, const int col_s_in,
            Float* var1_full, Float* var2_full, Float* var3_full,
            const Float* var1_sub, const Float* var2_sub, const Float* var3_sub)
    { Float* f[3] = {var1_full, var2_full, var3_full}; const Float* s[3] = {var1_sub, var2_sub, var3_sub};
      scatter_(ncol, nlay, nbnd, ncol_in, col_s_in, 3, f, s); }
}
#endif
