/*
 * ppg_inst.hip — one pair of instantiations of a large path kernel per translation unit (ppg_launch.h): compiled with
 * -DPPG_INST=0..3 (k_shade: FUSED x NEE, both FULL settings), 4..7 (k_tail: SMALL x NEE, both FULL settings), 8 (k_commit, all six),
 * 9 (k_shade<false, false, FULL, MSET_COMMON>: the common material classes of a FULL scene).
 */
#include <hip/hip_runtime.h>

#include "ppg_launch.h"

#ifndef PPG_INST
#error "compile with -DPPG_INST=0..9"
#endif

#if PPG_INST < 4
#define PAIR_F ((PPG_INST & 2) != 0)
#define PAIR_N ((PPG_INST & 1) != 0)
#define PPG_CAT2(a, b) a##b
#define PPG_CAT(a, b) PPG_CAT2(a, b)
void PPG_CAT(ppg_launch_shade_pair, PPG_INST)(int variant, const ShadeLaunch &a) {
    if (variant & 1)
        hipLaunchKernelGGL((k_shade<PAIR_F, PAIR_N, true>), dim3(a.grid), dim3(PPG_BLOCK), a.lds, a.stream, a.P, a.S, a.T, a.R, a.Q, a.qin, a.small_scene, a.sorted_items);
    else
        hipLaunchKernelGGL((k_shade<PAIR_F, PAIR_N, false>), dim3(a.grid), dim3(PPG_BLOCK), a.lds, a.stream, a.P, a.S, a.T, a.R, a.Q, a.qin, a.small_scene, a.sorted_items);
}
#elif PPG_INST < 8
#define PAIR_S (((PPG_INST - 4) & 2) != 0)
#define PAIR_N (((PPG_INST - 4) & 1) != 0)
#if PPG_INST == 4
#define PPG_TAIL_FN ppg_launch_tail_pair0
#elif PPG_INST == 5
#define PPG_TAIL_FN ppg_launch_tail_pair1
#elif PPG_INST == 6
#define PPG_TAIL_FN ppg_launch_tail_pair2
#else
#define PPG_TAIL_FN ppg_launch_tail_pair3
#endif
void PPG_TAIL_FN(int variant, const TailLaunch &a) {
    if (variant & 1)
        hipLaunchKernelGGL((k_tail<PAIR_S, PAIR_N, true>), dim3(a.grid), dim3(PPG_BLOCK), a.lds, a.stream, a.P, a.S, a.T, a.R, a.dense, a.total, a.ticket, a.stats, a.lds_tris, a.longest, a.strag, a.lane_limit);
    else
        hipLaunchKernelGGL((k_tail<PAIR_S, PAIR_N, false>), dim3(a.grid), dim3(PPG_BLOCK), a.lds, a.stream, a.P, a.S, a.T, a.R, a.dense, a.total, a.ticket, a.stats, a.lds_tris, a.longest, a.strag, a.lane_limit);
}
#elif PPG_INST == 9
void ppg_launch_shade_common(const ShadeLaunch &a) {
    hipLaunchKernelGGL((k_shade<false, false, true, MSET_COMMON>), dim3(a.grid), dim3(PPG_BLOCK), a.lds, a.stream, a.P, a.S, a.T, a.R, a.Q, a.qin, a.small_scene, a.sorted_items);
}
#else
void ppg_launch_commit_all(int sf, int df, const CommitLaunch &a) {
#define PPG_COMMIT(SFV, DFV) hipLaunchKernelGGL((k_commit<SFV, DFV>), dim3(a.grid), dim3(PPG_BLOCK), 0, a.stream, a.P, a.T, a.R, a.Q, a.nv8, a.list, a.list_n)
    if (sf == SF_NEAREST && df == DF_NEAREST) PPG_COMMIT(SF_NEAREST, DF_NEAREST);
    else if (sf == SF_NEAREST) PPG_COMMIT(SF_NEAREST, DF_BOX);
    else if (sf == SF_STOCHASTIC && df == DF_NEAREST) PPG_COMMIT(SF_STOCHASTIC, DF_NEAREST);
    else if (sf == SF_STOCHASTIC) PPG_COMMIT(SF_STOCHASTIC, DF_BOX);
    else if (df == DF_NEAREST) PPG_COMMIT(SF_BOX, DF_NEAREST);
    else PPG_COMMIT(SF_BOX, DF_BOX);
#undef PPG_COMMIT
}
void ppg_launch_commit_records_all(int sf, const CommitLaunch &a) {
    if (sf == SF_STOCHASTIC)
        hipLaunchKernelGGL((k_commit_records<SF_STOCHASTIC>), dim3(a.grid), dim3(PPG_BLOCK), 0, a.stream, a.P, a.T, a.R, a.Q, a.nv8, a.list, a.list_n, a.splat, a.flag_shift);
    else
        hipLaunchKernelGGL((k_commit_records<SF_NEAREST>), dim3(a.grid), dim3(PPG_BLOCK), 0, a.stream, a.P, a.T, a.R, a.Q, a.nv8, a.list, a.list_n, a.splat, a.flag_shift);
}
void ppg_launch_splat_all(int df, const SplatLaunch &a) {
    if (df == DF_BOX) hipLaunchKernelGGL((k_splat_sorted<DF_BOX>), dim3(a.grid), dim3(PPG_BLOCK), 0, a.stream, a.T, a.keys, a.idx, a.splat, a.n, a.leaf_bits, a.lds_nodes);
    else hipLaunchKernelGGL((k_splat_sorted<DF_NEAREST>), dim3(a.grid), dim3(PPG_BLOCK), 0, a.stream, a.T, a.keys, a.idx, a.splat, a.n, a.leaf_bits, a.lds_nodes);
}
#endif
