/*
 * ppg_launch.h — launchers of the large path kernels.  k_shade and k_tail exist in eight instantiations each (FUSED / SMALL x NEE x FULL)
 * and k_commit in six (spatial x directional filter); each pair of instantiations is its own translation unit (ppg_inst.hip compiled with
 * -DPPG_INST=n), so the library builds in parallel instead of in one 2.5-minute compile.  No device code crosses a translation unit.
 */
#ifndef PPG_LAUNCH_H
#define PPG_LAUNCH_H

#include "ppg_kernels.h"

struct ShadeLaunch {
    int grid;
    size_t lds;
    hipStream_t stream;
    PathState P; DevScene S; DevTree T; RenderParams R; Queues Q;
    int qin, small_scene;
    const unsigned int *sorted_items;
};
struct TailLaunch {
    int grid;
    size_t lds;
    hipStream_t stream;
    PathState P; DevScene S; DevTree T; RenderParams R;
    const unsigned int *dense;
    const unsigned long long *total;
    unsigned int *ticket;
    BlockStats *stats;
    int lds_tris;
    unsigned int *longest;   // the longest path (bounces) this launch finishes: what its duration is made of
    StragOut strag;          // R.defer_depth > 0: where the paths still alive at that depth go
    unsigned int lane_limit; // paths a wave takes at a time (64; fewer for a launch over a handful of stragglers)
};
struct CommitLaunch {
    int grid;
    hipStream_t stream;
    PathState P; DevTree T; RenderParams R; Queues Q;
    const unsigned char *nv8;   // k_commit_prepare's byte per path, k_commit_prepare_list's per list entry
    const unsigned int *list;
    const unsigned long long *list_n;
    float4 *splat;              // k_commit_records: the round's splat records
    unsigned int flag_shift;    // ... and the key bit of "a splat only"
};
struct SplatLaunch {
    int grid;
    hipStream_t stream;
    DevTree T;
    const unsigned long long *keys;
    const unsigned int *idx;
    const float4 *splat;
    unsigned int n, leaf_bits;
    unsigned int lds_nodes;     // D-trees of up to this many nodes are staged in LDS (<= PPG_SPLAT_NODES)
};

// variant = (FUSED ? 4 : 0) | (NEE ? 2 : 0) | (FULL ? 1 : 0)
void ppg_launch_shade(int variant, const ShadeLaunch &a);
// variant = (SMALL ? 4 : 0) | (NEE ? 2 : 0) | (FULL ? 1 : 0)
void ppg_launch_tail(int variant, const TailLaunch &a);
void ppg_launch_commit(int spatial_filter, int directional_filter, const CommitLaunch &a);
// a round of the optimiser: k_commit_records (nearest / stochastic spatial filter), then — after the sort — k_splat_sorted
void ppg_launch_commit_records(int spatial_filter, const CommitLaunch &a);
void ppg_launch_splat(int directional_filter, const SplatLaunch &a);

// one function per translation unit (pair = variant >> 1)
void ppg_launch_shade_pair0(int variant, const ShadeLaunch &a);
void ppg_launch_shade_pair1(int variant, const ShadeLaunch &a);
void ppg_launch_shade_pair2(int variant, const ShadeLaunch &a);
void ppg_launch_shade_pair3(int variant, const ShadeLaunch &a);
void ppg_launch_tail_pair0(int variant, const TailLaunch &a);
void ppg_launch_tail_pair1(int variant, const TailLaunch &a);
void ppg_launch_tail_pair2(int variant, const TailLaunch &a);
void ppg_launch_tail_pair3(int variant, const TailLaunch &a);
void ppg_launch_commit_all(int spatial_filter, int directional_filter, const CommitLaunch &a);
void ppg_launch_commit_records_all(int spatial_filter, const CommitLaunch &a);
void ppg_launch_splat_all(int directional_filter, const SplatLaunch &a);
// k_shade<false, false, FULL, MSET_COMMON> over the front part of the sorted slices (a.qin = QIN_SORTED_COMMON)
void ppg_launch_shade_common(const ShadeLaunch &a);

#endif
