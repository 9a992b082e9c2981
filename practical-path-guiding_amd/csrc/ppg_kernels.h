/*
 * ppg_kernels.h — wavefront kernels of the guided path tracer (included by ppg_hip.hip).
 *
 * One render pass (= one BlockedRenderProcess of the reference, GP:1087-1106) is
 *     k_generate → [k_trace → k_shade]* → (k_tail) → k_commit → k_film, and per round of passes (sort →) k_adam_apply
 * over SoA path state.  Path i = j * n_pix + k is sample j of the k-th owned pixel, so a wave holds 64
 * neighbouring pixels.  Queues are arrays of path indices compacted with one wave-aggregated atomic;
 * all accumulation into shared state is integer (fixed point), so results do not depend on the order
 * in which waves run.  k_shade(b) finishes bounce b-1 (GP:2078-2145) with the hit found by k_trace and
 * starts bounce b (GP:1902-2040).
 */
#ifndef PPG_KERNELS_H
#define PPG_KERNELS_H

#include "ppg_device.h"

#define PPG_MAX_VERTICES 32  // MAX_NUM_VERTICES, GP:1771
#define PPG_BLOCK 256
#ifndef PPG_SHADE_WAVES
#define PPG_SHADE_WAVES 4  // waves per SIMD requested for k_shade: 128 VGPRs without spilling (5 or 6 spill and are slower, 2-3 waste occupancy)
#endif
#ifndef PPG_LEAF_VOTE
#define PPG_LEAF_VOTE 16  // k_trace: lanes holding a leaf wait until this many lanes of the wave do (0 = test leaves at once).  KITCHEN 720p,
                          // k_trace over 127 passes: 0 → 486 ms, 8 → 412, 16 → 387, 32 → 390
#endif
#ifndef PPG_SHADE_WAVES_FULL
#define PPG_SHADE_WAVES_FULL 3  // ... for the FULL material set (k_shade<.., FULL>, k_tail<.., FULL>), whose BSDF code needs more registers: at 128 VGPRs it
                                // spills 257 of them (KITCHEN 720p: 4 waves 76.5, 3 waves 81.3, 2 waves 79.6 Msamples/s)
#endif

#ifndef PPG_TAIL_WAVES_FULL
#define PPG_TAIL_WAVES_FULL PPG_SHADE_WAVES_FULL  // ... for k_tail<.., FULL> on its own
#endif

#ifndef PPG_SHADE_WAVES_COMMON
#define PPG_SHADE_WAVES_COMMON 3  // ... for k_shade<.., FULL, MSET_COMMON>: the common lobes of a FULL scene (see MSET_COMMON)
#endif

// Material subsets of the FULL kernels.  A FULL scene's queue slices are counting-sorted by the BSDF at the new hit (k_sort_slices); the
// COMMON classes — diffuse / two-sided diffuse, smooth and rough conductor, plastic, rough plastic on a triangle, no mask, no bump map —
// come first and are shaded by k_shade<.., FULL, MSET_COMMON>, a variant from which everything else is compiled out: analytic spheres, rays
// that left the scene (environment lookup), null components and the look-through trace, glass / thin glass / rough dielectric lobes, the
// bump-mapped frame.  The rest of the slice goes through the complete kernel (MSET_ALL) in a second launch.  Same arithmetic per path.
enum { MSET_ALL = 0, MSET_COMMON = 1 };
D bool mset_common_type(int type) {
    return type == PPG_BSDF_DIFFUSE || type == PPG_BSDF_TWOSIDED_DIFFUSE || type == PPG_BSDF_MIRROR || type == PPG_BSDF_CONDUCTOR ||
           type == PPG_BSDF_ROUGHCONDUCTOR || type == PPG_BSDF_PLASTIC || type == PPG_BSDF_ROUGHPLASTIC;
}

enum { NEE_NEVER = 0, NEE_KICKSTART = 1, NEE_ALWAYS = 2 };
enum { SF_NEAREST = 0, SF_STOCHASTIC = 1, SF_BOX = 2 };
enum { DF_NEAREST = 0, DF_BOX = 1 };
enum { LOSS_NONE = 0, LOSS_KL = 1, LOSS_VAR = 2 };

// flags word of a path
#define FL_DEPTH_MASK 0x0007ffffu  // rRec.depth
#define FL_PEND_NULL (1u << 19)    // the sampled bounce was a null (pass-through) interaction, GP:2045-2075
#define FL_NV_SHIFT 20             // nVertices (0..32), 6 bits
#define FL_NV_MASK (0x3fu << FL_NV_SHIFT)
#define FL_SCATTERED (1u << 26)
#define FL_EMITTED_OK (1u << 27)   // rRec.type & EEmittedRadiance
#define FL_PENDING (1u << 28)      // a bounce was sampled; its trace result is consumed by the next k_shade
#define FL_PEND_TREE (1u << 29)    // that bounce had a D-tree
#define FL_PEND_DELTA (1u << 30)   // that bounce sampled a delta component
#define FL_PEND_REFN (1u << 31)    // dot(ray.d, dRec.refN) >= 0 at the vertex that sampled the bounce (AreaLight::pdfDirect, area.cpp:175-183)

struct RenderParams {
    int nee, spatial_filter, directional_filter, loss;
    float bsdf_sampling_fraction;
    int rr_depth, max_depth, strict_normals, hide_emitters;
    int spp;              // sppPerPass
    int is_final_iter, do_nee;
    unsigned long long seed;
    unsigned int pass_index;  // m_passesRendered at the start of this pass
    unsigned int pass_index_spp;  // sample index of the batch's first sample = pass_index * sppPerPass
    // Final iteration (nothing is recorded; include/ppg.h "Final iteration: groups of passes"): the launch holds whole GROUPS of passes, group k
    // of the launch = samples [k * group_samples, (k + 1) * group_samples) of a pixel, whose sample indices start at
    // pass_index_spp + k * group_stride (a rank of a sharded render holds every world-th group: stride = world * group_samples).  0: one run.
    unsigned int group_samples, group_stride;
    int max_vertices;         // vertex slots allocated per path
    unsigned int img_pixels;  // width * height of the whole image (path ids of the Adam records)
    // k_tail: a path still alive when its rRec.depth has reached this value leaves the launch as a STRAGGLER — its state goes to the compact
    // straggler set (StragOut) and a second k_tail finishes it beside the next batch (ppg_hip.hip "Stragglers").  0: every path runs to its end.
    unsigned int defer_depth;
#ifdef PPG_PROBE
    unsigned long long *probe;  // development builds only (make EXTRA=-DPPG_PROBE): cycle sums per section of a LONE path's bounce in k_tail
#endif
};

// Development probe (make EXTRA=-DPPG_PROBE OUT=../lib/libppg_hip_probe.so): where do the cycles of a lone path's bounce go?  k_tail marks
// the sections of every loop iteration in which its wave holds exactly ONE live path; s_memtime after a full s_waitcnt, differences summed
// in LDS (fire-and-forget ds_add) and flushed to RenderParams::probe when the workgroup ends.  Not compiled into the product.
#ifdef PPG_PROBE
#define PPG_PROBE_SLOTS 24
#define PROBE_MARK(cs_, seg) do { if ((cs_) && (cs_)->pon) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
        atomicAdd(&(cs_)->plds[seg], t_ - (cs_)->pt); (cs_)->pt = t_; } } while (0)
#else
#define PROBE_MARK(cs_, seg) do { } while (0)
#endif

// A field of the per-path (or per-vertex-slot) state: element i lives at base[i * stride].  stride 1 = one array per field (SoA, the
// default: a wave of NEIGHBOURING paths reads full lines — the first bounces, where most of the rays are); PPG_PATH_LAYOUT=aos interleaves
// the fields of one path in one 128-byte record (stride 8; 4 / 6 for vertex slots), which suits the late bounces — slices compacted and
// sorted by BSDF type, lane k holds an arbitrary path, six 16-byte accesses cost six sectors in six lines (KITCHEN: k_shade moves
// 970 + 642 B per ray through HBM, profiles/r02_pmc_traffic_kitchen.json) — and costs the early ones as much as it gains.
template <typename T>
struct Field {
    T *base;
    unsigned int stride;
    D T &operator[](size_t i) const { return base[i * stride]; }
    D explicit operator bool() const { return base != nullptr; }
};

struct PathState {
    unsigned int n_paths;  // n_pix * spp
    unsigned int n_pix;    // owned pixels
    const unsigned int *pixels;  // owned pixel list (row-major pixel indices)
    Field<float4> ray_o;   // (o, mint)
    Field<float4> ray_d;   // (d, maxt)
    Field<float4> thr;     // (throughput, eta)
    Field<float4> li;      // (Li, woPdf of the pending bounce)
    Field<float4> hit;     // (t, u, v, prim)
    Field<uint4> misc;     // (key, dim, flags, leaf of the pending bounce)
    // vertex slots, index = slot * n_paths + path
    Field<float4> v_d;     // (ray.d, woPdf)
    Field<float4> v_thr;   // (throughput, bsdfPdf)
    Field<float4> v_bsdf;  // (bsdfVal, dTreePdf)
    Field<float4> v_rad;   // (radiance, leaf | delta << 31)
    Field<float4> v_o;     // (ray.o, -)        only if spatial filter != nearest
    Field<float4> v_vox;   // (voxel size, -)   only if spatial filter != nearest
    float *nee_cos;  // dot(ray.d, dRec.refN) of the pending bounce (-2 when refN = 0): ConstantBackgroundEmitter::pdfDirect needs the
                     // value, not just its sign; only allocated for scenes with an environment emitter
    // The compact state of a batch's STRAGGLERS (ppg_hip.hip "Stragglers"): path j of this state is path orig[j] of the batch it was taken
    // from, whose n_pix / pixels stay in force — the Adam records it leaves carry that path's id and the "deferred" bit.  nullptr: a batch.
    const unsigned int *orig;
};

// Where k_tail puts a straggler (RenderParams::defer_depth): one interleaved record of eight float4 per path — ray origin, direction,
// throughput, Li, hit, (key, dim, flags, leaf), two unused — at position j = atomic count, and the path's index in its batch.
struct StragOut {
    float4 *rec;
    unsigned int *orig;
    unsigned long long *count;
};

// Per-workgroup statistics (zeroed per ppg_render_passes, summed on the host).  A single global counter
// would take one same-address atomic per wave and round — measured at ~11 ns each that was half of
// k_shade's run time — so every persistent workgroup owns a private record and a private slice of the queues.
struct BlockStats {
    unsigned long long rays, path_len, committed;
    unsigned long long bvh_nodes, bvh_tris;  // k_trace<.., COUNT>: BVH4 nodes visited / triangles tested (kernel timing runs only: the roofline's n, t)
    unsigned long long max_len;              // longest path finished by k_tail (diagnostics: PPG_DEBUG_BATCH)
    unsigned long long shade_common;         // rays k_sort_slices dealt to k_shade<.., MSET_COMMON> (the units of its roofline line)
};

// Queues.  Every wavefront bounce reads the DENSE list of live paths (items[1], dense_n entries), dealt to the persistent workgroups in
// chunks of PPG_DCHUNK, and k_shade appends the survivors to its workgroup's own slice of items[0] (entries [b * cap, b * cap + count[0][b]):
// compaction needs only an LDS counter, no global atomics).  k_scan_counts + k_gather_slices then rebuild the dense list — two tiny launches
// per bounce.  (Until round 3 a workgroup kept its slice for the whole batch; once fewer paths are alive than the GPU has lanes, every one
// of the 4096 workgroups then still holds a few paths and every launch costs several rounds of almost empty workgroups — KITCHEN 720p, one pass
// in flight: k_trace took 0.5 ms per bounce whether 920 k or 300 k rays were left.)
struct Queues {
    unsigned int *items[2];  // [0]: k_shade's output slices; [1]: the dense list
    unsigned int *count[2];  // [0][b]: entries of output slice b; [1][b]: entries of workgroup b's slice of the sorted list (k_sort_slices)
    unsigned int *n_common;  // [b]: how many of them, at the front, belong to the COMMON material classes (MSET_COMMON); nullptr: no split
    unsigned int cap;        // entries per workgroup slice
    unsigned int n_blocks;
    BlockStats *stats;       // [n_blocks]
    const unsigned long long *dense_n;  // entries of the dense list
    const unsigned int *stop;           // != 0: the batch has gone to the persistent threads (k_tail); wavefront kernels launched after that return at once
};

// what a wavefront kernel reads (its `qin` argument)
#define QIN_FIRST (-1)   // bounce 1: every path of the batch, dealt in chunks of PPG_CHUNK (a wave = 64 neighbouring pixels)
#define QIN_DENSE (-2)   // the dense list, dealt in chunks of PPG_DCHUNK
#define QIN_SORTED (-3)  // k_shade after k_sort_slices: the workgroup's share of the dense list, re-ordered, in its slice of `sorted_items`
#define QIN_SORTED_COMMON (-4)  // ... its front part: the hits on the common material classes (k_shade<.., MSET_COMMON>)
#define QIN_SORTED_REST (-5)    // ... the rest of it (the complete k_shade, launched second: it APPENDS to the first launch's output slice)
#ifndef PPG_DCHUNK
#define PPG_DCHUNK 256
#endif

// wave-aggregated append to the workgroup's queue slice: one LDS atomic per wave
D unsigned int queue_append(unsigned int *lds_counter, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return 0;
    int leader = __ffsll((long long)mask) - 1;
    int lane = threadIdx.x & 63;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(lds_counter, (unsigned int)__popcll(mask));
    base = __shfl(base, leader);
    return base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
}

// workgroup-wide sum of a per-thread count into the workgroup's private statistics word (no global atomics)
D void block_add_u64(unsigned long long *lds_acc, unsigned long long *dst, unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (threadIdx.x == 0) *lds_acc = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(lds_acc, v);
    __syncthreads();
    if (threadIdx.x == 0 && *lds_acc) atomicAdd(dst, *lds_acc);  // atomic: k_tail and k_commit may run side by side on two streams
}

// number of paths workgroup b handles at the first bounce and its k-th path (chunks dealt round-robin)
// (chunks of one wave: 64 neighbouring pixels; dealing small chunks round-robin spreads empty image regions
// evenly over the workgroups, whose queue slices then shrink at the same rate)
#define PPG_CHUNK 64
D unsigned int first_share(unsigned int n_paths, unsigned int b, unsigned int nb) {
    unsigned int chunks = (n_paths + PPG_CHUNK - 1) / PPG_CHUNK;
    unsigned int mine = chunks > b ? (chunks - b + nb - 1) / nb : 0;
    return mine * PPG_CHUNK;  // the last chunk may be partial: callers test i < n_paths
}
D unsigned int first_path(unsigned int k, unsigned int b, unsigned int nb) { return ((k / PPG_CHUNK) * nb + b) * PPG_CHUNK + (k % PPG_CHUNK); }


// the same dealing of the dense list, in chunks of PPG_DCHUNK (one workgroup's worth: with fewer live paths than lanes the busy workgroups
// are full and the others return at once)
D unsigned int dense_share(unsigned int n, unsigned int b, unsigned int nb) {  // exact: the last chunk of the list may be partial
    const unsigned int chunks = (n + PPG_DCHUNK - 1) / PPG_DCHUNK;
    const unsigned int mine = chunks > b ? (chunks - b + nb - 1) / nb : 0;
    if (mine == 0) return 0;
    const unsigned int last = (mine - 1) * nb + b;  // this workgroup's last chunk
    const unsigned int tail = n - last * PPG_DCHUNK;
    return (mine - 1) * PPG_DCHUNK + (tail < PPG_DCHUNK ? tail : PPG_DCHUNK);
}
D unsigned int dense_index(unsigned int k, unsigned int b, unsigned int nb) { return ((k / PPG_DCHUNK) * nb + b) * PPG_DCHUNK + (k % PPG_DCHUNK); }

// The work of workgroup b in a wavefront kernel: `count` positions, position k holds path work_item(k).
struct Work {
    const unsigned int *items;
    unsigned int count;
    unsigned int n_paths;  // mode 0: paths of the batch
    int mode;  // 0: first bounce (work_item returns ~0 beyond the batch: the last chunk may be partial), 1: items[k], 2: items[dense_index(k)]
};
D Work work_of(const PathState &P, const Queues &Q, int qin, const unsigned int *sorted_items, unsigned int b, unsigned int nb) {
    Work w;
    w.n_paths = P.n_paths;
    if (qin == QIN_FIRST) { w.items = nullptr; w.count = first_share(P.n_paths, b, nb); w.mode = 0; }
    else if (qin == QIN_SORTED && sorted_items) { w.items = sorted_items + (size_t)b * Q.cap; w.count = Q.count[1][b]; w.mode = 1; }
    else if (qin == QIN_SORTED_COMMON && sorted_items) { w.items = sorted_items + (size_t)b * Q.cap; w.count = Q.n_common[b]; w.mode = 1; }
    else if (qin == QIN_SORTED_REST && sorted_items) { const unsigned int nc = Q.n_common[b]; w.items = sorted_items + (size_t)b * Q.cap + nc; w.count = Q.count[1][b] - nc; w.mode = 1; }
    else { w.items = Q.items[1]; w.count = dense_share((unsigned int)*Q.dense_n, b, nb); w.mode = 2; }
    return w;
}
D unsigned int work_item(const Work &w, unsigned int k, unsigned int b, unsigned int nb) {
    if (w.mode == 0) {
        const unsigned int idx = first_path(k, b, nb);
        return idx < w.n_paths ? idx : 0xffffffffu;
    }
    return w.items[w.mode == 2 ? dense_index(k, b, nb) : k];
}

// Keyed accumulation with wave-level pre-combination: lanes of a wave that add to the same key are summed
// in registers and issue ONE atomic (up to ROUNDS distinct keys are combined, the rest go out directly).
// Early iterations send every record of a wave to the same D-tree (iteration 0: all of them to one
// statistical-weight counter — 15 M same-address atomics per pass at 720p without this).  Integer adds
// commute, so the combination does not change the result.  Must be called by all lanes of the wave.
template <int ROUNDS>
D void wave_key_add(unsigned long long *dst, unsigned int key, unsigned long long val, bool active) {
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int r = 0; r < ROUNDS && todo; ++r) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned int k = __shfl(key, leader);
        const bool mine = active && key == k;
        const unsigned long long grp = __ballot(mine);
        const unsigned long long vl = __shfl(val, leader);
        unsigned long long total;
        if (__ballot(mine && val != vl) == 0) {
            total = vl * (unsigned long long)__popcll(grp);  // all equal (unit weights): no reduction needed
        } else {
            unsigned long long v = mine ? val : 0ull;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            total = v;
        }
        if (lane == leader) atomicAdd(dst + k, total);
        todo &= ~grp;
        if (mine) active = false;
    }
    if (active) atomicAdd(dst + key, val);
}

// Closest hit over the triangles of a small scene held in LDS, no BVH: every lane reads the same triangle (LDS broadcast),
// no dependent memory chain; same (t, original index) minimum as the BVH.  The records are grouped by projection axis
// (DevScene::accel_small), one loop per axis with the ray components permuted once.
D Hit trace_small(const float4 *lds_tris, const DevScene &S, F3 o, F3 d, float rayMint, float maxt) {
    float rayMinT = rayMint;
    if (rayMinT == PPG_EPSILON)  // adaptive ray epsilon, skdtree.cpp:125-129
        rayMinT *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
    Hit h;
    h.t = __builtin_inff(); h.u = 0; h.v = 0; h.prim = -1;
    int bestOrig = 0x7fffffff;
    const float4 *Tk = lds_tris;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        // triaccel.h:113-140: k = 0 → (u, v, k) = (y, z, x); k = 1 → (z, x, y); k = 2 → (x, y, z)
        const float o_u = axis == 0 ? o.y : (axis == 1 ? o.z : o.x), o_v = axis == 0 ? o.z : (axis == 1 ? o.x : o.y);
        const float o_k = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
        const float d_u = axis == 0 ? d.y : (axis == 1 ? d.z : d.x), d_v = axis == 0 ? d.z : (axis == 1 ? d.x : d.y);
        const float d_k = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
        const int n = S.small_n[axis];
        for (int k = 0; k < n; ++k, Tk += 3) {
            float tt, uu, vv;
            // t == h.t still passes: ties go to the smaller original index
            if (tri_hit_axis(Tk, o_u, o_v, o_k, d_u, d_v, d_k, rayMinT, fminf(maxt, h.t), tt, uu, vv)) {
                const float4 a2 = Tk[2];
                const int orig = __float_as_int(a2.w);
                if (tt < h.t || (tt == h.t && orig < bestOrig)) { h.t = tt; h.u = uu; h.v = vv; h.prim = __float_as_int(a2.z); bestOrig = orig; }
            }
        }
    }
    return h;
}

// ------------------------------------------------------------------------------------------------
// k_generate — renderBlock's sample loop head (GP:1613-1630) + PerspectiveCamera::sampleRayDifferential
// ------------------------------------------------------------------------------------------------
// FUSED (small scenes): the camera ray is traced right here from the LDS copy of the scene.
template <bool FUSED>
__global__ __launch_bounds__(PPG_BLOCK) void k_generate(PathState P, DevScene S, RenderParams R, Queues Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned long long acc;
    const float4 *lds_tris = (const float4 *)lds_raw;
    if (FUSED) {
        for (int k = threadIdx.x; k < 3 * S.n_tris; k += blockDim.x) ((float4 *)lds_raw)[k] = S.accel_small[k];
        __syncthreads();
    }
    unsigned int traced = 0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_paths; i += gridDim.x * blockDim.x) {
        unsigned int k = i % P.n_pix, j = i / P.n_pix;
        unsigned int pixel = P.pixels[k];
        const unsigned int sample = R.group_samples ? R.pass_index_spp + (j / R.group_samples) * R.group_stride + (j % R.group_samples) : R.pass_index_spp + j;
        unsigned int key = ppg_path_key(R.seed, pixel, sample);
        unsigned int dim = 0;
        float u1 = ppg_rand(key, dim++);
        float u2 = ppg_rand(key, dim++);
        int px = (int)(pixel % (unsigned int)S.cam.width), py = (int)(pixel / (unsigned int)S.cam.width);
        float sx = (float)px + u1, sy = (float)py + u2;  // GP:1620
        F3 nearP = xf_point(S.cam.s2c, f3(sx * S.cam.inv_w, sy * S.cam.inv_h, 0.0f));
        F3 dl = norm3(nearP);
        float invZ = 1.0f / dl.z;
        float mint = S.cam.near_clip * invZ, maxt = S.cam.far_clip * invZ;
        F3 o = f3(S.cam.c2w[3], S.cam.c2w[7], S.cam.c2w[11]);
        F3 d = xf_vec(S.cam.c2w, dl);
        if (FUSED) {
            Hit h = trace_small(lds_tris, S, o, d, mint, maxt);
            P.hit[i] = make_float4(h.t, h.u, h.v, __int_as_float(h.prim));
            ++traced;
        } else {
            P.ray_o[i] = make_float4(o.x, o.y, o.z, mint);
        }
        P.ray_d[i] = make_float4(d.x, d.y, d.z, maxt);
        P.thr[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        P.li[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        P.misc[i] = make_uint4(key, dim, 1u | FL_EMITTED_OK, 0u);
    }
    if (FUSED) block_add_u64(&acc, &Q.stats[blockIdx.x % Q.n_blocks].rays, traced);
}

// ------------------------------------------------------------------------------------------------
// k_trace — Scene::rayIntersect (skdtree.cpp:112-142) for every queued path
// ------------------------------------------------------------------------------------------------
// SMALL: the whole scene (<= 64 triangles) is tested from LDS without a BVH — every lane reads the same
// triangle (LDS broadcast), no divergence, no dependent memory chain.  Same (t, original index) minimum.
// stage the scene cache of LdsScene (first n_nodes BVH nodes, then n_tris triangles) into `lds_raw`
D LdsScene stage_scene(const DevScene &S, unsigned char *lds_raw, int lds_nodes, int lds_tris) {
    LdsScene L;
    float4 *dst = (float4 *)lds_raw;
    const float4 *srcN = (const float4 *)S.bvh;
    const int nN = lds_nodes * 4, nT = lds_tris * 3;
    for (int k = threadIdx.x; k < nN; k += blockDim.x) dst[k] = srcN[k];
    for (int k = threadIdx.x; k < nT; k += blockDim.x) dst[nN + k] = S.accel_small[k];  // small scenes only (grouped by axis)
    __syncthreads();
    L.nodes = (const BvhNode *)lds_raw; L.tris = dst + nN; L.n_nodes = lds_nodes; L.n_tris = lds_tris;
    return L;
}

// BVH4 traversal of one queue slice with ray replacement: a lane whose ray is finished immediately takes the next
// entry of the slice (LDS ticket), so a wave stays full until the slice is empty instead of idling until its
// longest ray is done — secondary rays are incoherent and their traversal lengths differ by an order of magnitude.
#ifndef PPG_TRACE_PAIRS
#define PPG_TRACE_PAIRS 1  // k_trace: the leaf phase of a wave tests compacted (ray, triangle) pairs (leaf_pairs, ppg_device.h); 0 = a leaf per lane
#endif
#ifndef PPG_TRACE_STACK
#define PPG_TRACE_STACK (PPG_TRACE_PAIRS ? 12 : 16)  // rows of k_trace's LDS stack columns (with the pair scratch: 19 KB a workgroup, eight to a CU)
#endif
#if PPG_TRACE_PAIRS
// The traversal with (ray, triangle)-pair compaction.  Lanes stay in the loop until the whole wave is out of rays (a lane without a ray
// still tests other lanes' triangles); per iteration the wave first runs its leaf phase if the vote passes, then the node step of every
// lane holding an interior node — a lane that popped a node after its leaf takes that step in the same iteration.
template <bool COUNT>
D void trace_slice_bvh4(const PathState &P, const DevScene &S, int *lds_stack, unsigned int *ticket, const Work &work,
                        unsigned int b, unsigned int nb, unsigned int &traced, unsigned long long &n_nodes, unsigned long long &n_tris) {
    const unsigned int count = work.count;
    if (threadIdx.x == 0) *ticket = 0;
    __syncthreads();
    PairLds *W = reinterpret_cast<PairLds *>(lds_stack + PPG_TRACE_STACK * PPG_BLOCK) + (threadIdx.x >> 6);
    TStack st;
    int st_over[48 - PPG_TRACE_STACK];
    st.over = st_over;
    st.lds = lds_stack + threadIdx.x; st.stride = PPG_BLOCK; st.sp = 0; st.cap = PPG_TRACE_STACK;
    bool have = false, done = false;
    unsigned int w_nodes = 0, w_tris = 0;  // COUNT: this WAVE's node steps and triangle tests (uniform: the timed kernel keeps the registers of the other)
    unsigned int i = 0;
    F3 o = f3s(0.0f), d = f3s(0.0f), id = f3s(0.0f);
    float mint = 0, maxt = 0;
    Hit best;
    best.t = 0; best.u = 0; best.v = 0; best.prim = -1;
    int bestOrig = 0, cur = PPG_BVH4_EMPTY;
    for (;;) {
        if (!have && !done) {
            unsigned int k = atomicAdd(ticket, 1u);
            if (k >= count) done = true;
            else {
                i = work_item(work, k, b, nb);
                if (i < P.n_paths) {
                    float4 ro = P.ray_o[i], rd = P.ray_d[i];
                    o = f3(ro.x, ro.y, ro.z); d = f3(rd.x, rd.y, rd.z);
                    mint = ro.w; maxt = rd.w;
                    if (mint == PPG_EPSILON)  // adaptive ray epsilon
                        mint *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
                    id = f3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
                    best.t = __builtin_inff(); best.u = 0; best.v = 0; best.prim = -1;
                    bestOrig = 0x7fffffff; cur = 0; st.sp = 0;
                    have = true;
                }
            }
        }
        if (__ballot(!done) == 0ull) break;
        const bool isLeaf = have && cur < 0;
        const int code = ~cur;
        const int first = code >> 3, cnt = isLeaf ? (code & 7) + 1 : 0;
        unsigned int off, total;
        pair_prefix(cnt, off, total);
        if (total >= PPG_PAIR_VOTE || (total != 0u && __ballot(have && cur >= 0) == 0ull)) {  // (uniform)
            if (COUNT) w_tris += total;
            leaf_pairs(W, S.accel, first, cnt, off, total, o, d, mint, fminf(maxt, best.t), best, bestOrig);
            if (isLeaf) cur = st.sp > 0 ? st.pop() : PPG_BVH4_EMPTY;
        }
        const bool nodeStep = have && cur >= 0 && cur != PPG_BVH4_EMPTY;
        if (COUNT) w_nodes += (unsigned int)__popcll(__ballot(nodeStep));
        if (nodeStep) {
            const Bvh4Hits hc = bvh4_children(S.bvh4 + cur, o, id, mint, fminf(maxt, best.t));
            if (hc.m > 0) {
                st.push_children(hc.m, hc.c1, hc.c2, hc.c3);
                cur = hc.c0;
            } else cur = st.sp > 0 ? st.pop() : PPG_BVH4_EMPTY;
        }
        if (have && cur == PPG_BVH4_EMPTY) {  // stack empty: this ray is done
            if (S.n_spheres) sphere_pass<false>(S, o, d, mint, maxt, best);
            P.hit[i] = make_float4(best.t, best.u, best.v, __int_as_float(best.prim));
            ++traced;
            have = false;
        }
    }
    if (COUNT && (threadIdx.x & 63) == 0) { n_nodes += w_nodes; n_tris += w_tris; }
    __syncthreads();
}
#else
template <bool COUNT>
D void trace_slice_bvh4(const PathState &P, const DevScene &S, int *lds_stack, unsigned int *ticket, const Work &work,
                        unsigned int b, unsigned int nb, unsigned int &traced, unsigned long long &n_nodes, unsigned long long &n_tris) {
    const unsigned int count = work.count;
    if (threadIdx.x == 0) *ticket = 0;
    __syncthreads();
    TStack st;
    int st_over[48 - PPG_TRACE_STACK];
    st.over = st_over;
    st.lds = lds_stack + threadIdx.x; st.stride = PPG_BLOCK; st.sp = 0; st.cap = PPG_TRACE_STACK;
    bool have = false;
    unsigned int i = 0;
    F3 o = f3s(0.0f), d = f3s(0.0f), id = f3s(0.0f);
    float mint = 0, maxt = 0;
    Hit best;
    best.t = 0; best.u = 0; best.v = 0; best.prim = -1;
    int bestOrig = 0, cur = 0;
    for (;;) {
        if (!have) {
            unsigned int k = atomicAdd(ticket, 1u);
            if (k >= count) break;
            i = work_item(work, k, b, nb);
            if (i >= P.n_paths) continue;
            float4 ro = P.ray_o[i], rd = P.ray_d[i];
            o = f3(ro.x, ro.y, ro.z); d = f3(rd.x, rd.y, rd.z);
            mint = ro.w; maxt = rd.w;
            if (mint == PPG_EPSILON)  // adaptive ray epsilon
                mint *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
            id = f3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
            best.t = __builtin_inff(); best.u = 0; best.v = 0; best.prim = -1;
            bestOrig = 0x7fffffff; cur = 0; st.sp = 0;
            have = true;
        }
        // One step per iteration and lane: either an interior node (test its four child boxes, continue with the nearest child, push the
        // others) or a leaf popped from the stack (test its triangles).  Leaves are NOT tested inside the node step: a wave executes the
        // union of its lanes' branches, and with the triangle loops nested in the node step almost every iteration ran up to four
        // divergent triangle loops for the few lanes that had a leaf child (measured on KITCHEN: the traversal was bound by issue slots,
        // not by bytes — 64-byte nodes alone changed nothing).
#if PPG_LEAF_VOTE > 0
        // leaves wait until PPG_LEAF_VOTE lanes of the wave have one (or no lane has an interior node left): the triangle code then runs
        // for many lanes at once instead of in almost every iteration for one or two of them
        const unsigned long long leafLanes = __ballot(cur < 0), nodeLanes = __ballot(cur >= 0 && cur != PPG_BVH4_EMPTY);
        const bool doLeaves = __popcll(leafLanes) >= PPG_LEAF_VOTE || nodeLanes == 0ull;
#else
        const bool doLeaves = true;
#endif
        if (cur >= 0) {
            if (COUNT) ++n_nodes;
            const Bvh4Hits hc = bvh4_children(S.bvh4 + cur, o, id, mint, fminf(maxt, best.t));
            if (hc.m > 0) {
                st.push_children(hc.m, hc.c1, hc.c2, hc.c3);
                cur = hc.c0;
            } else cur = st.sp > 0 ? st.pop() : PPG_BVH4_EMPTY;
        } else if (doLeaves) {
            const int code = ~cur;
            const int first = code >> 3, cnt = (code & 7) + 1;
            if (COUNT) n_tris += (unsigned long long)cnt;
            for (int q = first; q < first + cnt; ++q) {
                float tt, uu, vv;
                const float4 *T = S.accel + 3 * q;
                int orig;
                if (tri_hit(T, o, d, mint, fminf(maxt, best.t), tt, uu, vv, orig)) {
                    if (tt < best.t || (tt == best.t && orig < bestOrig)) { best.t = tt; best.u = uu; best.v = vv; best.prim = q; bestOrig = orig; }
                }
            }
            cur = st.sp > 0 ? st.pop() : PPG_BVH4_EMPTY;
        }
        if (cur == PPG_BVH4_EMPTY) {  // stack empty: this ray is done
            if (S.n_spheres) sphere_pass<false>(S, o, d, mint, maxt, best);
            P.hit[i] = make_float4(best.t, best.u, best.v, __int_as_float(best.prim));
            ++traced;
            have = false;
        }
    }
    __syncthreads();
}
#endif

// The counting sort of one workgroup's queue slice by the BSDF type at the new hit (see k_sort_slices below, which is this function as a
// kernel).  Since round 6 k_trace runs it for its own slice when the slice is traced: one launch per bounce fewer, and a workgroup that
// waits for its three dependent loads per ray (hit -> triangle -> material) does so while the CU's other workgroups still trace.
struct SortArgs {
    Field<float4> hit;
    const float4 *tris, *materials;
    unsigned int *out;        // this workgroup's slice of the sorted list
    unsigned char *kk;        // ... and of the key scratch
    unsigned int *count1, *n_common;  // this workgroup's entries
    BlockStats *stats;
    unsigned int n_paths;
    int n_tris;
};
D SortArgs sort_args(const PathState &P, const DevScene &S, const Queues &Q, unsigned int b, unsigned int *sorted, unsigned char *keys) {
    SortArgs a;
    a.hit = P.hit; a.tris = S.tris; a.materials = S.materials;
    a.out = sorted + (size_t)b * Q.cap; a.kk = keys + (size_t)b * Q.cap;
    a.count1 = Q.count[1] + b; a.n_common = Q.n_common ? Q.n_common + b : nullptr; a.stats = Q.stats + b;
    a.n_paths = P.n_paths; a.n_tris = S.n_tris;
    return a;
}
D void sort_slice(const SortArgs &a, const Work &work, unsigned int b, unsigned int nb, unsigned int *hist, unsigned int *offs) {
    if (threadIdx.x < 16) hist[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned int k = threadIdx.x; k < work.count; k += blockDim.x) {
        const unsigned int i = work_item(work, k, b, nb);
        unsigned int key = 255u;  // no path at this position (partial last chunk)
        if (i < a.n_paths) {
            // bins 0..7: the COMMON classes (MSET_COMMON), by BSDF type; 8..15: everything else — by type, then bump-mapped or masked
            // surfaces, spheres, and last the rays that left the scene
            const int prim = __float_as_int(a.hit[i].w);
            key = 15u;
            if (prim >= 0) {
                if (prim >= a.n_tris) key = 14u;
                else {
                    const int m = __float_as_int(a.tris[3 * (size_t)prim].w);
                    const float4 *mr = a.materials + PPG_MAT_STRIDE * (size_t)m;
                    const int type = (int)mr[0].w, flags = __float_as_int(mr[2].w);
                    const unsigned int tex = __float_as_uint(mr[5].x);
                    if ((flags & PPG_MAT_MASK) || (tex >> 16)) key = 13u;
                    else if (mset_common_type(type)) key = type == PPG_BSDF_ROUGHPLASTIC ? 6u : (unsigned int)type;  // 0..6
                    else key = 8u + ((unsigned int)type & 3u);  // dielectric 10, thin dielectric 11, rough dielectric 8
                }
            }
            atomicAdd(&hist[key], 1u);
        }
        a.kk[k] = (unsigned char)key;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int j = 0; j < 16; ++j) { offs[j] = acc; acc += hist[j]; if (j == 7 && a.n_common) { *a.n_common = acc; a.stats->shade_common += acc; } }
        *a.count1 = acc;
    }
    __syncthreads();
    for (unsigned int k = threadIdx.x; k < work.count; k += blockDim.x) {
        const unsigned int key = a.kk[k];
        if (key == 255u) continue;
        const unsigned int pos = atomicAdd(&offs[key], 1u);
        a.out[pos] = work_item(work, k, b, nb);
    }
}

// trace the rays of one queue slice
template <bool SMALL, bool COUNT>
D void trace_slice(const PathState &P, const DevScene &S, const LdsScene &L, int *lds_stack, unsigned int *ticket, const Work &work,
                   unsigned int b, unsigned int nb, unsigned int &traced, unsigned long long &n_nodes, unsigned long long &n_tris) {
    if (!SMALL) {
        trace_slice_bvh4<COUNT>(P, S, lds_stack, ticket, work, b, nb, traced, n_nodes, n_tris);
        return;
    }
    const unsigned int *items = work.items;
    const unsigned int count = work.count;
    const bool dense = work.mode == 2;
    for (unsigned int k = threadIdx.x; k < count; k += blockDim.x) {
        unsigned int i;
        if (items) i = items[dense ? dense_index(k, b, nb) : k];
        else i = first_path(k, b, nb);
        if (i >= P.n_paths) continue;
        float4 ro = P.ray_o[i], rd = P.ray_d[i];
        F3 o = f3(ro.x, ro.y, ro.z), d = f3(rd.x, rd.y, rd.z);
        Hit h = trace_small(L.tris, S, o, d, ro.w, rd.w);
        P.hit[i] = make_float4(h.t, h.u, h.v, __int_as_float(h.prim));
        ++traced;
    }
}

// SMALL: the whole scene (<= 64 triangles) is tested from LDS without a BVH.
// (with the pair compaction the kernel would take 72 VGPRs and seven waves a SIMD; held to 64 and eight it spills nothing and was measured
// 1.3 % (20 passes) / 1.7 % (127) faster on KITCHEN — profiles/r06_experiments.json)
#if !defined(PPG_TRACE_WAVES) && PPG_TRACE_PAIRS
#define PPG_TRACE_WAVES 8
#endif
template <bool SMALL, bool COUNT = false>
#ifdef PPG_TRACE_WAVES
__attribute__((amdgpu_waves_per_eu(PPG_TRACE_WAVES, PPG_TRACE_WAVES)))
#endif
__global__ __launch_bounds__(PPG_BLOCK) void k_trace(PathState P, DevScene S, Queues Q, int qin, int lds_nodes, int lds_tris, unsigned int *sorted, unsigned char *sort_keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ unsigned long long acc;
    __shared__ unsigned int ticket;
    __shared__ unsigned int sort_hist[16], sort_offs[16];
    __shared__ SortArgs sort_lds;
    const unsigned int b = blockIdx.x, nb = gridDim.x;
    if (Q.stop && *Q.stop) return;  // (uniform per workgroup)
    const Work work = work_of(P, Q, qin, nullptr, b, nb);
    if (work.count == 0) {
        if (sorted && threadIdx.x == 0) { Q.count[1][b] = 0; if (Q.n_common) Q.n_common[b] = 0; }
        return;
    }
    // what the sort at the end needs is put into LDS now and read back then: kept in scalar registers across the traversal loop (where the
    // compiler's merged kernel-argument loads leave it) it made the loop spill — k_trace 19.1 -> 20.8 ms with the sort switched off
    if (sorted && threadIdx.x == 0) sort_lds = sort_args(P, S, Q, b, sorted, sort_keys);
    // dynamic LDS: SMALL → the triangles; otherwise the traversal stacks [PPG_LDS_STACK][PPG_BLOCK]
    LdsScene L;
    if (SMALL) L = stage_scene(S, lds_raw, 0, lds_tris);
    else { L.nodes = nullptr; L.tris = nullptr; L.n_nodes = 0; L.n_tris = 0; }
    unsigned int traced = 0;
    unsigned long long n_nodes = 0, n_tris = 0;
    trace_slice<SMALL, COUNT>(P, S, L, (int *)lds_raw, &ticket, work, b, nb, traced, n_nodes, n_tris);
    block_add_u64(&acc, &Q.stats[b].rays, traced);
    if (COUNT) { block_add_u64(&acc, &Q.stats[b].bvh_nodes, n_nodes); block_add_u64(&acc, &Q.stats[b].bvh_tris, n_tris); }
    // the slice's hits are written (and visible to the workgroup: the barriers of trace_slice / block_add_u64): sort it for k_shade
    if (sorted) { const SortArgs sa = sort_lds; sort_slice(sa, work, b, nb, sort_hist, sort_offs); }
}

// grid[cell] for stree_lookup: descend at most PPG_GRID_LEVELS levels along the cell's coordinate bits
static __global__ void k_build_grid(const int4 *stree, unsigned int *grid) {
    unsigned int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= PPG_GRID_DIM * PPG_GRID_DIM * PPG_GRID_DIM) return;
    unsigned int coord[3] = {c % PPG_GRID_DIM, (c / PPG_GRID_DIM) % PPG_GRID_DIM, c / (PPG_GRID_DIM * PPG_GRID_DIM)};
    int used[3] = {0, 0, 0};
    int idx = 0, depth = 0;
    for (; depth < PPG_GRID_LEVELS; ++depth) {
        int4 n = stree[idx];
        if (n.y == 0) break;
        int a = n.x;
        unsigned int bit = (coord[a] >> (PPG_GRID_BITS - 1 - used[a])) & 1u;
        ++used[a];
        idx = bit ? n.z : n.y;
    }
    grid[c] = (unsigned int)idx | ((unsigned int)depth << 27);
}

// ------------------------------------------------------------------------------------------------
// Splatting: DTree::recordIrradiance (GP:395-413) into the building tree of one S-tree leaf
// ------------------------------------------------------------------------------------------------
// (the `statisticalWeight += w` half of recordIrradiance is done by the caller, wave-combined)
#ifndef PPG_BOX_STACK
#define PPG_BOX_STACK 8
#endif
// The building tree a record is splatted into: the pool in HBM (TreeGlobal: accumulation by global atomics) or one D-tree staged in LDS by
// k_splat_sorted (TreeLds: topology and accumulators private to the workgroup, accumulation by LDS atomics, flushed once).  The sums are
// integers: the order and the place of the additions do not show in the result.
struct TreeGlobal {
    const ushort4 *child;      // first node of this D-tree in the pool
    unsigned long long *acc;   // its accumulators [node * 4 + slot]
    D ushort4 node(unsigned int n) const { return child[n]; }
    D void add(unsigned int n, int slot, unsigned long long v) const { atomicAdd(&acc[(size_t)n * 4 + slot], v); }
};
template <typename TR>
D void dtree_record_t(const TR &tree, float px, float py, float irradiance, float w, int dfilter, unsigned long long *box_stack = nullptr, int box_stride = 0) {
    if (!(ppg_isfinite(w) && w > 0)) return;
    if (!(ppg_isfinite(irradiance) && irradiance > 0)) return;
    if (dfilter == DF_NEAREST) {  // QuadTreeNode::record, GP:303-312
        unsigned int node = 0;
        for (;;) {
            int index = quad_child_index(px, py);
            const ushort4 ch = tree.node(node);
            const unsigned int c = index == 0 ? ch.x : (index == 1 ? ch.y : (index == 2 ? ch.z : ch.w));
            if (c == 0) {
                tree.add(node, index, ppg_to_fixed(irradiance * w));
                break;
            }
            node = c;
        }
    } else {
        // depthAt (GP:247-255), then the box splat (GP:403-409, 322-338) with an explicit stack
        int depth = 0;
        {
            float qx = px, qy = py;
            unsigned int node = 0;
            for (;;) {
                int index = quad_child_index(qx, qy);
                ++depth;
                const ushort4 ch = tree.node(node);
                const unsigned int c = index == 0 ? ch.x : (index == 1 ? ch.y : (index == 2 ? ch.z : ch.w));
                if (c == 0) break;
                node = c;
            }
        }
        const float size = ppg_exp2i(-depth);
        const float ox = px - size / 2, oy = py - size / 2;
        const float value = irradiance * w / (size * size);
        // Explicit stack of (node, cell) pairs.  A cell of level l is [ix, ix + 1) x [iy, iy + 1) * 2^-l — dyadic, so its float corner
        // ix * 2^-l is exactly what the reference's running `origin + childSize` additions produce — and an entry packs into 64 bits:
        // node (16) | level (5) | ix (21) | iy (21).  With `box_stack` (k_commit, k_splat_sorted: an LDS column per lane) the first
        // PPG_BOX_STACK entries never leave the CU; a 64-entry array in scratch memory — 1 KB per lane, far beyond L1 — was the whole stack before.
        unsigned long long over[64];
        int sp = 0;
        auto push = [&](unsigned long long v) {
            if (box_stack && sp < PPG_BOX_STACK) box_stack[sp * box_stride] = v; else over[box_stack ? sp - PPG_BOX_STACK : sp] = v;
            ++sp;
        };
        auto pop = [&]() -> unsigned long long {
            --sp;
            return (box_stack && sp < PPG_BOX_STACK) ? box_stack[sp * box_stride] : over[box_stack ? sp - PPG_BOX_STACK : sp];
        };
        push(0ull);
        while (sp) {
            const unsigned long long e = pop();
            const unsigned int enode = (unsigned int)(e & 0xffffu), level = (unsigned int)(e >> 16) & 31u;
            const unsigned int ix = (unsigned int)(e >> 21) & 0x1fffffu, iy = (unsigned int)(e >> 42) & 0x1fffffu;
            const float es = ppg_exp2i(-(int)level);
            const float ex = (float)ix * es, ey = (float)iy * es;
            float childSize = es / 2;
            ushort4 ch = tree.node(enode);
            const unsigned short cc[4] = {ch.x, ch.y, ch.z, ch.w};
            // the reference recurses depth-first in child order; contributions to distinct leaf slots commute (integer adds)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float cx = ex, cy = ey;
                if (i & 1) cx += childSize;
                if (i & 2) cy += childSize;
                float lx = ppg_max(ppg_min(ox + size, cx + childSize) - ppg_max(ox, cx), 0.0f);
                float ly = ppg_max(ppg_min(oy + size, cy + childSize) - ppg_max(oy, cy), 0.0f);
                float ww = lx * ly;
                if (ww > 0.0f) {
                    if (cc[i] == 0) tree.add(enode, i, ppg_to_fixed(value * ww));
                    else push((unsigned long long)cc[i] | ((unsigned long long)(level + 1) << 16) | ((unsigned long long)(2 * ix + (i & 1)) << 21) |
                              ((unsigned long long)(2 * iy + ((i >> 1) & 1)) << 42));
                }
            }
        }
    }
}
D void dtree_record(const DevTree &T, int leaf, float px, float py, float irradiance, float w, int dfilter, unsigned long long *box_stack = nullptr,
                    int box_stride = 0) {
    if (!(ppg_isfinite(w) && w > 0)) return;
    if (!(ppg_isfinite(irradiance) && irradiance > 0)) return;
    const unsigned int base = T.hdr[leaf].b_base;
    const TreeGlobal tree{T.bchild + base, T.bacc + (size_t)base * 4};
    dtree_record_t(tree, px, py, irradiance, w, dfilter, box_stack, box_stride);
}

struct Rec {  // DTreeRecord, GP:562-568
    F3 d;
    float radiance, product, woPdf, bsdfPdf, dTreePdf, statisticalWeight;
    bool isDelta;
    // place of this record's optimizeBsdfSamplingFraction call in the round's canonical order (include/ppg.h): path id, record code,
    // and its position in the record buffer (0xffffffff: append)
    unsigned int adamPath, adamCode, adamPos;
};

// path id of the Adam records: (sample index within the round) * (pixels of the whole image) + pixel index
// (a straggler — P.orig — committed one round late carries the "deferred" bit above the path bits: include/ppg.h "Stragglers")
D unsigned int adam_path_id(const PathState &P, const RenderParams &R, unsigned int i) {
    if (P.orig) {
        const unsigned int o = P.orig[i];
        return ((o / P.n_pix) * R.img_pixels + P.pixels[o % P.n_pix]) | PPG_ADAM_DEFER_PATH_BIT;
    }
    return (i / P.n_pix) * R.img_pixels + P.pixels[i % P.n_pix];
}

// DTreeWrapper::record (GP:575-584); of optimizeBsdfSamplingFraction (GP:672-697)
// the optimiser's record is only WRITTEN here (position known in advance, or appended); it is applied at the end of the round by k_adam_apply.
// COMBINE: called by all lanes of a wave (inactive lanes pass active = false); the per-D-tree counters
// are then pre-combined across the wave.
template <bool COMBINE>
D void wrapper_record(const DevTree &T, int leaf, const Rec &rec, int dfilter, int loss, bool active, unsigned long long *box_stack = nullptr,
                      int box_stride = 0) {
    const bool irr = active && !rec.isDelta;
    float irradiance = 0, px = 0, py = 0;
    if (irr) {
        irradiance = rec.radiance / rec.woPdf;
        dir_to_canonical(rec.d, px, py);
    }
    const bool wOk = irr && ppg_isfinite(rec.statisticalWeight) && rec.statisticalWeight > 0;
    const unsigned long long wf = wOk ? ppg_to_fixed(rec.statisticalWeight) : 0ull;
    if (COMBINE) wave_key_add<3>(T.bweight_rep, (unsigned int)leaf * PPG_REPLICAS + (blockIdx.x & (PPG_REPLICAS - 1)), wf, wOk);
    else if (wOk) atomicAdd(&T.bweight_rep[(size_t)leaf * PPG_REPLICAS + (blockIdx.x & (PPG_REPLICAS - 1))], wf);
    if (wOk) dtree_record(T, leaf, px, py, irradiance, rec.statisticalWeight, dfilter, box_stack, box_stride);

    // if (bsdfSamplingFractionLoss != ENone && rec.product > 0) optimizeBsdfSamplingFraction(rec, ...), GP:581-583: deferred to the end
    // of the round (k_adam_apply), here only written down
    if (active && loss != LOSS_NONE && rec.product > 0) {
        unsigned int pos = rec.adamPos;
        if (pos == 0xffffffffu) {
            pos = atomicAdd(T.adam_count, 1u);
            if (pos >= T.adam_cap) { T.adam_count[1] = 1u; return; }  // reported as an error by the host
        }
        AdamRec a;
        a.key = ((unsigned long long)(unsigned int)leaf << PPG_ADAM_LEAF_SHIFT) | ((unsigned long long)rec.adamPath << PPG_ADAM_CODE_BITS) | rec.adamCode;
        a.product = rec.product; a.woPdf = rec.woPdf; a.bsdfPdf = rec.bsdfPdf; a.dTreePdf = rec.dTreePdf; a.weight = rec.statisticalWeight; a.pad = 0.0f;
        T.adam_keys[pos] = a.key;
        T.adam_recs[pos] = a;
    }
}

// STree::record + STreeNode::record (GP:935-943, 823-839): box spatial filter
D void stree_record_box(const DevTree &T, F3 p, F3 vox, Rec rec, int dfilter, int loss) {
    float volume = 1;
    volume *= vox.x; volume *= vox.y; volume *= vox.z;
    rec.statisticalWeight /= volume;
    F3 min1 = p - vox * 0.5f, max1 = p + vox * 0.5f;
    struct E { int node; float mx, my, mz, sx, sy, sz; };
    E st[96];
    int sp = 0;
    st[sp++] = E{0, T.aabb_min[0], T.aabb_min[1], T.aabb_min[2], T.aabb_ext[0], T.aabb_ext[1], T.aabb_ext[2]};
    while (sp) {
        E e = st[--sp];
        float lx = ppg_max(ppg_min(max1.x, e.mx + e.sx) - ppg_max(min1.x, e.mx), 0.0f);
        float ly = ppg_max(ppg_min(max1.y, e.my + e.sy) - ppg_max(min1.y, e.my), 0.0f);
        float lz = ppg_max(ppg_min(max1.z, e.mz + e.sz) - ppg_max(min1.z, e.mz), 0.0f);
        float w = lx * ly * lz;
        if (!(w > 0)) continue;
        int4 n = T.stree[e.node];
        if (n.y == 0) {
            Rec r2 = rec;
            r2.statisticalWeight = rec.statisticalWeight * w;
            wrapper_record<false>(T, e.node, r2, dfilter, loss, true);
        } else {
            float m2[3] = {e.mx, e.my, e.mz}, s2[3] = {e.sx, e.sy, e.sz};
            s2[n.x] /= 2;
            E c0 = E{n.y, m2[0], m2[1], m2[2], s2[0], s2[1], s2[2]};
            m2[n.x] += s2[n.x];
            E c1 = E{n.z, m2[0], m2[1], m2[2], s2[0], s2[1], s2[2]};
            if (sp + 2 <= 96) { st[sp++] = c1; st[sp++] = c0; }
        }
    }
}

// Vertex::commit up to the spatial-filter dispatch (GP:1730-1744): false = the vertex is dropped
D bool vertex_to_rec(F3 radiance, F3 bsdfVal, F3 throughput, float woPdf, float bsdfPdf, float dTreePdf, F3 d, bool isDelta,
                     float statisticalWeight, Rec &rec) {
    if (!(woPdf > 0) || !isvalid3(radiance) || !isvalid3(bsdfVal)) return false;
    F3 localRadiance = f3s(0.0f);
    if (throughput.x * woPdf > PPG_EPSILON) localRadiance.x = radiance.x / throughput.x;
    if (throughput.y * woPdf > PPG_EPSILON) localRadiance.y = radiance.y / throughput.y;
    if (throughput.z * woPdf > PPG_EPSILON) localRadiance.z = radiance.z / throughput.z;
    F3 product = mul3(localRadiance, bsdfVal);
    rec.d = d;
    rec.radiance = avg3(localRadiance); rec.product = avg3(product);
    rec.woPdf = woPdf; rec.bsdfPdf = bsdfPdf; rec.dTreePdf = dTreePdf;
    rec.statisticalWeight = statisticalWeight;
    rec.isDelta = isDelta;
    return true;
}

// the stochastic filter's jittered lookup (GP:1746-1763)
D int stochastic_leaf(const DevTree &T, F3 o, F3 vox, unsigned int key, unsigned int dim) {
    F3 offset = vox;
    offset.x *= ppg_rand(key, dim++) - 0.5f;
    offset.y *= ppg_rand(key, dim++) - 0.5f;
    offset.z *= ppg_rand(key, dim++) - 0.5f;
    F3 og = o + offset;
    og.x = ppg_min(ppg_max(og.x, T.aabb_min[0]), T.aabb_max[0]);  // AABB::clip
    og.y = ppg_min(ppg_max(og.y, T.aabb_min[1]), T.aabb_max[1]);
    og.z = ppg_min(ppg_max(og.z, T.aabb_min[2]), T.aabb_max[2]);
    F3 dummy;
    return stree_lookup(T, og, dummy);
}

// Vertex::commit's filter dispatch for ONE lane (the direct-light vertex of GP:1994-2010, committed inside Li's loop)
D void commit_single(const DevTree &T, int sfilter, int dfilter, int loss, int leaf, F3 o, F3 vox, const Rec &rec, unsigned int key,
                     unsigned int dim) {
    if (sfilter == SF_BOX) {
        stree_record_box(T, o, vox, rec, dfilter, loss);
    } else {
        if (sfilter == SF_STOCHASTIC) leaf = stochastic_leaf(T, o, vox, key, dim);
        wrapper_record<false>(T, leaf, rec, dfilter, loss, true);
    }
}

// ------------------------------------------------------------------------------------------------
// k_shade — Li's loop body (GP:1798-2146), surface branch
// ------------------------------------------------------------------------------------------------
// Shadow-ray test of Scene::evalTransmittance (scene.cpp:619-679) without media / null BSDFs: true = occluded.
// small_tris != nullptr: the whole scene is staged in LDS (brute force); otherwise BVH4 any-hit with this lane's
// LDS stack column.
// SPH: the scene may hold analytic spheres (FULL kernels; such scenes never take the LDS brute-force path).
template <bool SPH = false>
D bool shadow_occluded(const DevScene &S, const float4 *small_tris, int *stack_col, F3 o, F3 d, float maxt) {
    if (small_tris) return trace_small(small_tris, S, o, d, PPG_EPSILON, maxt).prim >= 0;
    float rayMinT = PPG_EPSILON;  // adaptive ray epsilon, skdtree.cpp:125-129
    rayMinT *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
    return trace_closest4<true, SPH>(S, stack_col, PPG_BLOCK, o, d, rayMinT, maxt).prim >= 0;
}

// closest hit for rays traced inside k_shade (look-through / transmittance loops): LDS brute force or BVH4 with the lane's stack column
D Hit trace_inline(const DevScene &S, const float4 *small_tris, int *stack_col, F3 o, F3 d, float maxt) {  // FULL kernels only
    if (small_tris) return trace_small(small_tris, S, o, d, PPG_EPSILON, maxt);
    float rayMinT = PPG_EPSILON;  // adaptive ray epsilon, skdtree.cpp:125-129
    rayMinT *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
    return trace_closest4<false, true>(S, stack_col, PPG_BLOCK, o, d, rayMinT, maxt);
}
// intersection record of a hit found by trace_inline / k_trace in a FULL kernel: triangle or sphere (`o` = that ray's origin)
D void fill_isect_full(const DevScene &S, const Hit &h, F3 o, F3 d, Isect &I) {
    if (h.prim >= S.n_tris) fill_isect_sphere(S, h, o, d, I);
    else fill_isect(S, h, d, I);
}

// Scene::evalTransmittance (scene.cpp:619-679), surfaces only, for scenes with null-component BSDFs: zero behind an occluder,
// otherwise the product of the null components (evaluated in the geometric frame) of the surfaces passed.
D F3 shadow_transmittance(const DevScene &S, const float4 *small_tris, int *stack_col, F3 p1, F3 d, float remaining, float lengthFactor,
                          int maxInteractions, unsigned int &traced) {
    F3 o = p1;
    float maxt = remaining * lengthFactor;
    F3 transmittance = f3s(1.0f);
    int interactions = 0;
    while (remaining > 0) {
        ++traced;
        const Hit h = trace_inline(S, small_tris, stack_col, o, d, maxt);
        const bool surface = h.prim >= 0;
        Mat M;
        Isect I;
        if (surface) { fill_isect_full(S, h, o, d, I); M = load_material(S, I.material); }
        if (surface && (interactions == maxInteractions || !mat_has_null(M))) return f3s(0.0f);
        if (!surface || iszero3(transmittance)) break;
        const float cosThetaI = dot3(I.geoN, -d);
        transmittance = mul3(transmittance, mat_eval_null(M, cosThetaI));
        if (++interactions > 100) break;
        o = o + d * h.t;
        remaining -= h.t;
        maxt = remaining * lengthFactor;
    }
    return transmittance;
}

// LDS the next-event-estimation variants add to k_shade / k_tail
struct NeeLds {
    const float4 *small_tris;  // staged triangles (small scenes) or nullptr
    int *stack_col;            // BVH4 stack column of this lane
};

// One queue slice through Li's loop body.  FUSED (small scenes): the ray sampled here is traced here too.
// NEE: the variant with luminaire sampling (GP:1962-2021) and MIS against it (GP:2083-2088); the shadow ray is
// traced and the direct-light vertex committed in place, as in the reference's loop.
// FULL: the complete material set (ppg_device.h "Full material set"); otherwise only diffuse / two-sided diffuse / mirror.
// Li's loop body for ONE path: returns whether the path goes on (a new ray was written).  Free of cross-lane operations, so
// it may be called under divergence (k_tail).  plen receives rRec.depth when the path ends here.
// CARRY (k_tail): the path's state travels in the lane's registers from bounce to bounce (`cs`) instead of through memory — in the tail a
// bounce of a lone path is a chain of dependent latencies, and the store -> load round trips of ray, hit and state through L2 were two
// of its links.  Memory receives the words others read afterwards (misc, li) when the path ends.
struct Carried {
    uint4 misc; float4 thr, li, hit, ro, rd;
#ifdef PPG_PROBE
    unsigned long long *plds, pt, prt; bool pon;
#endif
};
template <bool FUSED, bool NEE, bool FULL, bool CARRY = false, int MSET = MSET_ALL>
D bool shade_one(const PathState &P, const DevScene &S, const DevTree &T, const RenderParams &R, const unsigned int i, const LdsColumn &fcol,
                 const float4 *lds_tris, unsigned long long &plen, unsigned int &traced, const NeeLds &nee, unsigned long long &committed,
                 Carried *cs = nullptr) {
    bool alive = false;
    {
    uint4 m = CARRY ? cs->misc : P.misc[i];
    unsigned int key = m.x, dim = m.y, flags = m.z;
    unsigned int depth = flags & FL_DEPTH_MASK;
    unsigned int nV = (flags & FL_NV_MASK) >> FL_NV_SHIFT;
    float4 t4, l4, h4, d4;
    if (CARRY) { t4 = cs->thr; l4 = cs->li; h4 = cs->hit; d4 = cs->rd; }
    else { t4 = P.thr[i]; l4 = P.li[i]; h4 = P.hit[i]; d4 = P.ray_d[i]; }
    auto ray_origin = [&]() { return CARRY ? cs->ro : P.ray_o[i]; };
    F3 thr = f3(t4.x, t4.y, t4.z);
    float eta = t4.w;
    F3 Li = f3(l4.x, l4.y, l4.z);
    F3 d = f3(d4.x, d4.y, d4.z);
    Hit h;
    h.t = h4.x; h.u = h4.y; h.v = h4.z; h.prim = __float_as_int(h4.w);
    if (MSET == MSET_COMMON) __builtin_assume(h.prim >= 0);
    const bool valid = MSET == MSET_COMMON ? true : h.prim >= 0;
    Isect I;
    TexInfo X;
    X.tex = 0u;
    if (valid) {
        if (FULL && MSET != MSET_COMMON && h.prim >= S.n_tris) {
            const float4 ro4 = ray_origin();
            fill_isect_sphere(S, h, f3(ro4.x, ro4.y, ro4.z), d, I);
        } else if (FULL) fill_isect_tex(S, h, d, I, X);
        else fill_isect(S, h, d, I);
    }
    PROBE_MARK(cs, 2);
    bool go = true;

    if (FULL && (flags & FL_PENDING) && (flags & FL_PEND_NULL)) {
        // ---- the previous bounce passed straight through a null component, GP:2045-2075: no emitter lookup, no MIS,
        // no Russian roulette; rRec.type = scattered ? ERadianceNoEmission : ERadiance; rRec.depth++; continue ----
        if (flags & FL_SCATTERED) flags &= ~FL_EMITTED_OK; else flags |= FL_EMITTED_OK;
        ++depth;
        if (!((int)depth <= R.max_depth || R.max_depth < 0)) go = false;
        flags &= ~(FL_PENDING | FL_PEND_TREE | FL_PEND_DELTA | FL_PEND_REFN | FL_PEND_NULL);
    } else if (flags & FL_PENDING) {
        // ---- second half of the previous bounce: GP:2078-2145 ----
        F3 value = valid ? eval_Le(S, I, -d) : f3s(0.0f);  // rayIntersectAndLookForEmitter, GP:2229-2234
        // dRec.setQuery(ray, its) of the emitter that was found (records.inl:170-178)
        F3 em_n = I.n;
        float em_dist = h.t;
        int em_id = I.emitter;
        if (FULL && MSET != MSET_COMMON && !valid && S.env.w != 0) {  // GP:2236-2243: the ray left the scene
            const float4 ro4 = ray_origin();
            if (env_fill_direct(S, f3(ro4.x, ro4.y, ro4.z), d)) { value = env_radiance(S, d); em_id = S.n_emitters; }
        }
        if (FULL && MSET != MSET_COMMON && S.has_null && valid && I.emitter < 0) {
            // rayIntersectAndLookForEmitter GP:2184-2245: the path continues from THIS hit, but the search for an emitter
            // goes on through surfaces that have a null component (traced in place)
            Mat Mc = load_material(S, I.material);
            if (mat_has_null(Mc)) {
                const float4 ro4 = ray_origin();
                F3 ro = f3(ro4.x, ro4.y, ro4.z);
                F3 transmittance = f3s(1.0f);
                const int maxInteractions = R.max_depth - (int)depth - 1;
                int interactions = 0;
                bool abandoned = false, surface = true;
                Hit hc = h;
                Isect Ic = I;
                for (;;) {
                    if (interactions == maxInteractions || !mat_has_null(Mc) || Ic.emitter >= 0) break;
                    if (iszero3(transmittance)) { abandoned = true; break; }
                    const float cosThetaI = -to_local(Ic, d).z;  // bRec(its, -wo, wo) in the shading frame
                    transmittance = mul3(transmittance, mat_eval_null(Mc, cosThetaI));
                    ro = ro + d * hc.t;
                    if (++interactions > 100) { abandoned = true; break; }
                    hc = trace_inline(S, nee.small_tris, nee.stack_col, ro, d, __builtin_inff());
                    ++traced;
                    if (hc.prim < 0) { surface = false; break; }
                    fill_isect_full(S, hc, ro, d, Ic);
                    Mc = load_material(S, Ic.material);
                }
                if (!abandoned && surface && Ic.emitter >= 0) {
                    value = mul3(transmittance, eval_Le(S, Ic, -d));
                    em_n = Ic.n; em_dist = hc.t; em_id = Ic.emitter;  // dist from the LAST ray origin, as in the reference
                } else if (!abandoned && !surface && S.env.w != 0 && env_fill_direct(S, ro, d)) {
                    value = mul3(transmittance, env_radiance(S, d));
                    em_id = S.n_emitters;
                }
            }
        }
        const float woPdf = l4.w;
        const bool isDelta = (flags & FL_PEND_DELTA) != 0;
        const bool hasTree = (flags & FL_PEND_TREE) != 0;
        float emitterPdf = 0.0f;  // GP:2085: scene->pdfEmitterDirect(dRec) (scene.cpp:949-952, area.cpp:175-183, shape.cpp:117-126)
        if (NEE && R.do_nee && !isDelta && !iszero3(value)) {
            float pdfDirect = 0.0f;
            const float dn = dot3(d, em_n);
            if (FULL && em_id == S.n_emitters) {
                pdfDirect = S.env.w == 2.0f ? envmap_pdf_direction(S, envmap_to_local(S, d)) : env_pdf_direct(P.nee_cos[i]);
            } else if ((flags & FL_PEND_REFN) && dn < 0) {
                const int4 info = S.em_info[em_id];
                if (FULL && info.y < 0) {  // Sphere::pdfDirect needs dRec.ref = the previous vertex = this ray's origin
                    const float4 ro4 = ray_origin();
                    pdfDirect = sphere_pdf_direct(S.spheres + 4 * (-info.y - 1), f3(ro4.x, ro4.y, ro4.z), d, em_n, em_dist);
                } else pdfDirect = __int_as_float(info.w) * (em_dist * em_dist) / ppg_abs(dn);
            }
            emitterPdf = pdfDirect * (1.0f * S.em_sel_norm);
        }
        float pa = woPdf * woPdf, pb = emitterPdf * emitterPdf;  // miWeight(woPdf, emitterPdf), GP:2247-2250
        const float weight = pa / (pa + pb);
        F3 L = mul3(thr, value) * weight;
        if (!iszero3(L)) {  // recordRadiance, GP:1791-1796
            Li = Li + L;
            for (unsigned int v0 = 0; v0 < nV; v0 += 4) {  // 4 independent loads in flight, then 4 stores
                float4 rr[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v0 + k < nV) rr[k] = P.v_rad[(size_t)(v0 + k) * P.n_paths + i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v0 + k < nV) {
                        rr[k].x += L.x; rr[k].y += L.y; rr[k].z += L.z;
                        P.v_rad[(size_t)(v0 + k) * P.n_paths + i] = rr[k];
                    }
            }
        }
        if ((!isDelta || R.loss != LOSS_NONE) && hasTree && nV < PPG_MAX_VERTICES && nV < (unsigned int)R.max_vertices &&
            !R.is_final_iter) {
            if (1 / woPdf > 0) {  // the other vertex fields were written when the bounce was sampled
                F3 rad = (R.nee == NEE_ALWAYS) ? f3s(0.0f) : L;
                unsigned int bits = m.w | (isDelta ? 0x80000000u : 0u);
                P.v_rad[(size_t)nV * P.n_paths + i] = make_float4(rad.x, rad.y, rad.z, __uint_as_float(bits));
                ++nV;
            }
        }
        flags &= ~FL_EMITTED_OK;  // rRec.type = ERadianceNoEmission
        if (depth++ >= (unsigned int)R.rr_depth) {  // Russian roulette, GP:2124-2142
            float successProb = 1.0f;
            if (hasTree && !isDelta) {
                if (!T.is_built) successProb = max3(thr) * eta * eta;
                successProb = ppg_max(0.1f, ppg_min(successProb, 0.99f));
            }
            if (ppg_rand(key, dim++) >= successProb) go = false;
            else thr = div3(thr, successProb);
        }
        if (go) {
            flags |= FL_SCATTERED;
            if (!((int)depth <= R.max_depth || R.max_depth < 0)) go = false;
        }
        flags &= ~(FL_PENDING | FL_PEND_TREE | FL_PEND_DELTA | FL_PEND_REFN);
    }

    PROBE_MARK(cs, 3);
    // ---- first half of this bounce: GP:1902-2040 ----
    if (MSET != MSET_COMMON && go && !valid) {  // GP:1902-1914: possibly radiance from a background luminaire, then the path ends
        if (FULL && S.env.w != 0 && (flags & FL_EMITTED_OK) && (!R.hide_emitters || (flags & FL_SCATTERED)))
            Li = Li + mul3(thr, env_radiance(S, d));  // (nVertices == 0 whenever emission is still enabled)
        go = false;
    }
    if (go) {
        if (I.emitter >= 0 && (flags & FL_EMITTED_OK) && (!R.hide_emitters || (flags & FL_SCATTERED)))
            Li = Li + mul3(thr, eval_Le(S, I, -d));  // GP:1917-1919 (nVertices == 0 here)
        if ((int)depth >= R.max_depth && R.max_depth != -1) go = false;
    }
    if (go) {
        float wiDotGeoN = -dot3(I.geoN, d), wiDotShN = I.wi.z;
        if (wiDotGeoN * wiDotShN < 0 && R.strict_normals) go = false;
    }
    if (go) {
        Mat M;
        if (FULL) {
            M = load_material(S, I.material);
            if (MSET == MSET_COMMON) {  // what k_sort_slices put into the common part of the slice
                if (!(M.type == PPG_BSDF_DIFFUSE || M.type == PPG_BSDF_MIRROR || M.type == PPG_BSDF_CONDUCTOR || M.type == PPG_BSDF_ROUGHCONDUCTOR ||
                      M.type == PPG_BSDF_PLASTIC || M.type == PPG_BSDF_ROUGHPLASTIC)) __builtin_unreachable();
                if (M.flags & PPG_MAT_MASK) __builtin_unreachable();
            }
        } else {
            const float4 mat = S.materials[PPG_MAT_STRIDE * (size_t)I.material];
            M.type = (int)mat.w; M.flags = 0; M.refl = f3(mat.x, mat.y, mat.z);
        }
        const bool smooth = FULL ? mat_is_smooth(M) : bsdf_is_smooth(M.type);  // bsdf->getType() & ESmooth: only those are guided
        // bitmap on the diffuse reflectance (m_reflectance->eval(bRec.its)); `bumpmap` adapter: the frame perturbed by the displacement
        // texture's gradient, in which the nested BSDF is queried (bumpmap.cpp:162-219)
        bool bumped = false;
        F3 ps = f3s(0.0f), pt = f3s(0.0f), pn = f3s(0.0f);
        if (FULL && X.tex) {
            if (X.tex & 0xffffu) M.refl = tex_eval(S.textures[(X.tex & 0xffffu) - 1u], X.u, X.v);
            if (MSET != MSET_COMMON && (X.tex >> 16)) { bump_frame(S, I, X, ps, pt, pn); bumped = true; }
        }
        // (one call site per BSDF function: the bumped variant only transforms the arguments first — two inlined copies of the whole
        // material switch per function made the FULL kernels 330 KB of code, five times the instruction cache)
        auto to_pert = [&](F3 v_) { const F3 w_ = to_world(I, v_); return f3(dot3(w_, ps), dot3(w_, pt), dot3(w_, pn)); };
        auto b_eval = [&](F3 wi_, F3 wo_) {
            if (!FULL) return bsdf_eval(M.type, M.refl, wi_, wo_);
            F3 wia = wi_, woa = wo_;
            if (bumped) {
                wia = to_pert(wi_); woa = to_pert(wo_);
                if (wo_.z * woa.z <= 0) return f3s(0.0f);
            }
            return mat_eval(M, wia, woa);
        };
        auto b_pdf = [&](F3 wi_, F3 wo_) {
            if (!FULL) return bsdf_pdf(M.type, wi_, wo_);
            F3 wia = wi_, woa = wo_;
            if (bumped) {
                wia = to_pert(wi_); woa = to_pert(wo_);
                if (wo_.z * woa.z <= 0) return 0.0f;
            }
            return mat_pdf(M, wia, woa);
        };
        float sampledEta = 1.0f;
        bool sampledNull = false;
        auto b_sample = [&](float u_, float v_, F3 &wo_, float &pdf_, bool &delta_) {
            if (!FULL) return bsdf_sample(M.type, M.refl, I.wi, u_, v_, wo_, pdf_, delta_);
            F3 wop = f3s(0.0f);
            F3 result = mat_sample(M, bumped ? to_pert(I.wi) : I.wi, u_, v_, wop, pdf_, delta_, sampledEta, sampledNull, key, dim);
            if (bumped) {
                if (!iszero3(result)) {
                    wo_ = to_local(I, ps * wop.x + pt * wop.y + pn * wop.z);
                    if (wo_.z * wop.z <= 0) result = f3s(0.0f);
                }
            } else wo_ = wop;
            return result;
        };
        PROBE_MARK(cs, 4);
        F3 vox = f3s(0.0f);
        int leaf = 0;
        DTreeRef hd;
        hd.s_base = 0; hd.s_sum = 0; hd.s_statw = 0;
        float frac = R.bsdf_sampling_fraction;  // GP:1946-1949
        if (smooth) {
            leaf = stree_lookup(T, I.p, vox);  // GP:1942-1944
            const float4 h4 = *reinterpret_cast<const float4 *>(&T.hdr[leaf]);  // {s_base, s_num, s_sum, s_statw}
            hd.s_base = __float_as_uint(h4.x); hd.s_sum = h4.z; hd.s_statw = h4.w;
            if (R.loss != LOSS_NONE) frac = logistic(T.theta_frozen ? T.theta_frozen[leaf] : T.hdr[leaf].theta);
        }

        PROBE_MARK(cs, 5);
        // sampleMat, GP:1650-1691
        float sx = ppg_rand(key, dim++);
        float sy = ppg_rand(key, dim++);
        F3 wo_l, bsdfWeight;
        float woPdf, bsdfPdf, dTreePdf;
        bool sampledDelta = false;
        // ONE call site for the BSDF's sample(): called from two places the lambda is not inlined, and everything it captures by
        // reference (intersection record, material, texture info) then lives in scratch memory
        const bool unguided = !T.is_built || !smooth;  // !m_isBuilt || !dTree || all components are delta
        const bool viaBsdf = unguided || sx < frac;
        F3 sampled = f3s(0.0f);
        if (viaBsdf) {
            if (!unguided) sx /= frac;
            sampled = b_sample(sx, sy, wo_l, bsdfPdf, sampledDelta);
        }
        if (unguided) {
            bsdfWeight = sampled;
            woPdf = bsdfPdf;
            dTreePdf = 0;
        } else {
            F3 result;
            bool zero = false, deltaEarly = false;
            if (viaBsdf) {
                result = sampled;
                if (iszero3(result)) zero = true;
                else if (FULL && sampledDelta) deltaEarly = true;  // GP:1672-1676: a delta lobe of a mixed BSDF
                else result = result * bsdfPdf;
            } else {
                // sample.x is remapped but unused on this branch (GP:1680-1682)
                float cx, cy;
                dtree_sample(T, hd, key, dim, cx, cy);
                wo_l = to_local(I, canonical_to_dir(cx, cy));
                sampledEta = 1.0f; sampledNull = false;
                result = b_eval(I.wi, wo_l);
            }
            if (zero) {
                woPdf = bsdfPdf = dTreePdf = 0;
                bsdfWeight = f3s(0.0f);
                wo_l = f3s(0.0f);
            } else if (deltaEarly) {
                dTreePdf = 0;
                woPdf = bsdfPdf * frac;
                bsdfWeight = div3(result, frac);
            } else {
                // pdfMat, GP:1693-1710
                PROBE_MARK(cs, 6);
                dTreePdf = 0;
                bsdfPdf = b_pdf(I.wi, wo_l);
                PROBE_MARK(cs, 7);
                if (!ppg_isfinite(bsdfPdf)) {
                    woPdf = 0;
                } else {
                    float cx, cy;
                    dir_to_canonical(to_world(I, wo_l), cx, cy);
                    dTreePdf = dtree_pdf(T, hd, cx, cy, fcol);
                    woPdf = frac * bsdfPdf + (1 - frac) * dTreePdf;
                }
                PROBE_MARK(cs, 8);
                bsdfWeight = (woPdf == 0) ? f3s(0.0f) : div3(result, woPdf);
            }
        }
        // Luminaire sampling, GP:1962-2021
        const bool noRefN = FULL ? mat_backside_or_transmission(M) : (M.type == PPG_BSDF_TWOSIDED_DIFFUSE);
        const F3 refN = noRefN ? f3s(0.0f) : I.n;  // DirectSamplingRecord(its), records.inl:160-164
        if (NEE && R.do_nee && smooth) {
            const float ex = ppg_rand(key, dim++);
            const float ey = ppg_rand(key, dim++);
            DirectSample ds;
            F3 value = emitter_sample_direct(S, I.p, refN, ex, ey, ds);
            if (ds.pdf != 0) {
                if (FULL && S.has_null) {  // value *= evalTransmittance(...) / emPdf, scene.cpp:887-889
                    const F3 tr = shadow_transmittance(S, nee.small_tris, nee.stack_col, I.p, ds.sd, ds.sdist, ds.is_env ? 1.0f : 1 - PPG_SHADOW_EPSILON,
                                                       R.max_depth - (int)depth - 1, traced);
                    if (iszero3(tr)) value = f3s(0.0f);
                    else { value = div3(mul3(value, tr), ds.em_pdf); ds.pdf *= ds.em_pdf; }
                } else {
                    ++traced;
                    if (shadow_occluded<FULL>(S, nee.small_tris, nee.stack_col, I.p, ds.sd, ds.sdist * ((FULL && ds.is_env) ? 1.0f : 1 - PPG_SHADOW_EPSILON))) {
                        value = f3s(0.0f);
                    } else {
                        value = div3(value, ds.em_pdf);
                        ds.pdf *= ds.em_pdf;
                    }
                }
            }
            if (!iszero3(value)) {
                const F3 wo_e = to_local(I, ds.d);
                const float woDotGeoNE = dot3(I.geoN, ds.d);
                if (!R.strict_normals || woDotGeoNE * wo_e.z > 0) {
                    const F3 bsdfVal = b_eval(I.wi, wo_e);
                    float woPdfE = 0, bsdfPdfE = 0, dTreePdfE = 0;  // pdfMat, GP:1693-1710
                    if (!T.is_built) {
                        woPdfE = bsdfPdfE = b_pdf(I.wi, wo_e);
                    } else {
                        bsdfPdfE = b_pdf(I.wi, wo_e);
                        if (ppg_isfinite(bsdfPdfE)) {
                            float cx, cy;
                            dir_to_canonical(to_world(I, wo_e), cx, cy);
                            dTreePdfE = dtree_pdf(T, hd, cx, cy, fcol);
                            woPdfE = frac * bsdfPdfE + (1 - frac) * dTreePdfE;
                        }
                    }
                    const float qa = ds.pdf * ds.pdf, qb = woPdfE * woPdfE;
                    const float weightE = qa / (qa + qb);
                    value = mul3(value, bsdfVal);
                    const F3 L = mul3(thr, value) * weightE;
                    if (!R.is_final_iter && R.nee != NEE_ALWAYS) {  // the direct-light vertex, GP:1994-2010
                        Rec rec;
                        if (vertex_to_rec(L, bsdfVal, div3(mul3(thr, bsdfVal), ds.pdf), ds.pdf, bsdfPdfE, dTreePdfE, ds.d, false, 0.5f, rec)) {
                            rec.adamPath = adam_path_id(P, R, i);
                            rec.adamCode = depth < (unsigned int)PPG_ADAM_CODE_VERTEX ? depth : (unsigned int)PPG_ADAM_CODE_VERTEX - 1u;
                            rec.adamPos = 0xffffffffu;
                            commit_single(T, R.spatial_filter, R.directional_filter, T.is_built ? R.loss : LOSS_NONE, leaf, I.p, vox, rec,
                                          key, PPG_DIM_NEE_COMMIT + 3u * depth);
                            ++committed;
                        }
                    }
                    if (!iszero3(L)) {  // recordRadiance, GP:1791-1796
                        Li = Li + L;
                        for (unsigned int v0 = 0; v0 < nV; ++v0) {
                            float4 rr = P.v_rad[(size_t)v0 * P.n_paths + i];
                            rr.x += L.x; rr.y += L.y; rr.z += L.z;
                            P.v_rad[(size_t)v0 * P.n_paths + i] = rr;
                        }
                    }
                }
            }
        }
        if (iszero3(bsdfWeight)) go = false;  // GP:2024-2025
        if (go) {
            const F3 wo = to_world(I, wo_l);
            float woDotGeoN = dot3(I.geoN, wo);
            if (woDotGeoN * wo_l.z <= 0 && R.strict_normals) go = false;  // GP:2031-2032
            if (go) {
                thr = mul3(thr, bsdfWeight);  // GP:2039-2040
                if (FULL) eta *= sampledEta;
                d = wo;
                if (FUSED) {
                    Hit hn = trace_small(lds_tris, S, I.p, wo, PPG_EPSILON, __builtin_inff());
                    P.hit[i] = make_float4(hn.t, hn.u, hn.v, __int_as_float(hn.prim));
                    ++traced;
                } else if (CARRY) {
                    cs->ro = make_float4(I.p.x, I.p.y, I.p.z, PPG_EPSILON);
                } else {
                    P.ray_o[i] = make_float4(I.p.x, I.p.y, I.p.z, PPG_EPSILON);
                }
                if (CARRY) cs->rd = make_float4(wo.x, wo.y, wo.z, __builtin_inff());
                else P.ray_d[i] = make_float4(wo.x, wo.y, wo.z, __builtin_inff());
                if (smooth && nV < PPG_MAX_VERTICES && nV < (unsigned int)R.max_vertices && !R.is_final_iter) {
                    size_t vi = (size_t)nV * P.n_paths + i;
                    F3 bv = bsdfWeight * woPdf;
                    P.v_d[vi] = make_float4(wo.x, wo.y, wo.z, woPdf);
                    P.v_thr[vi] = make_float4(thr.x, thr.y, thr.z, bsdfPdf);
                    P.v_bsdf[vi] = make_float4(bv.x, bv.y, bv.z, dTreePdf);
                    if (P.v_o) {
                        P.v_o[vi] = make_float4(I.p.x, I.p.y, I.p.z, 0.0f);
                        P.v_vox[vi] = make_float4(vox.x, vox.y, vox.z, 0.0f);
                    }
                }
                flags |= FL_PENDING | (smooth ? FL_PEND_TREE : 0u) | (sampledDelta ? FL_PEND_DELTA : 0u);
                if (NEE && dot3(wo, refN) >= 0) flags |= FL_PEND_REFN;
                if (NEE && FULL && P.nee_cos) P.nee_cos[i] = noRefN ? -2.0f : dot3(wo, refN);
                if (FULL && sampledNull) {
                    // GP:2045-2075: a sampled null interaction.  Smooth/null hybrids (mask) record it for the sampling-fraction
                    // optimiser (GP:2047-2068): the slot's d / throughput / bsdfVal were written just above; radiance stays 0, delta.
                    flags |= FL_PEND_NULL;
                    if (R.loss != LOSS_NONE && smooth && nV < PPG_MAX_VERTICES && nV < (unsigned int)R.max_vertices && !R.is_final_iter) {
                        if (1 / woPdf > 0) {
                            P.v_rad[(size_t)nV * P.n_paths + i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((unsigned int)leaf | 0x80000000u));
                            ++nV;
                        }
                    }
                }
                m.w = (unsigned int)leaf;
                l4.w = woPdf;
                alive = true;
            }
        }
    }
    flags = (flags & ~(FL_DEPTH_MASK | FL_NV_MASK)) | (depth & FL_DEPTH_MASK) | (nV << FL_NV_SHIFT);
    if (CARRY && alive) {
        cs->misc = make_uint4(key, dim, flags, m.w);
        cs->li = make_float4(Li.x, Li.y, Li.z, l4.w);
        cs->thr = make_float4(thr.x, thr.y, thr.z, eta);
    } else {
        P.misc[i] = make_uint4(key, dim, flags, m.w);
        P.li[i] = make_float4(Li.x, Li.y, Li.z, l4.w);
        if (alive) P.thr[i] = make_float4(thr.x, thr.y, thr.z, eta);
    }
    if (!alive) plen = depth;  // avgPathLength += rRec.depth, GP:2147-2148
    PROBE_MARK(cs, 9);
    }
    return alive;
}

// One queue slice through Li's loop body.
template <bool FUSED, bool NEE, bool FULL, int MSET = MSET_ALL>
D void shade_slice(const PathState &P, const DevScene &S, const DevTree &T, const RenderParams &R, const Work &work,
                   unsigned int b, unsigned int nb, unsigned int *out_items, unsigned int *out_count, const LdsColumn &fcol,
                   const float4 *lds_tris, unsigned long long &plen_sum, unsigned int &traced, const NeeLds &nee,
                   unsigned long long &committed) {
    const unsigned int rounds = (work.count + PPG_BLOCK - 1) / PPG_BLOCK;
    for (unsigned int r = 0; r < rounds; ++r) {
        unsigned int q = r * PPG_BLOCK + threadIdx.x;
        bool active = q < work.count;
        bool alive = false;
        unsigned long long plen = 0;
        unsigned int i = 0;
        if (active) {
            i = work_item(work, q, b, nb);
            active = i < P.n_paths;
        }
        if (active) {
            alive = shade_one<FUSED, NEE, FULL, false, MSET>(P, S, T, R, i, fcol, lds_tris, plen, traced, nee, committed);
        }
        unsigned int slot = queue_append(out_count, alive);
        if (alive) out_items[slot] = i;
        plen_sum += plen;
    }
}

template <bool FUSED, bool NEE, bool FULL, int MSET = MSET_ALL>
__global__ __launch_bounds__(PPG_BLOCK, (MSET == MSET_COMMON ? PPG_SHADE_WAVES_COMMON : (FULL ? PPG_SHADE_WAVES_FULL : PPG_SHADE_WAVES))) void k_shade(PathState P, DevScene S, DevTree T, RenderParams R, Queues Q, int qin,
                                                                     int small_scene, const unsigned int *sorted_items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float pdf_factors[20 * PPG_BLOCK];  // QuadTreeNode::pdf's per-level factors (tree depth <= 20, GP:1112)
    __shared__ unsigned int out_count;
    __shared__ unsigned long long acc;
    const float4 *lds_tris = (const float4 *)lds_raw;
    // dynamic LDS: the triangles (small scenes) or the BVH stack columns of the rays traced in place (shadow rays, look-through)
    const bool staged = FUSED || ((NEE || (FULL && S.has_null)) && small_scene);
    if (staged) {
        for (int k = threadIdx.x; k < 3 * S.n_tris; k += blockDim.x) ((float4 *)lds_raw)[k] = S.accel_small[k];
    }
    NeeLds nee;
    nee.small_tris = staged ? lds_tris : nullptr;
    nee.stack_col = (int *)lds_raw + threadIdx.x;
    const LdsColumn fcol{pdf_factors + threadIdx.x, PPG_BLOCK};
    const unsigned int b = blockIdx.x, nb = gridDim.x;
    if (Q.stop && *Q.stop) return;
    // QIN_SORTED: the workgroup's share of the dense list re-ordered by the BSDF type at the new hit (k_sort_slices); same paths, other order
    const Work work = work_of(P, Q, qin, sorted_items, b, nb);
    if (threadIdx.x == 0) out_count = qin == QIN_SORTED_REST ? Q.count[0][b] : 0;  // the second launch over a slice appends to the first one's output
    __syncthreads();
    unsigned long long plen_sum = 0, committed = 0;
    unsigned int traced = 0;
    shade_slice<FUSED, NEE, FULL, MSET>(P, S, T, R, work, b, nb, Q.items[0] + (size_t)b * Q.cap, &out_count, fcol, lds_tris, plen_sum, traced,
                                        nee, committed);
    __syncthreads();
    if (threadIdx.x == 0) Q.count[0][b] = out_count;
    block_add_u64(&acc, &Q.stats[b].path_len, plen_sum);
    if (FUSED || NEE || FULL) block_add_u64(&acc, &Q.stats[b].rays, traced);
    if (NEE) block_add_u64(&acc, &Q.stats[b].committed, committed);
}

// Tail of unbounded paths (maxDepth < 0): persistent threads.  The reference's Russian roulette keeps a guided path alive with
// probability 0.99 (GP:2125-2137): after the bulk bounces a few per cent of the paths are left, and a handful of them go on for
// hundreds of bounces.  Every LANE takes one surviving path from the dense list (wave-aggregated global ticket) and carries it
// through trace → shade → trace → ... until it ends, then takes the next one: no queues, no barriers between bounces, all
// workgroups share one list (work stealing), and the run time of the launch is the longest path's chain of dependent bounces
// rather than (number of bounces) x (launch + barrier latency).  Measured on KITCHEN 720p (round 3): the longest path of a
// batch has 300-900 bounces; a wave's bounce takes ~20 us with one live lane and ~120 us with 64 (the union of the lanes' BSDF branches,
// 64 scattered lines per load), so the wave that holds the long path speeds up as the crowd around it dies — which is why this plain
// "one lane per path, until it ends" beat every re-compaction of the survivors into dense waves that was tried (generations of launches
// on shrinking or on full grids, tails on side streams beside the next sub-batch's wavefront: profiles/r03_tail_experiments.json).
template <bool SMALL, bool NEE, bool FULL>
__global__ __launch_bounds__(PPG_BLOCK, (FULL ? PPG_TAIL_WAVES_FULL : PPG_SHADE_WAVES)) void k_tail(PathState P, DevScene S, DevTree T, RenderParams R, const unsigned int *dense,
                                                                    const unsigned long long *total_ptr, unsigned int *ticket, BlockStats *stats,
                                                                    int lds_tris, unsigned int *longest, StragOut so, unsigned int lane_limit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ float pdf_factors[20 * PPG_BLOCK];
    __shared__ int trace_park[3 * PPG_BLOCK];  // a suspended traversal's (entry at hand, stack pointer, original index of the best hit): trace_closest4_resume
    __shared__ unsigned long long acc;
#ifdef PPG_PROBE
    __shared__ unsigned long long probe_lds[PPG_PROBE_SLOTS];
    if (threadIdx.x < PPG_PROBE_SLOTS) probe_lds[threadIdx.x] = 0;
    __syncthreads();
#endif
    const LdsColumn fcol{pdf_factors + threadIdx.x, PPG_BLOCK};
    const unsigned int total = (unsigned int)*total_ptr;
    LdsScene L;
    if (SMALL) L = stage_scene(S, lds_raw, 0, lds_tris);
    else { L.nodes = nullptr; L.tris = nullptr; L.n_nodes = 0; L.n_tris = 0; }
    NeeLds nee;
    nee.small_tris = SMALL ? L.tris : nullptr;     // SMALL: all triangles are staged
    nee.stack_col = (int *)lds_raw + threadIdx.x;  // !SMALL: this lane's BVH stack column (trace and shade never overlap in a lane)
    unsigned long long plen_sum = 0, committed = 0, plen_max = 0;
    unsigned int traced = 0;
    const int lane = threadIdx.x & 63;
    // (lane_limit < 64: a launch over a FEW paths — the stragglers' — deals them thinly, lane_limit paths per wave, so that every one of them
    // runs at the speed of a lone path (cooperative traversal, no union of branches) instead of 64 of them sharing a wave)
    bool have = false, drained = (unsigned int)lane >= lane_limit;
    bool pending = false;  // this lane's traversal is suspended (its path sits out the shading of this iteration)
    unsigned int i = 0;
    Carried cs;
    cs.misc = make_uint4(0u, 0u, 0u, 0u);
    cs.thr = cs.li = cs.hit = cs.ro = cs.rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#ifdef PPG_PROBE
    cs.plds = probe_lds; cs.pt = 0; cs.pon = false;
#endif
    for (;;) {
#ifdef PPG_PROBE
        {   // lone = this wave held exactly one live path when the iteration began (and R.probe is set)
            const unsigned long long live0 = __ballot(have);
            cs.pon = R.probe && have && __popcll(live0) == 1;
            if (cs.pon) { __builtin_amdgcn_s_waitcnt(0); cs.pt = __builtin_readcyclecounter(); cs.prt = __builtin_amdgcn_s_memrealtime(); atomicAdd(&probe_lds[20], 1ull); }
        }
#endif
        const unsigned long long need = __ballot(!have && !drained);
        if (need) {
            const int leader = __ffsll((long long)need) - 1;
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(ticket, (unsigned int)__popcll(need));
            base = __shfl(base, leader);
            if (!have && !drained) {
                const unsigned int k = base + (unsigned int)__popcll(need & ((1ull << lane) - 1ull));
                if (k < total) {
                    i = dense[k]; have = true; pending = false;
                    cs.misc = P.misc[i]; cs.thr = P.thr[i]; cs.li = P.li[i]; cs.ro = P.ray_o[i]; cs.rd = P.ray_d[i];
                } else drained = true;
            }
        }
        bool spilled = false;  // (wave-uniform)
        if (R.defer_depth) {
            // A path that is alive at depth defer_depth — just taken from the list at that depth, or after the bounce below — leaves this
            // launch: its state, as carried, goes to the straggler set (one wave-aggregated ticket), and the word of the batch's own state
            // says "no vertices" — the batch's commit passes it by.  Its lane takes the next path in the next iteration.
            const bool spill = have && (cs.misc.z & FL_DEPTH_MASK) >= R.defer_depth;
            const unsigned long long sm = __ballot(spill);
            if (sm) {
                spilled = true;
                const int leader = __ffsll((long long)sm) - 1;
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(so.count, (unsigned long long)__popcll(sm));
                base = __shfl(base, leader);
                if (spill) {
                    const size_t j = (size_t)base + (size_t)__popcll(sm & ((1ull << lane) - 1ull));
                    float4 *r = so.rec + 8 * j;
                    r[0] = cs.ro; r[1] = cs.rd; r[2] = cs.thr; r[3] = cs.li; r[4] = cs.hit;
                    r[5] = make_float4(__uint_as_float(cs.misc.x), __uint_as_float(cs.misc.y), __uint_as_float(cs.misc.z), __uint_as_float(cs.misc.w));
                    so.orig[j] = i;
                    P.misc[i] = make_uint4(cs.misc.x, cs.misc.y, cs.misc.z & ~FL_NV_MASK, cs.misc.w);
                    have = false; pending = false;  // (a suspended traversal is not carried over: the stragglers' launch traces the ray anew)
                }
            }
        }
        const unsigned long long live = __ballot(have);
        if (!live) {
            if (spilled) continue;  // (the lanes that just gave their paths away have not asked for new ones yet)
            break;
        }
        PROBE_MARK(&cs, 0);
        bool traced_coop = false;
        if (!SMALL && __popcll(live) <= PPG_COOP_MAX) {
            // a handful of live paths in this wave: each of their rays is traversed by the WHOLE wave (trace_closest4_wave)
            for (unsigned long long todo = live; todo; todo &= todo - 1ull) {
                const int src = __ffsll((long long)todo) - 1;
                const F3 o = f3(__shfl(cs.ro.x, src), __shfl(cs.ro.y, src), __shfl(cs.ro.z, src));
                const F3 d = f3(__shfl(cs.rd.x, src), __shfl(cs.rd.y, src), __shfl(cs.rd.z, src));
                float mint = __shfl(cs.ro.w, src);
                const float maxt = __shfl(cs.rd.w, src);
                if (mint == PPG_EPSILON)  // adaptive ray epsilon, skdtree.cpp:125-129
                    mint *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
#ifdef PPG_PROBE
                int psteps = 0;
                const Hit h = trace_closest4_wave(S, (int *)lds_raw + (threadIdx.x & ~63), PPG_BLOCK, o, d, mint, maxt, &psteps);
                if (cs.pon) { atomicAdd(&probe_lds[21], (unsigned long long)psteps); atomicAdd(&probe_lds[22], 1ull); }
#else
                const Hit h = trace_closest4_wave(S, (int *)lds_raw + (threadIdx.x & ~63), PPG_BLOCK, o, d, mint, maxt);
#endif
                if (lane == src) {
                    pending = false;  // (a suspended traversal is dropped: the wave has traced the ray from its origin)
                    Hit hs = h;
                    if (h.prim == -2) hs = trace_closest4<false, true, false>(S, nee.stack_col, PPG_BLOCK, o, d, mint, maxt);  // wave stack overrun: this lane alone
                    cs.hit = make_float4(hs.t, hs.u, hs.v, __int_as_float(hs.prim)); ++traced;
                }
            }
            traced_coop = true;
        }
        PROBE_MARK(&cs, 1);
        if (have) {
            const float4 ro = cs.ro, rd = cs.rd;
            const F3 o = f3(ro.x, ro.y, ro.z), d = f3(rd.x, rd.y, rd.z);
            Hit h;
            if (traced_coop) {
                h.t = 0; h.u = 0; h.v = 0; h.prim = -1;  // (in cs.hit already)
            } else if (SMALL) {
                h = trace_small(L.tris, S, o, d, ro.w, rd.w);
            } else {
                float mint = ro.w;
                if (mint == PPG_EPSILON)  // adaptive ray epsilon, skdtree.cpp:125-129
                    mint *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
                if (pending) { h.t = cs.hit.x; h.u = cs.hit.y; h.v = cs.hit.z; h.prim = __float_as_int(cs.hit.w); }
                pending = !trace_closest4_resume<true>(S, nee.stack_col, PPG_BLOCK, o, d, mint, rd.w, pending, trace_park + threadIdx.x, PPG_BLOCK, h, PPG_TAIL_SUSPEND);
            }
            if (!traced_coop) { cs.hit = make_float4(h.t, h.u, h.v, __int_as_float(h.prim)); if (!pending) ++traced; }
            if (!pending) {
                unsigned long long plen = 0;
                const bool alive = shade_one<false, NEE, FULL, true>(P, S, T, R, i, fcol, L.tris, plen, traced, nee, committed, &cs);
                plen_sum += plen;
                if (plen > plen_max) plen_max = plen;
                if (!alive) have = false;
            }
        }
        PROBE_MARK(&cs, 10);
#ifdef PPG_PROBE
        if (cs.pon) atomicAdd(&probe_lds[11], __builtin_amdgcn_s_memrealtime() - cs.prt);  // 100 MHz ticks of the lone iterations: cycles / ticks = the clock
#endif
    }
#ifdef PPG_PROBE
    __syncthreads();
    if (R.probe && threadIdx.x < PPG_PROBE_SLOTS && probe_lds[threadIdx.x]) atomicAdd(&R.probe[threadIdx.x], probe_lds[threadIdx.x]);
#endif
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(plen_max, off); if (o > plen_max) plen_max = o; }
    if (lane == 0 && plen_max) { atomicMax(&stats[blockIdx.x].max_len, plen_max); atomicMax(longest, (unsigned int)plen_max); }  // *longest: this launch's longest path
    block_add_u64(&acc, &stats[blockIdx.x].path_len, plen_sum);
    block_add_u64(&acc, &stats[blockIdx.x].rays, traced);
    if (NEE) block_add_u64(&acc, &stats[blockIdx.x].committed, committed);
}

// Scenes with many BSDF types (k_shade<FULL>): after the first bounce the rays of a wave hit unrelated surfaces, and a wave executes the
// union of its lanes' BSDF branches (rough plastic, rough conductor, glass, ...).  Every workgroup therefore counting-sorts its queue
// slice by the BSDF type at the new hit (16 bins; rays that left the scene last, so that the lanes of their waves finish together).
// The order of a slice has no influence on any result: per-path random numbers, integer accumulation.
static __global__ __launch_bounds__(PPG_BLOCK) void k_sort_slices(PathState P, DevScene S, Queues Q, int qin, unsigned int *sorted, unsigned char *keys) {
    __shared__ unsigned int hist[16], offs[16];
    const unsigned int b = blockIdx.x, nb = gridDim.x;
    if (Q.stop && *Q.stop) return;
    const Work work = work_of(P, Q, qin, nullptr, b, nb);  // QIN_FIRST (every path of the batch) or QIN_DENSE
    if (work.count == 0) { if (threadIdx.x == 0) { Q.count[1][b] = 0; if (Q.n_common) Q.n_common[b] = 0; } return; }
    sort_slice(sort_args(P, S, Q, b, sorted, keys), work, b, nb, hist, offs);
}

// copy every workgroup's queue slice into one dense array (offsets = exclusive scan of the slice counts)
static __global__ void k_gather_slices(const unsigned int *items, const unsigned int *count, const unsigned int *offsets, unsigned int cap,
                                unsigned int *dense) {
    const unsigned int b = blockIdx.x, n = count[b], off = offsets[b];
    for (unsigned int k = threadIdx.x; k < n; k += blockDim.x) dense[off + k] = items[(size_t)b * cap + k];
}

// After a bounce: offsets[b] = exclusive scan of the output slice counts (k_gather_slices builds the dense list from them), *total_out =
// the number of live paths, also recorded per bounce for the host (which sizes the next batch's schedule by it).  When fewer than
// `stop_below` paths are left, *stop is set: the wavefront kernels of the bounces that were launched beyond this one return at once and
// the persistent threads (k_tail) take the dense list as it is now.  One workgroup of 1024 threads.
static __global__ __launch_bounds__(1024) void k_scan_counts(const unsigned int *counts, unsigned int *offsets, unsigned int n, unsigned long long *total_out,
                                                      unsigned int *bounce_count, unsigned int *stop, unsigned int stop_below) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    const bool stopped = stop && *stop;  // (read by every thread before thread 0 may write it)
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    if (stopped) return;
    for (unsigned int start = 0; start < n; start += 1024) {
        unsigned int i = start + threadIdx.x;
        unsigned long long v = i < n ? counts[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (unsigned int off = 1; off < 1024; off <<= 1) {
            unsigned long long t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) offsets[i] = (unsigned int)(carry + part[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (bounce_count) *bounce_count = (unsigned int)carry;
        if (stop && carry < (unsigned long long)stop_below) *stop = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// k_commit — Vertex::commit for every recorded vertex of every path (GP:1730-1768, 2150-2154)
// ------------------------------------------------------------------------------------------------
// Work item w = (vertex slot v, path i), slot-major: a wave handles the same slot of 64 neighbouring paths, whose
// vertices tend to lie in the same S-tree leaf (so the per-leaf counters combine) and whose loads coalesce.
// The stochastic filter's three draws for slot v use sampler dimensions dim_end + 3v .. +2 (dim_end = the
// path's dimension counter when Li returned) — the same rule as the oracle.
// Which paths: all of them; or (skip) those not flagged — the paths that had ended when the persistent-thread tail took over, committed
// on a second stream while k_tail runs —; or (list) the flagged ones afterwards.
template <int SF, int DF>
__global__ __launch_bounds__(PPG_BLOCK) void k_commit(PathState P, DevTree T, RenderParams R, Queues Q, const unsigned char *nv8,
                                                      const unsigned int *list, const unsigned long long *list_n) {
    __shared__ unsigned long long acc;
    __shared__ unsigned long long box_lds[DF == DF_BOX ? PPG_BOX_STACK * PPG_BLOCK : 1];  // the box splat's stack, one column per lane
    unsigned long long committed_sum = 0;
    const float statisticalWeight = (R.nee == NEE_KICKSTART && R.do_nee) ? 0.5f : 1.0f;
    const int loss = T.is_built ? R.loss : LOSS_NONE;
    const unsigned int n_sel = list ? (unsigned int)*list_n : P.n_paths;
    const unsigned long long items = (unsigned long long)R.max_vertices * n_sel;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long w0 = (unsigned long long)blockIdx.x * blockDim.x; w0 < items; w0 += stride) {
        const unsigned long long w = w0 + threadIdx.x;
        bool act = w < items;
        unsigned int v = 0, i = 0, key = 0, dim = 0;
        if (act) {
            v = (unsigned int)(w / n_sel); i = (unsigned int)(w % n_sel);
            if (nv8) act = v < (unsigned int)nv8[i];  // (k_commit_prepare: 0 for the paths this launch leaves to the one after the tail; [_list]: by list position)
            if (list && act) i = list[i];
            if (act) {
                uint4 m = P.misc[i];
                key = m.x; dim = m.y + 3u * v;
                act = v < ((m.z & FL_NV_MASK) >> FL_NV_SHIFT);
            }
        }
        if (!__any(act)) continue;
        Rec rec;
        rec.d = f3s(0.0f); rec.radiance = rec.product = rec.woPdf = rec.bsdfPdf = rec.dTreePdf = 0; rec.statisticalWeight = statisticalWeight;
        rec.isDelta = false; rec.adamPath = 0; rec.adamCode = 0; rec.adamPos = 0xffffffffu;
        int leaf = 0;
        const size_t vi = (size_t)v * P.n_paths + i;
        if (act) {
            float4 a = P.v_d[vi], b = P.v_thr[vi], c = P.v_bsdf[vi], e = P.v_rad[vi];
            const float woPdf = a.w, bsdfPdf = b.w, dTreePdf = c.w;
            F3 radiance = f3(e.x, e.y, e.z), bsdfVal = f3(c.x, c.y, c.z), throughput = f3(b.x, b.y, b.z);
            unsigned int bits = __float_as_uint(e.w);
            const bool isDelta = (bits & 0x80000000u) != 0;
            if (!vertex_to_rec(radiance, bsdfVal, throughput, woPdf, bsdfPdf, dTreePdf, f3(a.x, a.y, a.z), isDelta, statisticalWeight, rec)) {
                act = false;
            } else {
                leaf = (int)(bits & 0x7fffffffu);
                ++committed_sum;
                if (loss != LOSS_NONE) {
                    rec.adamPath = adam_path_id(P, R, i);
                    rec.adamCode = (unsigned int)PPG_ADAM_CODE_VERTEX + v;
                    rec.adamPos = T.adam_base ? T.adam_base[i] + v : 0xffffffffu;
                }
            }
        }
        if (SF == SF_BOX) {
            if (act) {
                float4 o4 = P.v_o[vi], x4 = P.v_vox[vi];
                stree_record_box(T, f3(o4.x, o4.y, o4.z), f3(x4.x, x4.y, x4.z), rec, DF, loss);
            }
        } else {
            if (SF == SF_STOCHASTIC && act) {  // GP:1746-1763
                float4 o4 = P.v_o[vi], x4 = P.v_vox[vi];
                leaf = stochastic_leaf(T, f3(o4.x, o4.y, o4.z), f3(x4.x, x4.y, x4.z), key, dim);
            }
            wrapper_record<true>(T, leaf, rec, DF, loss, act, DF == DF_BOX ? box_lds + threadIdx.x : nullptr, PPG_BLOCK);
        }
    }
    block_add_u64(&acc, &Q.stats[blockIdx.x].committed, committed_sum);
}

// ------------------------------------------------------------------------------------------------
// The commit of a ROUND of the sampling-fraction optimiser: records first, splats by D-tree afterwards
// ------------------------------------------------------------------------------------------------
// In a round (include/ppg.h "Learning the BSDF sampling fraction") every committed vertex already leaves a record at a position known in
// advance (DevTree::adam_base), and the round ends with a stable sort of the record keys by S-tree leaf.  So the splat into the building
// tree (DTree::recordIrradiance, GP:395-413) need not happen where the vertex is read — one lane per vertex, every lane walking a different
// D-tree through L2 and adding to it with global atomics: k_commit, bound by exactly that — but can follow the sort:
//   k_commit_records  Vertex::commit (GP:1730-1768) up to the record: per (slot, path) item the vertex is read, turned into a DTreeRecord, the
//                     stochastic filter's leaf looked up; written are the 64-bit key, the optimiser's 32-byte record (if product > 0) and a
//                     16-byte SPLAT record (canonical direction, irradiance, statistical weight).  No atomics, no tree walk.
//   radix sort        by (flag, leaf) — the sort the optimiser needs anyway; a record that is a splat only (product <= 0) carries a flag bit
//                     just above its leaf bits and sorts behind all records of the optimiser, which therefore sees exactly the keys and the
//                     order it saw before (k_adam_apply, the round hook's exchange).
//   k_splat_sorted    the sorted records in chunks; a workgroup stages the building D-tree of a run of equal leaves in LDS (topology 8 B,
//                     accumulators 32 B per node), its lanes splat into it with LDS atomics, and the non-zero accumulators are added to the
//                     pool once per run.  Integer sums: the same bits as k_commit's.
#define PPG_SPLAT_NODES 512    // D-trees of up to this many nodes are staged in LDS (20 KB); larger ones are splatted in the pool (TreeGlobal)
#define PPG_SPLAT_PER_LANE 8   // records per lane and chunk
#define PPG_SPLAT_CHUNK (PPG_BLOCK * PPG_SPLAT_PER_LANE)
#ifndef PPG_SPLAT_MAX_RUNS
#define PPG_SPLAT_MAX_RUNS 64   // a chunk (2048 records) of more runs than this is splatted record by record
#endif
#ifndef PPG_SPLAT_RUN_FACTOR
#define PPG_SPLAT_RUN_FACTOR 4   // a run is staged in LDS when it holds at least nodes / 4 records
#endif
static_assert(PPG_BLOCK % 64 == 0, "k_splat_sorted / k_commit_records reduce over full waves");
static_assert(PPG_SPLAT_NODES * (sizeof(ushort4) + 4 * sizeof(unsigned long long)) + PPG_BOX_STACK * PPG_BLOCK * sizeof(unsigned long long) + 64 <= 64 * 1024,
              "k_splat_sorted: the staged D-tree and the box stacks must fit the workgroup's LDS");

// PATH-major: four lanes take one path and walk its vertex slots; a wave's 16 neighbouring paths read four slots at a time (the slots are
// stored [slot][path]) and own ONE contiguous run of record positions (adam_base is the scan of the vertex counts), which they fill within
// a few iterations — the partial lines meet in L2 before they reach HBM.  (Work items (slot, path) as in k_commit — slot v of all paths, then slot
// v + 1 — wrote every record into a line whose neighbours followed a whole sweep later, tested a byte for the 6 of 7 items beyond their
// path's last vertex, and divided 64-bit item numbers: 48 ms of a 127-pass render)
template <int SF>
__global__ __launch_bounds__(PPG_BLOCK) void k_commit_records(PathState P, DevTree T, RenderParams R, Queues Q, const unsigned char *nv8,
                                                              const unsigned int *list, const unsigned long long *list_n, float4 *splat,
                                                              unsigned int flag_shift) {
    __shared__ unsigned long long acc;
    unsigned long long committed_sum = 0;
    const float statisticalWeight = (R.nee == NEE_KICKSTART && R.do_nee) ? 0.5f : 1.0f;
    const int loss = T.is_built ? R.loss : LOSS_NONE;
    const unsigned int n_sel = list ? (unsigned int)*list_n : P.n_paths;
    // FOUR lanes per path (q = lane & 3 takes vertices q, q + 4, ...): a path's records are adjacent, so each step of a quad writes up to
    // four neighbouring keys (32 B), optimiser records (128 B) and splat records (64 B) — whole sectors instead of one record per line and
    // step (profiles/r05_pmc_traffic_kitchen.json: 302 B of HBM traffic per vertex against 156 algorithmic with one lane per path)
    const unsigned int q = threadIdx.x & 3u;
    const unsigned int stride = (gridDim.x * blockDim.x) >> 2;
    for (unsigned int k0 = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >> 2; k0 < n_sel; k0 += stride) {  // (k0: the wave's first entry)
        const unsigned int k = k0 + ((threadIdx.x & 63u) >> 2);
        unsigned int nv = 0, i = 0;
        if (k < n_sel) {
            nv = nv8[k];  // (k_commit_prepare: 0 for a path this launch leaves to the one after the tail; [_list]: by list position)
            i = list ? list[k] : k;
        }
        if (!__any(nv != 0)) continue;
        uint4 m = make_uint4(0, 0, 0, 0);
        unsigned int base = 0, pathId = 0;
        if (nv) {
            m = P.misc[i];
            const unsigned int have = (m.z & FL_NV_MASK) >> FL_NV_SHIFT;
            nv = nv < have ? nv : have;
            base = T.adam_base[i];
            pathId = adam_path_id(P, R, i);
        }
        for (unsigned int v = q; __any(v < nv); v += 4u) {
            if (!(v < nv)) continue;
            const size_t vi = (size_t)v * P.n_paths + i;
            const float4 a = P.v_d[vi], b = P.v_thr[vi], c = P.v_bsdf[vi], e = P.v_rad[vi];
            const unsigned int bits = __float_as_uint(e.w);
            Rec rec;
            if (!vertex_to_rec(f3(e.x, e.y, e.z), f3(c.x, c.y, c.z), f3(b.x, b.y, b.z), a.w, b.w, c.w, f3(a.x, a.y, a.z), (bits & 0x80000000u) != 0, statisticalWeight, rec))
                continue;
            ++committed_sum;
            int leaf = (int)(bits & 0x7fffffffu);
            if (SF == SF_STOCHASTIC) {  // GP:1746-1763
                const float4 o4 = P.v_o[vi], x4 = P.v_vox[vi];
                leaf = stochastic_leaf(T, f3(o4.x, o4.y, o4.z), f3(x4.x, x4.y, x4.z), m.x, m.y + 3u * v);
            }
            // DTreeWrapper::record (GP:575-584), written down: the splat ...
            float irradiance = 0, px = 0, py = 0, sw = 0;
            if (!rec.isDelta && ppg_isfinite(rec.statisticalWeight) && rec.statisticalWeight > 0) {
                sw = rec.statisticalWeight;
                irradiance = rec.radiance / rec.woPdf;
                if (ppg_isfinite(irradiance) && irradiance > 0) dir_to_canonical(rec.d, px, py);
                else irradiance = 0;  // the statistical weight only (recordIrradiance, GP:396-398)
            }
            // ... and the optimiser's call (GP:581-583)
            const bool optimise = loss != LOSS_NONE && rec.product > 0;
            if (!optimise && !(sw > 0)) continue;  // nothing to do for this vertex: its position stays a hole
            const unsigned int pos = base + v;
            unsigned long long key = ((unsigned long long)(unsigned int)leaf << PPG_ADAM_LEAF_SHIFT) | ((unsigned long long)pathId << PPG_ADAM_CODE_BITS) |
                                     ((unsigned int)PPG_ADAM_CODE_VERTEX + v);
            if (optimise) {
                AdamRec r;
                r.key = key; r.product = rec.product; r.woPdf = rec.woPdf; r.bsdfPdf = rec.bsdfPdf; r.dTreePdf = rec.dTreePdf; r.weight = rec.statisticalWeight; r.pad = 0.0f;
                T.adam_recs[pos] = r;
            } else key |= 1ull << flag_shift;
            T.adam_keys[pos] = key;
            splat[pos] = make_float4(px, py, irradiance, sw);
        }
    }
    block_add_u64(&acc, &Q.stats[blockIdx.x].committed, committed_sum);
}

// one D-tree staged in LDS (k_splat_sorted)
struct TreeLds {
    const ushort4 *child;
    unsigned long long *acc;
    D ushort4 node(unsigned int n) const { return child[n]; }
    D void add(unsigned int n, int slot, unsigned long long v) const { atomicAdd(&acc[n * 4u + (unsigned int)slot], v); }
};

// keys / idx: the round's records sorted by (flag, leaf), idx[t] = position of record t in `splat`; n = positions (holes, key ~0, at the end);
// leaf_bits: the sort field is key >> PPG_ADAM_LEAF_SHIFT, leaf_bits + 1 wide
template <int DF>
__global__ __launch_bounds__(PPG_BLOCK) void k_splat_sorted(DevTree T, const unsigned long long *keys, const unsigned int *idx, const float4 *splat, unsigned int n,
                                                            unsigned int leaf_bits, unsigned int lds_nodes) {
    __shared__ ushort4 s_child[PPG_SPLAT_NODES];
    __shared__ unsigned long long s_acc[4 * PPG_SPLAT_NODES];
    __shared__ unsigned long long s_stack[DF == DF_BOX ? PPG_BOX_STACK * PPG_BLOCK : 1];
    __shared__ unsigned long long s_weight;
    __shared__ unsigned int s_field, s_end, s_runs;
    // (leaf_bits <= 22: flag + leaf fit the 24 key bits above PPG_ADAM_LEAF_SHIFT — the host commits larger S-trees with k_commit; n <= 0xfffffff0:
    // lo + t + j * PPG_BLOCK stays below 2^32)
    const unsigned int leaf_mask = (1u << leaf_bits) - 1u, field_mask = (2u << leaf_bits) - 1u;
    const unsigned int chunks = n / PPG_SPLAT_CHUNK + (n % PPG_SPLAT_CHUNK ? 1u : 0u);
    const unsigned int t = threadIdx.x;
    for (unsigned int c = blockIdx.x; c < chunks; c += gridDim.x) {
        const unsigned int lo = c * PPG_SPLAT_CHUNK, hi = (n - lo) < PPG_SPLAT_CHUNK ? n : lo + PPG_SPLAT_CHUNK;
        // this lane's records of the chunk: positions lo + t + j * PPG_BLOCK
        unsigned int field[PPG_SPLAT_PER_LANE], src[PPG_SPLAT_PER_LANE];
#pragma unroll
        for (int j = 0; j < PPG_SPLAT_PER_LANE; ++j) {
            const unsigned int p = lo + t + (unsigned int)j * PPG_BLOCK;
            field[j] = 0xffffffffu; src[j] = 0;
            if (p < hi) { field[j] = (unsigned int)(keys[p] >> PPG_ADAM_LEAF_SHIFT) & field_mask; src[j] = idx[p]; }
        }
        // How many runs does the chunk hold?  (A record starts a run when the record before it — the previous chunk's last one included — has
        // another field.)  Many short runs — the records that are splats only, a few per leaf; any record of a small round — are not worth a
        // barrier-synchronised pass each: the chunk is then splatted record by record, every lane walking the D-tree of its own record's leaf in
        // the pool (what k_commit does), the statistical weights combined per wave.
        // (a chunk whose first and last record lie a few leaves apart cannot hold many runs: the large rounds' chunks — one or two leaves each —
        // skip the count)
        const unsigned int f_first = (unsigned int)(keys[lo] >> PPG_ADAM_LEAF_SHIFT) & field_mask, f_last = (unsigned int)(keys[hi - 1u] >> PPG_ADAM_LEAF_SHIFT) & field_mask;
        const bool count_runs = (f_last & leaf_mask) == leaf_mask || f_last - f_first >= PPG_SPLAT_MAX_RUNS;
        if (t == 0) s_runs = 0u;
        if (count_runs) {
            __syncthreads();
            unsigned int starts = 0;
#pragma unroll
            for (int j = 0; j < PPG_SPLAT_PER_LANE; ++j) {
                const unsigned int p = lo + t + (unsigned int)j * PPG_BLOCK;
                if (p < hi && (field[j] & leaf_mask) != leaf_mask) {
                    const unsigned int before = p == 0 ? 0xffffffffu : ((unsigned int)(keys[p - 1] >> PPG_ADAM_LEAF_SHIFT) & field_mask);
                    starts += before != field[j] ? 1u : 0u;
                }
            }
            for (int off = 32; off > 0; off >>= 1) starts += __shfl_xor(starts, off);
            if ((t & 63u) == 0 && starts) atomicAdd(&s_runs, starts);
            __syncthreads();
        }
        if (count_runs && s_runs > PPG_SPLAT_MAX_RUNS) {
#pragma unroll
            for (int j = 0; j < PPG_SPLAT_PER_LANE; ++j) {
                const unsigned int leaf = field[j] & leaf_mask;
                const bool valid = field[j] != 0xffffffffu && leaf != leaf_mask;
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (valid) r = splat[src[j]];
                const bool act = valid && r.w > 0;
                wave_key_add<3>(T.bweight, leaf, act ? ppg_to_fixed(r.w) : 0ull, act);  // statisticalWeight += w, GP:396-398
                if (act && r.z > 0) {
                    const unsigned int base = T.hdr[leaf].b_base;
                    const TreeGlobal tree{T.bchild + base, T.bacc + (size_t)base * 4};
                    dtree_record_t(tree, r.x, r.y, r.z, r.w, DF, DF == DF_BOX ? s_stack + t : nullptr, PPG_BLOCK);
                }
            }
            __syncthreads();
            continue;
        }
        unsigned int pos = lo;
        for (;;) {  // one run of equal (flag, leaf) after the other
            __syncthreads();
            if (t == (pos - lo) % PPG_BLOCK) {
                unsigned int f = 0xffffffffu;
#pragma unroll
                for (int j = 0; j < PPG_SPLAT_PER_LANE; ++j) if ((pos - lo) / PPG_BLOCK == (unsigned int)j) f = field[j];
                s_field = f; s_end = hi; s_weight = 0ull;
            }
            __syncthreads();
            const unsigned int f0 = s_field;
            if ((f0 & leaf_mask) == leaf_mask) break;  // holes from here to the end of the array
            {   // the run ends at the first record of a larger field (the keys are sorted)
                unsigned int mine = hi;
#pragma unroll
                for (int j = PPG_SPLAT_PER_LANE - 1; j >= 0; --j) {
                    const unsigned int p = lo + t + (unsigned int)j * PPG_BLOCK;
                    if (p < hi && field[j] > f0) mine = p;
                }
                for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(mine, off); mine = o < mine ? o : mine; }
                if ((t & 63u) == 0 && mine < hi) atomicMin(&s_end, mine);
            }
            const unsigned int leaf = f0 & leaf_mask;
            const unsigned int base = T.hdr[leaf].b_base, nn = T.hdr[leaf].b_num;
            __syncthreads();
            const unsigned int end = s_end;
            // Staging pays for a LONG run: the D-tree's topology in, its accumulators zeroed, every non-zero one added to the pool afterwards —
            // a few microseconds whatever the run holds.  A small round (a rank's share, a round over one group of blocks) leaves a handful of
            // records per leaf, and the records that are splats only (the flag bit: they sort behind the optimiser's, leaf by leaf) come a few
            // per leaf in ANY round: a chunk of 2048 of them was 500 runs, 2 ms for one workgroup while the launch waited (round 6: 2.3 ms of a
            // 4.5 ms region round).  Short runs splat straight into the pool.  Integer sums: the same bits either way.
            const bool staged = nn <= lds_nodes && (end - pos) * PPG_SPLAT_RUN_FACTOR >= nn;  // (lds_nodes: PPG_SPLAT_NODES, or fewer: PPG_SPLAT_LDS_NODES, for the tests)
            if (staged) {
                for (unsigned int k = t; k < nn; k += PPG_BLOCK) {
                    s_child[k] = T.bchild[base + k];
                    s_acc[4 * k] = 0ull; s_acc[4 * k + 1] = 0ull; s_acc[4 * k + 2] = 0ull; s_acc[4 * k + 3] = 0ull;
                }
                __syncthreads();
            }
            unsigned long long wsum = 0ull;
#pragma unroll
            for (int j = 0; j < PPG_SPLAT_PER_LANE; ++j) {
                const unsigned int p = lo + t + (unsigned int)j * PPG_BLOCK;
                if (p < pos || p >= end) continue;
                const float4 r = splat[src[j]];  // (px, py, irradiance, statistical weight)
                if (!(r.w > 0)) continue;
                wsum += ppg_to_fixed(r.w);  // statisticalWeight += w, GP:396-398
                if (!(r.z > 0)) continue;
                if (staged) {
                    const TreeLds tree{s_child, s_acc};
                    dtree_record_t(tree, r.x, r.y, r.z, r.w, DF, DF == DF_BOX ? s_stack + t : nullptr, PPG_BLOCK);
                } else {
                    const TreeGlobal tree{T.bchild + base, T.bacc + (size_t)base * 4};
                    dtree_record_t(tree, r.x, r.y, r.z, r.w, DF, DF == DF_BOX ? s_stack + t : nullptr, PPG_BLOCK);
                }
            }
            for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor(wsum, off);
            if ((t & 63u) == 0 && wsum) atomicAdd(&s_weight, wsum);
            __syncthreads();
            if (staged)
                for (unsigned int k = t; k < 4 * nn; k += PPG_BLOCK) {
                    const unsigned long long v = s_acc[k];
                    if (v) atomicAdd(&T.bacc[(size_t)base * 4 + k], v);
                }
            if (t == 0 && s_weight) atomicAdd(&T.bweight[leaf], s_weight);
            pos = end;
            if (pos >= hi) break;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Per-round application of the sampling-fraction optimiser's records
// ------------------------------------------------------------------------------------------------
// compact[i] += Σ_r rep[i][r]; rep = 0.  Idempotent (a second call adds zeros).
static __global__ void k_fold_replicas(unsigned long long *compact, unsigned long long *rep, unsigned int n_nodes) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    unsigned long long s = 0;
    for (int r = 0; r < PPG_REPLICAS; ++r) {
        unsigned long long v = rep[(size_t)i * PPG_REPLICAS + r];
        if (v) { s += v; rep[(size_t)i * PPG_REPLICAS + r] = 0; }
    }
    if (s) compact[i] += s;
}

// nv[i] = number of vertices path i recorded = the number of positions it owns in the Adam record buffer (fast mode)
// (a path still alive when the tail takes over — `straggler` — reserves the maximum; what it does not use stays a hole)
static __global__ void k_path_nv(PathState P, unsigned int *nv, const unsigned char *straggler, unsigned int max_vertices) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_paths) nv[i] = (straggler && straggler[i]) ? max_vertices : (P.misc[i].z & FL_NV_MASK) >> FL_NV_SHIFT;
}
// Before k_commit: nv8[i] = how many vertex slots of path i this commit takes (0 for a path still alive when the persistent threads took over:
// it is committed after the tail) — k_commit tests ONE BYTE per (slot, path) work item, and four fifths of the items lie beyond their path's
// last vertex: reading the 16-byte path word for each of them was 30 GB of a 127-pass render.  With interleaved path records also the
// contiguous copy of the per-path word (key, dim, flags, leaf) that the items that do commit read.
static __global__ void k_commit_prepare(PathState P, uint4 *misc_out, unsigned char *nv8, const unsigned char *straggler) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_paths) return;
    const uint4 m = P.misc[i];
    if (misc_out) misc_out[i] = m;
    nv8[i] = (straggler && straggler[i]) ? (unsigned char)0 : (unsigned char)((m.z & FL_NV_MASK) >> FL_NV_SHIFT);
}
// the same for the launch over a LIST of paths (the tail's, after it): nv8[k] belongs to list[k] — the items read it in order
static __global__ void k_commit_prepare_list(PathState P, const unsigned int *list, const unsigned long long *list_n, unsigned char *nv8) {
    const unsigned int n = (unsigned int)*list_n;
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        nv8[k] = (unsigned char)((P.misc[list[k]].z & FL_NV_MASK) >> FL_NV_SHIFT);
}
// flag[list[k]] = 1 for the n = *list_n entries of a dense path list
static __global__ void k_mark_list(const unsigned int *list, const unsigned long long *list_n, unsigned char *flag) {
    const unsigned int n = (unsigned int)*list_n;
    for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) flag[list[k]] = 1;
}
// a[i] = i, *total = n (the dense list of a batch that goes to the persistent threads without a wavefront bounce)
static __global__ void k_iota_total(unsigned int *a, unsigned int n, unsigned long long *total) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = n;
}
static __global__ void k_iota(unsigned int *a, unsigned int n) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}
// ---- stragglers (ppg_hip.hip "Stragglers") ----
// The vertex slots of the n stragglers, from the batch's slots (path orig[j]) to the compact state's (path j); four lanes per path.
static __global__ void k_extract_vertices(PathState P, PathState Ps, unsigned int n, unsigned int max_vertices) {
    const unsigned int q = threadIdx.x & 3u;
    for (unsigned int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; j < n; j += (gridDim.x * blockDim.x) >> 2) {
        const unsigned int i = Ps.orig[j];
        // (the recorded vertices AND the slot behind them: a live path has a bounce pending, whose vertex was half written when it was sampled —
        // direction, throughput, BSDF value — and is completed, and counted, by the next k_shade step)
        unsigned int nv = ((Ps.misc[j].z & FL_NV_MASK) >> FL_NV_SHIFT) + 1u;
        if (nv > max_vertices) nv = max_vertices;
        for (unsigned int v = q; v < nv; v += 4u) {
            const size_t a = (size_t)v * P.n_paths + i, b = (size_t)v * Ps.n_paths + j;
            Ps.v_d[b] = P.v_d[a]; Ps.v_thr[b] = P.v_thr[a]; Ps.v_bsdf[b] = P.v_bsdf[a]; Ps.v_rad[b] = P.v_rad[a];
            if (P.v_o) { Ps.v_o[b] = P.v_o[a]; Ps.v_vox[b] = P.v_vox[a]; }
        }
        if (q == 0 && P.nee_cos) Ps.nee_cos[j] = P.nee_cos[i];
    }
}
static __global__ void k_extract_nee_cos(PathState P, PathState Ps, unsigned int n) {
    for (unsigned int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) Ps.nee_cos[j] = P.nee_cos[Ps.orig[j]];
}
// out[i] = the optimiser's variable of S-tree node i (DevTree::theta_frozen)
static __global__ void k_copy_theta(const LeafHdr *hdr, unsigned int n, float *out) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = hdr[i].theta;
}
// keep[i] = Li of path i of the batch (its film kernel runs after the stragglers have ended, when the batch's own state is gone)
static __global__ void k_copy_li(PathState P, float4 *keep) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_paths; i += gridDim.x * blockDim.x) keep[i] = P.li[i];
}
// ... and the stragglers' own, once they have ended
static __global__ void k_scatter_li(PathState Ps, unsigned int n, float4 *keep) {
    for (unsigned int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) keep[Ps.orig[j]] = Ps.li[j];
}
// base[perm[r]] = first + r * stride: the record positions of the stragglers' vertices in the round that applies them, in the order of
// their paths (perm = the stragglers sorted by their path's index in its batch)
static __global__ void k_ranked_base(unsigned int *base, const unsigned int *perm, unsigned int n, unsigned int first, unsigned int stride) {
    for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) base[perm[r]] = first + r * stride;
}
static __global__ void k_record_keys(const AdamRec *recs, unsigned long long *keys, unsigned int n) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = recs[i].key;
}
// out[k] = recs[idx[k]] for the first n_valid sorted records (what the round hook hands to the other ranks)
static __global__ void k_gather_records(const AdamRec *recs, const unsigned int *idx, AdamRec *out, unsigned int n) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = recs[idx[i]];
}
// first position whose key is >= bound
D unsigned int adam_lower_bound(const unsigned long long *keys, unsigned int n, unsigned long long bound) {
    unsigned int lo = 0, hi = n;
    while (lo < hi) {
        const unsigned int mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < bound) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Sharded optimiser (include/ppg.h): bounds[r] = first record (in key order) of owner r = first key >= (r * segment) << LEAF_SHIFT
static __global__ void k_owner_bounds(const unsigned long long *keys, unsigned int n, unsigned int segment, unsigned int world, unsigned long long *bounds) {
    const unsigned int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > world) return;
    bounds[r] = r == world ? (unsigned long long)adam_lower_bound(keys, n, ~0ull)
                           : (unsigned long long)adam_lower_bound(keys, n, (unsigned long long)((unsigned long long)r * segment) << PPG_ADAM_LEAF_SHIFT);
}
// AdamOptimizer::State of every S-tree node <-> six 32-bit words per node (theta, iter, m, v, batchGradient, batchAccumulation)
template <bool IMPORT>
__global__ void k_adam_state(LeafHdr *hdr, unsigned int n_nodes, unsigned int *state) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    unsigned int *w = state + 6 * (size_t)i;
    LeafHdr &h = hdr[i];
    if (IMPORT) {
        h.theta = __uint_as_float(w[0]); h.adam_iter = (int)w[1]; h.adam_m = __uint_as_float(w[2]); h.adam_v = __uint_as_float(w[3]);
        h.adam_bg = __uint_as_float(w[4]); h.adam_ba = __uint_as_float(w[5]);
    } else {
        w[0] = __float_as_uint(h.theta); w[1] = (unsigned int)h.adam_iter; w[2] = __float_as_uint(h.adam_m); w[3] = __float_as_uint(h.adam_v);
        w[4] = __float_as_uint(h.adam_bg); w[5] = __float_as_uint(h.adam_ba);
    }
}
static __global__ void k_count_valid(const unsigned long long *keys, unsigned int n, unsigned int *out, unsigned long long bound) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = adam_lower_bound(keys, n, bound);
}
// count[k] = records of the optimiser for the k-th S-tree leaf, order[k] = k: sorted by count (descending) they are k_adam_apply's schedule
static __global__ void k_adam_counts(const unsigned int *leaves, unsigned int n_leaves, const unsigned long long *keys, unsigned int n, unsigned int *count, unsigned int *order) {
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_leaves) return;
    const unsigned int leaf = leaves[k];
    const unsigned int lo = adam_lower_bound(keys, n, (unsigned long long)leaf << PPG_ADAM_LEAF_SHIFT);
    const unsigned int hi = adam_lower_bound(keys, n, (unsigned long long)(leaf + 1u) << PPG_ADAM_LEAF_SHIFT);
    count[k] = hi - lo; order[k] = k;
}

// The deferred optimizeBsdfSamplingFraction calls of one round (include/ppg.h "Learning the BSDF sampling fraction"): one WAVE
// per S-tree leaf walks that leaf's records in key order — 64 records are fetched at once, then applied in that order with
// the reference's arithmetic: the gradient at the current variable (GP:672-691; every lane its own record, once per batch of append()),
// AdamOptimizer::append (GP:85-95) and step (GP:97-109; the same scalar sequence in every lane).  Lane 0 writes the state back.
static __global__ __launch_bounds__(256) void k_adam_apply(LeafHdr *hdr, const unsigned int *leaves, const unsigned int *order, unsigned int n_leaves,
                                                    const unsigned long long *keys, const unsigned int *idx, const AdamRec *recs, unsigned int n, int loss) {
    const unsigned int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int lane = threadIdx.x & 63u;
    if (w >= n_leaves) return;
    const unsigned int leaf = leaves[order ? order[w] : w];  // (order: busiest D-trees first, k_adam_counts)
    const unsigned int lo = adam_lower_bound(keys, n, (unsigned long long)leaf << PPG_ADAM_LEAF_SHIFT);
    const unsigned int hi = adam_lower_bound(keys, n, (unsigned long long)(leaf + 1u) << PPG_ADAM_LEAF_SHIFT);
    if (lo == hi) return;
    const LeafHdr h = hdr[leaf];
    float variable = h.theta, firstMoment = h.adam_m, secondMoment = h.adam_v, batchGradient = h.adam_bg, batchAccumulation = h.adam_ba;
    int iter = h.adam_iter;
    // functions of the variable alone, refreshed after every step (the reference re-evaluates them per record: same values)
    float samplingFraction = logistic(variable);
    float dFraction = samplingFraction * (1 - samplingFraction);
    float l2RegGradient = 0.01f * variable;
    for (unsigned int base = lo; base < hi; base += 64u) {
        float4 pay = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float wgt = 0.0f;
        if (base + lane < hi) {
            const AdamRec *r = recs + idx[base + lane];
            const float4 a = reinterpret_cast<const float4 *>(r)[0], b = reinterpret_cast<const float4 *>(r)[1];
            pay = make_float4(a.z, a.w, b.x, b.y);  // product, woPdf, bsdfPdf, dTreePdf
            wgt = b.z;
        }
        const unsigned int cnt = (hi - base) < 64u ? (hi - base) : 64u;
        // Which records complete a batch depends on the weights only (append(), GP:85-95: batchAccumulation += weight; step once it
        // exceeds batchSize = 1) — so the steps' iteration numbers are known before any gradient is: the lanes work out the learning
        // rates of "their" steps (the two pow() of GP:100 are the longest dependency chain of a step) in parallel, off the serial path.
        unsigned long long stepMask = 0ull;
        {
            // (record t is read with v_readlane — t is uniform —, not with a shuffle through the LDS crossbar: the walk over a D-tree's
            // records is a serial chain, and five ds_bpermute round trips per record were most of it)
            float ba = batchAccumulation;
            for (unsigned int t = 0; t < cnt; ++t) {
                ba += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), (int)t));
                if (ba > 1.0f) { stepMask |= 1ull << t; ba = 0; }
            }
        }
        float myLr = 0.0f;
        if ((stepMask >> lane) & 1ull) {
            const int it = iter + (int)__popcll(stepMask & ((2ull << lane) - 1ull));
            myLr = ppg_adam_learning_rate(0.01f, 0.9f, 0.999f, it);  // GP:100, evaluated in double like the reference (ppg_detmath.h)
        }
        for (unsigned int t = 0; t < cnt;) {
            // Every lane: the gradient of ITS record at the current variable (optimizeBsdfSamplingFraction, GP:672-691).  The records up to the
            // next step see the same variable, so one evaluation — two IEEE divisions, the longest links of the chain — serves the whole
            // batch (two records at weight 1, three at the kick-start's 0.5) instead of one evaluation per record; the lanes beyond the
            // batch compute values nobody reads.  Same expressions, same order of the sums: same bits.
            const float mixPdf = samplingFraction * pay.z + (1 - samplingFraction) * pay.w;
            const float r = pay.x / mixPdf;
            const float ratio = (loss == LOSS_KL) ? r : r * r;
            const float dLoss_dSamplingFraction = -ratio / pay.y * (pay.z - pay.w);
            const float dLoss_dVariable = dLoss_dSamplingFraction * dFraction;
            const float lossGradient = l2RegGradient + dLoss_dVariable;
            const float weighted = lossGradient * wgt;
            bool step = false;
            do {  // AdamOptimizer::append, GP:85-95 (batchSize = 1)
                batchGradient += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(weighted), (int)t));
                batchAccumulation += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), (int)t));
                step = batchAccumulation > 1.0f;
                ++t;
            } while (!step && t < cnt);
            if (step) {
                const float gradient = batchGradient / batchAccumulation;  // step(), GP:97-109
                ++iter;
                const float lr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myLr), (int)(t - 1u)));
                firstMoment = 0.9f * firstMoment + (1 - 0.9f) * gradient;
                secondMoment = 0.999f * secondMoment + (1 - 0.999f) * gradient * gradient;
                variable -= lr * firstMoment / (__builtin_sqrtf(secondMoment) + 1e-08f);
                variable = ppg_min(ppg_max(variable, -20.0f), 20.0f);
                batchGradient = 0;
                batchAccumulation = 0;
                samplingFraction = logistic(variable);
                dFraction = samplingFraction * (1 - samplingFraction);
                l2RegGradient = 0.01f * variable;
            }
        }
    }
    if (lane == 0) {
        LeafHdr o = h;
        o.theta = variable; o.adam_iter = iter; o.adam_m = firstMoment; o.adam_v = secondMoment; o.adam_bg = batchGradient; o.adam_ba = batchAccumulation;
        hdr[leaf] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// k_film — block->put / squaredBlock->put / film->put for the spp samples of each owned pixel
// (GP:1633-1640; box filter ⇒ own pixel, unit weight)
// ------------------------------------------------------------------------------------------------
static __global__ void k_film(PathState P, int spp, float *image, float *sq_image, float *image_w, float *film, float *film_w) {
    unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.n_pix) return;
    unsigned int pixel = P.pixels[k];
    float ir = image[3 * pixel], ig = image[3 * pixel + 1], ib = image[3 * pixel + 2];
    float sr = sq_image[3 * pixel], sg = sq_image[3 * pixel + 1], sb = sq_image[3 * pixel + 2];
    float fr = film[3 * pixel], fg = film[3 * pixel + 1], fb = film[3 * pixel + 2];
    float iw = image_w[pixel], fw = film_w[pixel];
    for (int j = 0; j < spp; ++j) {
        float4 l = P.li[(size_t)j * P.n_pix + k];
        ir += l.x; ig += l.y; ib += l.z;
        sr += l.x * l.x; sg += l.y * l.y; sb += l.z * l.z;
        fr += l.x; fg += l.y; fb += l.z;
        iw += 1.0f; fw += 1.0f;
    }
    image[3 * pixel] = ir; image[3 * pixel + 1] = ig; image[3 * pixel + 2] = ib;
    sq_image[3 * pixel] = sr; sq_image[3 * pixel + 1] = sg; sq_image[3 * pixel + 2] = sb;
    film[3 * pixel] = fr; film[3 * pixel + 1] = fg; film[3 * pixel + 2] = fb;
    image_w[pixel] = iw; film_w[pixel] = fw;
}

// The same for the passes of a FINAL iteration (include/ppg.h "Final iteration: groups of passes"): the samples of a pixel are summed per GROUP
// of passes, in sample order and from whatever the group's partial holds (zero, or an earlier launch's part of the same group), into the
// group's slot of `partials` — one slot = image (3 n), squared image (3 n), weights (n), n = pixels of the whole film.  k_add_groups then adds
// the slots to image / squared image / weights / film in group order.  On one GPU that is the same sum as k_film's up to the association of
// the float additions; it is what lets the groups of a sharded render be rendered whole by different ranks and still add up to the same bits.
static __global__ void k_film_groups(PathState P, int spp, unsigned int group_samples, float *partials, unsigned int slot0, unsigned int slot_stride, unsigned int n_img) {
    unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.n_pix) return;
    const unsigned int pixel = P.pixels[k];
    for (int j0 = 0; j0 < spp; j0 += (int)group_samples) {
        float *slot = partials + (size_t)(slot0 + (unsigned int)(j0 / (int)group_samples) * slot_stride) * 7u * n_img;
        float *im = slot + 3 * (size_t)pixel, *sq = slot + 3 * (size_t)n_img + 3 * (size_t)pixel, *w = slot + 6 * (size_t)n_img + pixel;
        float ir = im[0], ig = im[1], ib = im[2], sr = sq[0], sg = sq[1], sb = sq[2], iw = *w;
        const int j1 = j0 + (int)group_samples < spp ? j0 + (int)group_samples : spp;
        for (int j = j0; j < j1; ++j) {
            const float4 l = P.li[(size_t)j * P.n_pix + k];
            ir += l.x; ig += l.y; ib += l.z;
            sr += l.x * l.x; sg += l.y * l.y; sb += l.z * l.z;
            iw += 1.0f;
        }
        im[0] = ir; im[1] = ig; im[2] = ib; sq[0] = sr; sq[1] = sg; sq[2] = sb; *w = iw;
    }
}
// image += slot, squared image += slot, weights += slot, film += slot (image part), film weights += slot, for slots first .. first + count - 1
// in this order; the slots are zeroed.
static __global__ void k_add_groups(unsigned int n_img, float *partials, unsigned int first, unsigned int count, float *image, float *sq_image, float *image_w,
                                    float *film, float *film_w) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    float im[3] = {image[3 * i], image[3 * i + 1], image[3 * i + 2]}, sq[3] = {sq_image[3 * i], sq_image[3 * i + 1], sq_image[3 * i + 2]};
    float fi[3] = {film[3 * i], film[3 * i + 1], film[3 * i + 2]}, iw = image_w[i], fw = film_w[i];
    for (unsigned int g = first; g < first + count; ++g) {
        float *slot = partials + (size_t)g * 7u * n_img;
        for (int c = 0; c < 3; ++c) {
            const float a = slot[3 * (size_t)i + c], b = slot[3 * (size_t)n_img + 3 * (size_t)i + c];
            im[c] += a; sq[c] += b; fi[c] += a;
            slot[3 * (size_t)i + c] = 0.0f; slot[3 * (size_t)n_img + 3 * (size_t)i + c] = 0.0f;
        }
        const float w = slot[6 * (size_t)n_img + i];
        iw += w; fw += w;
        slot[6 * (size_t)n_img + i] = 0.0f;
    }
    for (int c = 0; c < 3; ++c) { image[3 * i + c] = im[c]; sq_image[3 * i + c] = sq[c]; film[3 * i + c] = fi[c]; }
    image_w[i] = iw; film_w[i] = fw;
}

// per-pixel variance estimate of performRenderPasses (GP:1300-1311); the clamped luminance goes to `lum`, stored x-major
// (index x * H + y) — the order the reference's serial loop sums it in, so the host adds a contiguous array
static __global__ void k_variance(int n, int W, int N, const float *image, const float *sq_image, const float *image_w, float *var_rgb, float *lum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = image_w[i];
    float iw = w != 0 ? 1.0f / w : 0.0f;
    F3 pixel = f3(image[3 * i] * iw, image[3 * i + 1] * iw, image[3 * i + 2] * iw);
    F3 sq = f3(sq_image[3 * i] * iw, sq_image[3 * i + 1] * iw, sq_image[3 * i + 2] * iw);
    F3 localVar = sq - div3(mul3(pixel, pixel), (float)N);
    var_rgb[3 * i] = localVar.x; var_rgb[3 * i + 1] = localVar.y; var_rgb[3 * i + 2] = localVar.z;
    float l = localVar.x * 0.212671f + localVar.y * 0.715160f + localVar.z * 0.072169f;
    const int x = i % W, y = i / W, H = n / W;
    lum[(size_t)x * H + y] = ppg_min(l, 10000.0f);
}

static __global__ void k_normalise(int n, const float *rgb_sum, const float *w, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ww = w[i];
    float iw = ww != 0 ? 1.0f / ww : 0.0f;
    out[3 * i] = rgb_sum[3 * i] * iw; out[3 * i + 1] = rgb_sum[3 * i + 1] * iw; out[3 * i + 2] = rgb_sum[3 * i + 2] * iw;
}

static __global__ void k_axpy(int n, float a, const float *x, float *y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i] * a;
}

// ------------------------------------------------------------------------------------------------
// SD-tree rebuild kernels
// ------------------------------------------------------------------------------------------------
// DTree::reset (GP:456-514): one lane walks the reference's LIFO stack for one S-tree leaf, so the node
// numbering equals the reference's.  WRITE = false only counts nodes (first pass); an exclusive scan over
// the counts gives every leaf its block in the building pool; WRITE = true emits the child indices.
template <bool WRITE>
__global__ void k_dtree_reset(DevTree T, const unsigned int *leaves, unsigned int n_leaves, int newMaxDepth, float rho,
                              unsigned int *counts, ushort4 *bchild_out) {
    unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_leaves) return;
    const unsigned int leaf = leaves[k];
    const LeafHdr h = T.hdr[leaf];
    const float total = h.s_sum;
    const SNode *prev = T.snodes + h.s_base;
    const unsigned int base = WRITE ? h.b_base : 0;
    struct E { unsigned short nodeIndex, otherIndex; unsigned char depth, isThis; float val; };
    E st[64];
    int sp = 0;
    st[sp++] = E{0, 0, 1, 0, 0.0f};
    unsigned int size = 1;
    int maxDepth = 0;
    if (WRITE) bchild_out[base] = make_ushort4(0, 0, 0, 0);
    bool abort = false;
    while (sp && !abort) {
        E s = st[--sp];
        if ((int)s.depth > maxDepth) maxDepth = s.depth;
        unsigned short ch[4] = {0, 0, 0, 0};
        SNode other;
        if (!s.isThis) other = prev[s.otherIndex];
        for (int i = 0; i < 4; ++i) {
            const float osum = s.isThis ? s.val : other.sum[i];
            const float fraction = total > 0 ? (osum / total) : ppg_exp2i(-2 * (int)s.depth);
            if ((int)s.depth < newMaxDepth && fraction > rho) {
                if (!s.isThis && other.child[i] != 0) st[sp++] = E{(unsigned short)size, other.child[i], (unsigned char)(s.depth + 1), 0, 0.0f};
                else st[sp++] = E{(unsigned short)size, (unsigned short)size, (unsigned char)(s.depth + 1), 1, osum / 4};
                ch[i] = (unsigned short)size;
                if (WRITE) bchild_out[base + size] = make_ushort4(0, 0, 0, 0);
                ++size;
                if (size > 65535u) { abort = true; break; }
            }
        }
        if (WRITE) bchild_out[base + s.nodeIndex] = make_ushort4(ch[0], ch[1], ch[2], ch[3]);
    }
    if (!WRITE) counts[k] = size;
    else { T.hdr[leaf].b_depth = maxDepth; }
}

// DTree::build (GP:520-533, 346-366) + `sampling = building` (GP:610-613): children always have larger
// indices than their parent, so one backwards sweep evaluates the recursion's post-order.
static __global__ void k_dtree_build(DevTree T, const unsigned int *leaves, unsigned int n_leaves, SNode *snodes_out) {
    unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_leaves) return;
    const unsigned int leaf = leaves[k];
    LeafHdr h = T.hdr[leaf];
    const unsigned int base = h.b_base;
    for (int n = (int)h.b_num - 1; n >= 0; --n) {
        ushort4 c4 = T.bchild[base + n];
        const unsigned short cc[4] = {c4.x, c4.y, c4.z, c4.w};
        SNode out;
        for (int j = 0; j < 4; ++j) {
            out.child[j] = cc[j];
            if (cc[j] == 0) {
                out.sum[j] = ppg_from_fixed(T.bacc[(size_t)(base + n) * 4 + j]);
            } else {
                const SNode c = snodes_out[base + cc[j]];
                float sum = 0;
                for (int q = 0; q < 4; ++q) sum += c.sum[q];
                out.sum[j] = sum;
            }
        }
        out.pad[0] = out.pad[1] = 0;
        snodes_out[base + n] = out;
    }
    const SNode root = snodes_out[base];
    float sum = 0;
    for (int i = 0; i < 4; ++i) sum += root.sum[i];
    h.s_base = base; h.s_num = h.b_num; h.s_depth = h.b_depth;
    h.s_sum = sum;
    h.s_statw = ppg_from_fixed(T.bweight[leaf]);
    h.b_statw = h.s_statw;
    T.hdr[leaf] = h;
}

// ------------------------------------------------------------------------------------------------
// STree::refine (GP:957-998) + STreeNode::subdivide (GP:876-895) on the device
// ------------------------------------------------------------------------------------------------
// The reference walks the tree with an explicit LIFO stack (child 0 pushed first, so child 1 is visited first) and
// subdivides a leaf while its building weight exceeds the threshold; every subdivision appends two nodes that inherit the
// parent's D-trees with half its weight.  A leaf of weight W therefore becomes a COMPLETE binary subtree of depth
// n = #{halvings until W / 2^n <= threshold}, all of whose 2^n - 1 subdivisions happen consecutively (the stack finishes a
// subtree before anything else), in right-first preorder.  With the old leaves kept in that right-first order (`dfs`), node
// numbers are closed-form:
//   first new node of old leaf j      = n_old + 2 * (number of subdivisions of the leaves before j)      (exclusive scan)
//   children of subdivision e of leaf j = base_j + 2 e, base_j + 2 e + 1                                   (e = preorder index)
// so every subdivision of every leaf can be written independently.  Results are identical to the serial loop, node for node.
static __global__ void k_refine_count(const LeafHdr *hdr, const unsigned int *dfs, unsigned int n_leaves, float threshold, unsigned int *events,
                               unsigned int *new_leaves) {
    unsigned int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    float w = hdr[dfs[j]].b_statw;
    unsigned int n = 0;
    while (w > threshold && n < 30u) { w = w / 2; ++n; }  // shallSplit (GP:953-955); children get half the weight (GP:886)
    events[j] = (1u << n) - 1u;
    new_leaves[j] = 1u << n;
}

// one workgroup per old leaf; thread e handles subdivision e of that leaf's subtree
static __global__ void k_refine_fill(int4 *stree, LeafHdr *hdr, const unsigned int *dfs, unsigned int n_leaves, unsigned int n_old,
                              const unsigned int *events, const unsigned int *ev_off, const unsigned int *lv_off, unsigned int *dfs_out) {
    const unsigned int j = blockIdx.x;
    if (j >= n_leaves) return;
    const unsigned int L = dfs[j];
    const unsigned int n_ev = events[j];
    if (n_ev == 0) {
        if (threadIdx.x == 0) dfs_out[lv_off[j]] = L;
        return;
    }
    unsigned int n = 0;
    while (((1u << n) - 1u) < n_ev) ++n;  // depth of the complete subtree
    __shared__ LeafHdr H;
    __shared__ int axis0;
    if (threadIdx.x == 0) { H = hdr[L]; axis0 = stree[L].x; }
    __syncthreads();
    const unsigned int base = n_old + 2u * ev_off[j];
    LeafHdr cleared;  // cur.dTree = {} (GP:893)
    {
        unsigned int *z = reinterpret_cast<unsigned int *>(&cleared);
        for (unsigned int q = 0; q < sizeof(LeafHdr) / 4; ++q) z[q] = 0u;
    }
    for (unsigned int e = threadIdx.x; e < n_ev; e += blockDim.x) {
        // locate subdivision e: walk down from the leaf; at depth k each child subtree holds S = 2^(n-k-1) - 1 subdivisions,
        // the right child's come first
        unsigned int node = L, k = 0, cur = 0, rem = e, pos = 0;
        while (rem != 0) {
            rem -= 1;
            const unsigned int S = (1u << (n - k - 1)) - 1u;
            if (rem < S) { node = base + 2u * cur + 1u; cur = cur + 1u; }
            else { rem -= S; node = base + 2u * cur; cur = cur + 1u + S; pos += 1u << (n - k - 1); }
            ++k;
        }
        const unsigned int c0 = base + 2u * e, c1 = c0 + 1u;
        stree[node] = make_int4((axis0 + (int)k) % 3, (int)c0, (int)c1, 0);
        hdr[node] = cleared;
        if (k + 1 == n) {  // the children are the new leaves: copies of the old leaf with the weight halved once per level
            LeafHdr h = H;
            float w = h.b_statw;
            for (unsigned int t = 0; t <= k; ++t) w = w / 2;
            h.b_statw = w;
            const int ax = (axis0 + (int)k + 1) % 3;
            stree[c0] = make_int4(ax, 0, 0, 0); hdr[c0] = h;
            stree[c1] = make_int4(ax, 0, 0, 0); hdr[c1] = h;
            dfs_out[lv_off[j] + pos] = c1;       // right child first
            dfs_out[lv_off[j] + pos + 1u] = c0;
        }
    }
}

// exclusive scan of `counts` (n small: one S-tree leaf each) by a single workgroup; total → *total_out
static __global__ void k_scan_exclusive(const unsigned int *counts, unsigned int *offsets, unsigned int n, unsigned long long *total_out) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (unsigned int start = 0; start < n; start += 1024) {
        unsigned int i = start + threadIdx.x;
        unsigned long long v = i < n ? counts[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (unsigned int off = 1; off < 1024; off <<= 1) {
            unsigned long long t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) offsets[i] = (unsigned int)(carry + part[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

static __global__ void k_assign_blocks(DevTree T, const unsigned int *leaves, unsigned int n_leaves, const unsigned int *counts,
                                const unsigned int *offsets) {
    unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_leaves) return;
    T.hdr[leaves[k]].b_base = offsets[k];
    T.hdr[leaves[k]].b_num = counts[k];
}

// batched queries (ppg_query_pdf / ppg_query_sample)
static __global__ void k_query_pdf(DevTree T, unsigned int n, const float *pos, const float *dirs, float *out) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 vox;
    int leaf = stree_lookup(T, f3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), vox);
    float cx, cy;
    dir_to_canonical(f3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), cx, cy);
    RegColumn col;
    out[i] = dtree_pdf<RegColumn &>(T, dtree_ref(T.hdr[leaf]), cx, cy, col);
}
static __global__ void k_query_sample(DevTree T, unsigned int n, const float *pos, unsigned long long seed, float *out) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 vox;
    int leaf = stree_lookup(T, f3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), vox);
    unsigned int key = ppg_path_key(seed, i, 0), dim = 0;
    float cx, cy;
    dtree_sample(T, dtree_ref(T.hdr[leaf]), key, dim, cx, cy);
    F3 d = canonical_to_dir(cx, cy);
    out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}

#endif
