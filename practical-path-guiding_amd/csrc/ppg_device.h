/*
 * ppg_device.h — device-side data layout and per-path building blocks of the guided path tracer.
 *
 * Layout (everything lives in HBM, flat arrays, no pointers inside nodes):
 *   triangles   3 x float4 per triangle in BVH-leaf order: (p0, material), (p1, emitter), (p2, original index)
 *   BVH2        64-byte nodes: both child boxes + child refs (one node visit = one 64 B read)
 *   S-tree      int4 per node {axis, child0, child1, 0}; child0 == 0 ⇔ leaf   (STreeNode, GP:740-845)
 *   leaf hdr    LeafHdr per S-tree node (D-tree descriptors + Adam state)      (DTreeWrapper, GP:570-738)
 *   sampling D-trees  SNode pool: float sum[4] + u16 child[4] + pad = 32 B     (QuadTreeNode, GP:158-371)
 *   building D-trees  bchild pool (u16x4) + bacc pool (u64x4 fixed point), index-aligned with the SNode
 *                     pool of the next iteration
 * GP:n = /root/reference/mitsuba/src/integrators/path/guided_path.cpp:n.
 *
 * All floating-point expressions are written out operation by operation (no FMA contraction, no libm)
 * so that the result of every path is bit-identical to the CPU oracle given the same sampler key.
 */
#ifndef PPG_DEVICE_H
#define PPG_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppg_detmath.h"
#include "../../include/ppg_rng.h"
#include "../../include/ppg.h"

#define D __device__ __forceinline__

struct F3 {
    float x, y, z;
};
D F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
D F3 f3s(float v) { return f3(v, v, v); }
D F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
D F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
D F3 operator*(F3 a, float f) { return f3(a.x * f, a.y * f, a.z * f); }
D F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
D F3 mul3(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
D F3 div3(F3 a, float f) { float r = 1.0f / f; return f3(a.x * r, a.y * r, a.z * r); }  // vector.h:535-542
D float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
D F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
D float len3(F3 v) { return __builtin_sqrtf(dot3(v, v)); }
D F3 norm3(F3 v) { return div3(v, len3(v)); }  // vector.h:625-627
D float comp3(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
D bool iszero3(F3 s) { return s.x == 0.0f && s.y == 0.0f && s.z == 0.0f; }
D bool isvalid3(F3 s) {  // spectrum.h:467-472
    return ppg_isfinite(s.x) && !(s.x < 0.0f) && ppg_isfinite(s.y) && !(s.y < 0.0f) && ppg_isfinite(s.z) && !(s.z < 0.0f);
}
D float avg3(F3 s) { float r = 0.0f; r += s.x; r += s.y; r += s.z; return r * (1.0f / 3); }  // spectrum.h:481-486
D float max3(F3 s) { return ppg_max(ppg_max(s.x, s.y), s.z); }
D F3 ld3(const float4 *p) { float4 v = *p; return f3(v.x, v.y, v.z); }

// ------------------------------------------------------------------------------------------------
// include/ppg_detmath.h on the device: ONE out-of-line copy of each transcendental per kernel module instead of one inlined copy per
// call site.  Pure register-in / register-out functions (no pointers: nothing is forced into scratch memory); the arithmetic is the
// header's, so results are unchanged.  Inlined everywhere they were 4 400 of k_tail<FULL>'s 25 000 instructions — a kernel of 135 KB
// (330 KB before the BSDF call sites were merged) against a 64 KB instruction cache.  -DPPG_INLINE_MATH restores the inlined calls.
// ------------------------------------------------------------------------------------------------
#ifdef PPG_INLINE_MATH
#define DM_ATTR __device__ __forceinline__
#else
#define DM_ATTR static __device__ __attribute__((noinline))
#endif
DM_ATTR float2 dm_sincos(float x) { float s_, c_; ppg_sincos(x, &s_, &c_); return make_float2(s_, c_); }
DM_ATTR float dm_atan2(float y, float x) { return ppg_atan2(y, x); }
DM_ATTR float dm_exp(float x) { return ppg_exp(x); }
DM_ATTR float dm_log(float x) { return ppg_log(x); }
D void dm_sincos(float x, float *s, float *c) { const float2 r = dm_sincos(x); *s = r.x; *c = r.y; }
D float dm_acos(float x) { return dm_atan2(__builtin_sqrtf(ppg_max(0.0f, (1.0f - x) * (1.0f + x))), x); }  // = ppg_acos
D float dm_tan(float x) { const float2 r = dm_sincos(x); return r.x / r.y; }                                 // = ppg_tan
D float dm_pow(float x, float y) { return dm_exp(y * dm_log(x)); }                                            // = ppg_pow

// ------------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------------
struct BvhNode {  // 64 B
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int c0, c1;   // n > 0: first triangle (leaf-order index); n == 0: node index
    int n0, n1;   // > 0 leaf with n triangles, 0 interior, < 0 empty
};

// BVH4 node, 128 B = one L2 line: child boxes in SoA (six float4 loads in flight at once), then child refs.
// child >= 0: interior node index; child < 0: leaf, ~child = (first triangle << 3) | (count - 1); an unused
// slot is marked PPG_BVH4_EMPTY.
#define PPG_BVH4_EMPTY 0x7fffffff
struct Bvh4Node {  // host-side intermediate of the builder
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int child[4];
    int pad[4];
};
// What the kernels traverse: the same node in 64 B (four 16-byte loads instead of seven).  The child boxes are quantised to 8 bits per
// coordinate on a grid spanned by the node's own lower corner and a power-of-two cell size per axis: lo = origin + qlo * 2^e (rounded
// down at build time), hi = origin + qhi * 2^e (rounded up) — every decoded box CONTAINS the builder's padded float box (checked by
// the builder in the device's float arithmetic), so culling stays conservative and the closest hit is unchanged.
struct __attribute__((aligned(16))) Bvh4QNode {
    float ox, oy, oz;
    unsigned int exps;                     // byte a (0, 1, 2): biased exponent of the cell size along axis a, i.e. the float with bits (byte << 23)
    unsigned int qlox, qloy, qloz, qhix;   // child k in byte k
    unsigned int qhiy, qhiz;
    int child[4];
    int pad[2];
};
// decode one node: child boxes as floats, child refs
D void bvh4q_load(const Bvh4QNode *node, float lxs[4], float lys[4], float lzs[4], float hxs[4], float hys[4], float hzs[4], int chs[4]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(node);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const float ox = __uint_as_float(a.x), oy = __uint_as_float(a.y), oz = __uint_as_float(a.z);
    const float sx = __uint_as_float((a.w & 255u) << 23), sy = __uint_as_float(((a.w >> 8) & 255u) << 23), sz = __uint_as_float(((a.w >> 16) & 255u) << 23);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lxs[k] = ox + (float)((b.x >> (8 * k)) & 255u) * sx;
        lys[k] = oy + (float)((b.y >> (8 * k)) & 255u) * sy;
        lzs[k] = oz + (float)((b.z >> (8 * k)) & 255u) * sz;
        hxs[k] = ox + (float)((b.w >> (8 * k)) & 255u) * sx;
        hys[k] = oy + (float)((c.x >> (8 * k)) & 255u) * sy;
        hzs[k] = oz + (float)((c.y >> (8 * k)) & 255u) * sz;
    }
    chs[0] = (int)c.z; chs[1] = (int)c.w; chs[2] = (int)d.x; chs[3] = (int)d.y;
}

struct DevCamera {
    float s2c[16], c2w[16];
    float near_clip, far_clip, inv_w, inv_h;
    int width, height;
};

#define PPG_MAT_STRIDE 6  // float4 per material: (reflectance, type) (specular, alpha) (eta, flags) (k, fdrInt) (opacity, rtrans slice) (texture word, -, -, -)

// ppg_texture on the device: texels as (rgb, -)
struct DevTex {
    const float4 *texels;
    int w, h;
    float su, sv, ou, ov;
    int wrap_u, wrap_v, nearest, pad;
};

struct DevScene {
    const float4 *tris;     // 3 per triangle: positions (.w = material, emitter, original index) — intersection records only
    const float4 *accel;    // 3 per triangle: TriAccel (triaccel.h:36-58), what the traversal reads
    const float4 *accel_small;  // small scenes (brute force from LDS): the same records sorted by projection axis k,
    int small_n[3];             // n[k] of them per axis (degenerate triangles dropped); record[2].z = leaf-order index
    const float4 *normals;  // 3 per triangle or nullptr
    const BvhNode *bvh;
    const Bvh4QNode *bvh4;    // same tree collapsed to 4-wide nodes with quantised child boxes (generic traversal)
    const float4 *bvh_top;    // its top cut for trace_closest4_wave: n_top <= 64 subtrees, (lo, child ref) (hi, -) each; n_top == 0: start at the root
    int n_top;
    const float4 *materials;  // (reflectance rgb, type)
    const float4 *emitters;   // (radiance rgb, -)
    int n_tris;
    DevCamera cam;
    // next-event estimation: one area emitter = the triangles carrying its id, in index order
    int has_null;              // some BSDF has a null (pass-through) component: the look-through code of the FULL kernels is live
    float4 env;                // constant environment emitter (constant.cpp): radiance rgb, w != 0 if present (FULL kernels only);
    float4 bsphere;            // its bounding sphere (constant.cpp:67-78): centre, radius; it is emitter number n_emitters
    int n_emitters;            // area emitters
    const float *em_sel_cdf;   // [n_emitters (+ 1 with an environment emitter) + 1] emitter pmf (Scene::configure, samplingWeight = 1)
    float em_sel_norm;         // DiscreteDistribution::getNormalization()
    const int4 *em_info;       // (first triangle, triangle count, first cdf entry, invSurfaceArea bits)
    const float *em_area_cdf;  // per emitter: count + 1 entries (TriMesh::prepareSamplingTable)
    const float4 *em_tris;     // 3 per emitter triangle: positions
    const float4 *em_normals;  // 3 per emitter triangle or nullptr
    const float *rtrans;       // roughplastic: rough-transmittance slices, rtrans_n + 1 floats each (ppg_scene.rtrans)
    int rtrans_n;
    // analytic spheres (FULL kernels, BVH path only): 4 float4 each = (centre, radius) (R row 0, material) (R row 1, emitter)
    // (R row 2, flip normals); primitive numbers n_tris, n_tris + 1, ..
    const float4 *spheres;
    int n_spheres;
    // image-based environment emitter (envmap.cpp), env.w == 2: texels (rgb, -), the row / column cdfs and sin(theta) row weights
    // of EnvironmentMap::configure, m_normalization, m_scale, the pixel size in spherical coordinates, toWorld's rotation
    const float4 *em_texels;
    const float *em_cdf_rows, *em_cdf_cols, *em_row_weights;
    int em_w, em_h;
    float em_norm, em_scale, em_px, em_py;
    float em_R[9];
    // bitmap textures (FULL kernels): texture coordinates, 3 x float2 per triangle in leaf order (NaN = the mesh has none) or nullptr
    const float2 *uvs;
    const DevTex *textures;
};

struct Hit {
    float t, u, v;
    int prim;  // leaf-order triangle index, -1 = miss
};

// TriAccel::rayIntersect (triaccel.h:99-195) — Wald's pre-projected triangle test, the one Mitsuba's kd-tree leaves run;
// identical arithmetic to the oracle's.  A = 3 float4: (n_u, n_v, n_d, k) (a_u, a_v, b_nu, b_nv) (c_nu, c_nv, -, original index).
// Every lane of a wave tests the same triangle in the brute-force loop, so the switch on k is wave-uniform there.
// The test on a record that is in registers; the arithmetic of TriAccel::rayIntersect in its order.
D bool tri_hit_regs(const float4 a0, const float4 a1, const float4 a2, F3 o, F3 d, float mint, float maxt, float &tt, float &uu, float &vv) {
    const int k = __float_as_int(a0.w);
    // (selects, not three divergent blocks of moves: the lanes of a wave test unrelated triangles; measured equal on KITCHEN, less code)
    if ((unsigned int)k > 2u) return false;
    const bool k0 = k == 0, k1 = k == 1;
    const float o_u = k0 ? o.y : (k1 ? o.z : o.x), o_v = k0 ? o.z : (k1 ? o.x : o.y), o_k = k0 ? o.x : (k1 ? o.y : o.z);
    const float d_u = k0 ? d.y : (k1 ? d.z : d.x), d_v = k0 ? d.z : (k1 ? d.x : d.y), d_k = k0 ? d.x : (k1 ? d.y : d.z);
    const float t = (a0.z - o_u * a0.x - o_v * a0.y - o_k) / (d_u * a0.x + d_v * a0.y + d_k);
    if (t < mint || t > maxt) return false;
    const float hu = o_u + t * d_u - a1.x;
    const float hv = o_v + t * d_v - a1.y;
    const float u = hv * a1.z + hu * a1.w;
    const float v = hu * a2.x + hv * a2.y;
    if (!(u >= 0 && v >= 0 && u + v <= 1.0f)) return false;
    tt = t; uu = u; vv = v;
    return true;
}
// The three words of a record are fetched TOGETHER and waited for once: left to itself the compiler sinks each load to its first use
// behind the test's early-outs, which made a triangle four dependent memory round trips — projection axis, plane equation, edges,
// original index (ISA of round 3's k_trace; even as L1 hits that is four waits of a wave's slowest lane per leaf).  orig = a2.w.
#define PPG_PIN4(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
D bool tri_hit(const float4 *A, F3 o, F3 d, float mint, float maxt, float &tt, float &uu, float &vv, int &orig) {
    float4 a0 = A[0], a1 = A[1], a2 = A[2];
    PPG_PIN4(a0); PPG_PIN4(a1); PPG_PIN4(a2);
    orig = __float_as_int(a2.w);
    return tri_hit_regs(a0, a1, a2, o, d, mint, maxt, tt, uu, vv);
}

// The same test with the projection axis known at compile time (small scenes: triangles are grouped by axis, so the
// component selection is hoisted out of the loop).  (o_u, o_v, o_k) / (d_u, d_v, d_k) are already permuted.
D bool tri_hit_axis(const float4 *A, float o_u, float o_v, float o_k, float d_u, float d_v, float d_k, float mint, float maxt, float &tt,
                    float &uu, float &vv) {
    const float4 a0 = A[0];
    const float t = (a0.z - o_u * a0.x - o_v * a0.y - o_k) / (d_u * a0.x + d_v * a0.y + d_k);
    if (t < mint || t > maxt) return false;
    const float4 a1 = A[1];
    const float hu = o_u + t * d_u - a1.x;
    const float hv = o_v + t * d_v - a1.y;
    const float u = hv * a1.z + hu * a1.w;
    const float4 a2 = A[2];
    const float v = hu * a2.x + hv * a2.y;
    if (!(u >= 0 && v >= 0 && u + v <= 1.0f)) return false;
    tt = t; uu = u; vv = v;
    return true;
}

D float safe_inv(float d) { return d == 0.0f ? 1e30f : 1.0f / d; }

// Scene cache in LDS: the first `n_nodes` BVH nodes and (if they all fit) the triangles.  Small scenes such
// as CBOX live there entirely; for large ones the top of the tree, which every ray visits, does.
struct LdsScene {
    const BvhNode *nodes;
    const float4 *tris;
    int n_nodes, n_tris;
};

// Traversal stack: the first PPG_LDS_STACK entries of every lane live in an LDS column (conflict-free, no
// scratch traffic), deeper entries — rare — in a small private array.
#ifndef PPG_LDS_STACK
#define PPG_LDS_STACK 24
#endif
#define PPG_STACK_OVER (48 - PPG_LDS_STACK)  // entries of a lane's overflow array: LDS column + overflow = 48 entries, whatever the split
#ifndef PPG_LEAF_VOTE_TAIL
#define PPG_LEAF_VOTE_TAIL 16  // leaf vote of k_tail's traversals (trace_closest4<.., VOTE>, trace_closest4_resume).  8 until leaves held up to 4 triangles
                               // and a wave's last traversals were suspended (round 6: 16 +0.8 %, 24 +0.4 % on the driver's command)
#endif
// (the overflow array is a separate object: as a member it kept the whole struct — stack pointer included — in scratch memory, and
// every push / pop went through a scratch load and store)
struct TStack {
    int *lds;      // this lane's column, stride blockDim.x
    int stride;
    int *over;     // int[48 - cap] of the caller
    int sp;
    int cap;       // rows of the LDS column (PPG_LDS_STACK; k_trace: PPG_TRACE_STACK) — a constant after inlining
    D void push(int v) { if (sp < cap) lds[sp * stride] = v; else over[sp - cap] = v; ++sp; }
    D int pop() { --sp; return sp < cap ? lds[sp * stride] : over[sp - cap]; }
    // the children of a node that were hit, farthest first (c1 ends on top): one range check for all three instead of one per push
    D void push_children(int m, int c1, int c2, int c3) {
        if (sp + 3 <= cap) {
            int *q = lds + sp * stride;
            if (m > 3) { *q = c3; q += stride; }
            if (m > 2) { *q = c2; q += stride; }
            if (m > 1) *q = c1;
            sp += m - 1;
            return;
        }
        if (m > 3) push(c3);
        if (m > 2) push(c2);
        if (m > 1) push(c1);
    }
};

// Sphere::rayIntersect (sphere.cpp:164-189) with solveQuadraticDouble (util.cpp:487-525): double precision, like the reference
D bool sphere_hit(const float4 c, F3 ro, F3 rd, float mint, float maxt, float &t) {
    const double ox = (double)ro.x - (double)c.x, oy = (double)ro.y - (double)c.y, oz = (double)ro.z - (double)c.z;
    const double dx = rd.x, dy = rd.y, dz = rd.z;
    const double A = dx * dx + dy * dy + dz * dz;
    const double B = 2 * (ox * dx + oy * dy + oz * dz);
    const double C = (ox * ox + oy * oy + oz * oz) - c.w * c.w;  // m_radius * m_radius is a float product
    double nearT, farT;
    if (A == 0) {
        if (B != 0) nearT = farT = -C / B;
        else return false;
    } else {
        const double discrim = B * B - 4.0f * A * C;
        if (discrim < 0) return false;
        double temp;
        const double sqrtDiscrim = __builtin_sqrt(discrim);
        if (B < 0) temp = -0.5f * (B - sqrtDiscrim);
        else temp = -0.5f * (B + sqrtDiscrim);
        nearT = temp / A;
        farT = C / temp;
        if (nearT > farT) { const double sw = nearT; nearT = farT; farT = sw; }
    }
    if (!(nearT <= maxt && farT >= mint)) return false;
    if (nearT < mint) {
        if (farT > maxt) return false;
        t = (float)farT;
    } else {
        t = (float)nearT;
    }
    return true;
}
// the spheres after the triangles: a sphere wins only with a strictly smaller t (its primitive number is larger)
template <bool ANY>
D void sphere_pass(const DevScene &S, F3 o, F3 d, float mint, float maxt, Hit &best) {
    for (int k = 0; k < S.n_spheres; ++k) {
        float ts;
        if (sphere_hit(S.spheres[4 * k], o, d, mint, best.prim >= 0 ? fminf(maxt, best.t) : maxt, ts) && (best.prim < 0 || ts < best.t)) {
            best.t = ts; best.u = 0; best.v = 0; best.prim = S.n_tris + k;
            if (ANY) return;
        }
    }
}

// The four children of a node after the slab tests, nearest first, WITHOUT indexed local arrays (a dynamically indexed array lives in
// scratch memory: the insertion sort that stood here cost k_trace 120 B of scratch traffic per lane and step).  Children that were
// missed carry t = +inf and sort to the end; a 5-comparator network on registers.  Equal distances may swap — the closest hit does not
// depend on the visiting order (ties are broken by (t, original triangle index)).
#define PPG_CSWAP(ta, ca, tb, cb) { const bool sw_ = (ta) > (tb); const float t0_ = sw_ ? (tb) : (ta), t1_ = sw_ ? (ta) : (tb); \
                                    const int c0_ = sw_ ? (cb) : (ca), c1_ = sw_ ? (ca) : (cb); ta = t0_; tb = t1_; ca = c0_; cb = c1_; }
struct Bvh4Hits { int c0, c1, c2, c3, m; };
// Slab tests of a node's four child boxes.  The traversal is bound by VALU issue slots (k_trace: 73 % of all SIMD cycles issue a vector
// instruction, profiles/r03_pmc_wait_cycles.json), and decoding the boxes to world space first — 24 x (convert, scale, add origin), then
// 24 x (subtract ray origin, multiply) — was half of a node step.  Instead the RAY is taken to the node's grid: the cell sizes are powers
// of two, so  o' = (o - origin) / cell  and  1/d' = (1/d) * cell  are exact scalings, and a plane distance is
//     t = (q - o') * (1/d')                  q = the plane's byte
// — one convert, one subtract and one multiply per plane, the two planes of an axis in one packed instruction each (v_pk_add_f32 /
// v_pk_mul_f32).  Which plane of an axis is the near one follows from the sign of 1/d: selected once per axis and node on the packed
// bytes instead of min / max per child.  Against the exact plane  origin + q * cell  the computed t carries the rounding of (o - origin)
// — a shift of at most 2^-24 |o - origin| along that axis, which matters only where it is not small against the plane distance, i.e. for
// ray origins within the node's own extent: 6e-8 of the node's size, 3 % of the builder's box padding — and two further roundings, covered by
// the factor on the exit distance as before.  The builder checks the exact planes (in double) as well as the float decode that
// trace_closest4_wave still uses.
typedef float ppg_v2f __attribute__((ext_vector_type(2)));
D Bvh4Hits bvh4_children(const Bvh4QNode *node, F3 o, F3 id, float mint, float tlim) {
    const uint4 *q = reinterpret_cast<const uint4 *>(node);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const unsigned int ex = a.w & 255u, ey = (a.w >> 8) & 255u, ez = (a.w >> 16) & 255u;  // 64..154 (BvhBuilder::quantise): no overflow below
    const float osx = (o.x - __uint_as_float(a.x)) * __uint_as_float((254u - ex) << 23), idx = id.x * __uint_as_float(ex << 23);
    const float osy = (o.y - __uint_as_float(a.y)) * __uint_as_float((254u - ey) << 23), idy = id.y * __uint_as_float(ey << 23);
    const float osz = (o.z - __uint_as_float(a.z)) * __uint_as_float((254u - ez) << 23), idz = id.z * __uint_as_float(ez << 23);
    const bool nx = id.x < 0, ny = id.y < 0, nz = id.z < 0;  // (1/d is never -0, NaN or infinite: safe_inv)
    const unsigned int nearx = nx ? b.w : b.x, farx = nx ? b.x : b.w;
    const unsigned int neary = ny ? c.x : b.y, fary = ny ? b.y : c.x;
    const unsigned int nearz = nz ? c.y : b.z, farz = nz ? b.z : c.y;
    const int chs[4] = {(int)c.z, (int)c.w, (int)d.x, (int)d.y};
    float ts[4];
    int m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ppg_v2f px = {(float)((nearx >> (8 * k)) & 255u), (float)((farx >> (8 * k)) & 255u)};
        ppg_v2f py = {(float)((neary >> (8 * k)) & 255u), (float)((fary >> (8 * k)) & 255u)};
        ppg_v2f pz = {(float)((nearz >> (8 * k)) & 255u), (float)((farz >> (8 * k)) & 255u)};
        px = (px - osx) * idx; py = (py - osy) * idy; pz = (pz - osz) * idz;
        const float n = fmaxf(fmaxf(px.x, py.x), fmaxf(pz.x, mint));
        // exit distance widened by 1 + 2 gamma_3: the test stays conservative whatever the box padding
        const float f = fminf(fminf(fminf(px.y, py.y), pz.y) * 1.0000008f, tlim);
        const bool hit = (n <= f) && chs[k] != PPG_BVH4_EMPTY;
        ts[k] = hit ? n : __builtin_inff();
        m += hit ? 1 : 0;
    }
    float t0 = ts[0], t1 = ts[1], t2 = ts[2], t3 = ts[3];
    int c0 = chs[0], c1 = chs[1], c2 = chs[2], c3 = chs[3];
    PPG_CSWAP(t0, c0, t1, c1) PPG_CSWAP(t2, c2, t3, c3) PPG_CSWAP(t0, c0, t2, c2) PPG_CSWAP(t1, c1, t3, c3) PPG_CSWAP(t1, c1, t2, c2)
    Bvh4Hits r; r.c0 = c0; r.c1 = c1; r.c2 = c2; r.c3 = c3; r.m = m;
    return r;
}

// (ray, triangle)-PAIR compaction of a wave's leaf phase (k_trace).  When the leaf vote passes, some 16 – 30 lanes of the wave hold a leaf
// of 1 – 8 triangles; tested lane by lane the wave runs the triangle test max(cnt) times for a quarter of its lanes.  Here the wave lists
// its (lane, k-th triangle of that lane's leaf) pairs, deals them to ALL its lanes — a lane fetches the owner's ray with ds_bpermute —
// and runs the test once per 64 pairs.  A pair's result goes to its owner through an LDS minimum on the 64-bit key (bits of t, original
// index): the order of tri_hit's callers' update rule  t < best.t || (t == best.t && orig < bestOrig)  for t >= 0, so the closest hit is
// the same bit for bit (every pair is tested against the owner's bound at the time of the vote instead of a bound shrinking inside the
// leaf: that only admits candidates which lose the minimum).  Wave-synchronous: must be called by all 64 lanes, converged.
struct PairLds {
    unsigned long long key[64];  // per owner lane: the best (t, orig) so far
    float2 uv[64];               // ... and its barycentrics / leaf-order triangle index, written by the lane that holds the minimum
    int prim[64];
    unsigned char map[512];      // pair -> owner lane
};
// exclusive prefix sum and total of cnt (0..8) over the wave: four ballots
D void pair_prefix(int cnt, unsigned int &off, unsigned int &total) {
    off = 0; total = 0;
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
        const unsigned long long m = __ballot((cnt >> bit) & 1);
        off += __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u)) << bit;
        total += (unsigned int)__popcll(m) << bit;
    }
}
// cnt = this lane's triangles (0 = it holds no leaf), off / total = pair_prefix(cnt)
D void leaf_pairs(PairLds *W, const float4 *accel, int first, int cnt, unsigned int off, unsigned int total, F3 o, F3 d, float mint, float tmax, Hit &best,
                  int &bestOrig) {
    const int lane = threadIdx.x & 63;
    const bool isLeaf = cnt > 0;
    const unsigned long long own = ((unsigned long long)__float_as_uint(best.t) << 32) | (unsigned int)bestOrig;
    for (int k = 0; k < cnt; ++k) W->map[off + k] = (unsigned char)lane;
    if (isLeaf) W->key[lane] = own;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (unsigned int base = 0; base < total; base += 64) {  // (uniform)
        const unsigned int p = base + lane;
        const bool mine = p < total;
        const int src = mine ? (int)W->map[p] : lane;
        const int k = (int)(p - __shfl(off, src));
        const F3 os = f3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
        const F3 dsv = f3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
        const float mint_s = __shfl(mint, src), tmax_s = __shfl(tmax, src);
        const int q = __shfl(first, src) + k;
        float tt = 0, uu = 0, vv = 0;
        int orig = 0;
        const bool hit = mine && tri_hit(accel + 3 * q, os, dsv, mint_s, tmax_s, tt, uu, vv, orig);
        const unsigned long long key = ((unsigned long long)__float_as_uint(tt) << 32) | (unsigned int)orig;
        if (hit) atomicMin(&W->key[src], key);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (hit && W->key[src] == key) { W->uv[src] = make_float2(uu, vv); W->prim[src] = q; }  // (keys are unique: a triangle is in one leaf)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (isLeaf) {
        const unsigned long long kk = W->key[lane];
        if (kk != own) {
            const float2 uv = W->uv[lane];
            best.t = __uint_as_float((unsigned int)(kk >> 32)); best.u = uv.x; best.v = uv.y; best.prim = W->prim[lane];
            bestOrig = (int)(unsigned int)kk;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Closest hit by (t, original primitive index) through the BVH4 — equals brute force (conservative culling).
// ANY: return at the first triangle hit (shadow rays; only prim >= 0 is meaningful then).
// VOTE: as in k_trace, lanes holding a leaf wait until PPG_LEAF_VOTE lanes of the wave do (or none has an interior node left) — for callers
// whose whole wave traverses at once (k_tail); lanes that are done have left the loop and do not count.
template <bool ANY = false, bool SPH = false, bool VOTE = false>
D Hit trace_closest4(const DevScene &S, int *lds_stack_col, int stride, F3 o, F3 d, float mint, float maxt) {
    Hit best;
    best.t = __builtin_inff(); best.u = 0; best.v = 0; best.prim = -1;
    int bestOrig = 0x7fffffff;
    const F3 id = f3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    TStack st;
    int st_over[PPG_STACK_OVER];
    st.over = st_over;
    st.lds = lds_stack_col; st.stride = stride; st.sp = 0; st.cap = PPG_LDS_STACK;
    int cur = 0;
    for (;;) {  // one step per iteration: an interior node or a leaf popped from the stack (see trace_slice_bvh4)
        bool doLeaves = true;
        if (VOTE) {
            const unsigned long long leafLanes = __ballot(cur < 0), nodeLanes = __ballot(cur >= 0);
            doLeaves = __popcll(leafLanes) >= PPG_LEAF_VOTE_TAIL || nodeLanes == 0ull;
        }
        if (cur >= 0) {
            const Bvh4Hits hc = bvh4_children(S.bvh4 + cur, o, id, mint, fminf(maxt, best.t));
            if (hc.m > 0) {
                st.push_children(hc.m, hc.c1, hc.c2, hc.c3);
                cur = hc.c0;
            } else {
                if (st.sp == 0) break;
                cur = st.pop();
            }
        } else if (doLeaves) {
            const int code = ~cur;
            const int first = code >> 3, cnt = (code & 7) + 1;
            for (int q = first; q < first + cnt; ++q) {
                float tt, uu, vv;
                const float4 *T = S.accel + 3 * q;
                int orig;
                if (tri_hit(T, o, d, mint, fminf(maxt, best.t), tt, uu, vv, orig)) {
                    if (ANY) { best.t = tt; best.prim = q; return best; }
                    if (tt < best.t || (tt == best.t && orig < bestOrig)) { best.t = tt; best.u = uu; best.v = vv; best.prim = q; bestOrig = orig; }
                }
            }
            if (st.sp == 0) break;
            cur = st.pop();
        }
    }
    if (SPH && S.n_spheres) sphere_pass<ANY>(S, o, d, mint, maxt, best);
    return best;
}

#ifndef PPG_PAIR_VOTE
#define PPG_PAIR_VOTE 32       // k_trace: the leaf phase runs when the wave's leaves hold this many triangles (or no lane has an interior node)
#endif
// trace_closest4<false, SPH, VOTE> that can be SUSPENDED (k_tail's crowd phase).  A wave's lanes trace one ray each and the lengths of
// the traversals differ by an order of magnitude: without this the wave waits for its longest ray with most lanes idle.  When no more than
// `suspend_lanes` lanes are still traversing (of more than twice as many that began), they park their state — node / leaf at hand, stack
// pointer, original index of the best hit in `park`, the best hit in `best`, the stack where it is, in the lane's LDS column — and return
// false; the caller shades the other lanes' hits and calls again with resume = true together with those lanes' next rays: a long ray spans
// several rounds of the wave instead of holding one up.  A lane whose stack has overflowed into its private array is waited for (rare).
// The closest hit is the same bit for bit: the traversal is the same, only interleaved differently with other lanes' work.
#ifndef PPG_TAIL_SUSPEND
#define PPG_TAIL_SUSPEND 8
#endif
template <bool SPH>
D bool trace_closest4_resume(const DevScene &S, int *lds_stack_col, int stride, F3 o, F3 d, float mint, float maxt, bool resume, int *park, int pstride,
                             Hit &best, int suspend_lanes) {
    int bestOrig = 0x7fffffff, cur = 0;
    const F3 id = f3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    TStack st;
    int st_over[PPG_STACK_OVER];
    st.over = st_over;
    st.lds = lds_stack_col; st.stride = stride; st.sp = 0; st.cap = PPG_LDS_STACK;
    if (resume) { cur = park[0]; st.sp = park[pstride]; bestOrig = park[2 * pstride]; }
    else { best.t = __builtin_inff(); best.u = 0; best.v = 0; best.prim = -1; }
    const int n0 = (int)__popcll(__ballot(true));
    for (;;) {
        const unsigned long long leafLanes = __ballot(cur < 0), nodeLanes = __ballot(cur >= 0);
        if (suspend_lanes && n0 > 2 * suspend_lanes && (int)__popcll(leafLanes | nodeLanes) <= suspend_lanes && __ballot(st.sp > PPG_LDS_STACK) == 0ull) {
            park[0] = cur; park[pstride] = st.sp; park[2 * pstride] = bestOrig;
            return false;
        }
        const bool doLeaves = __popcll(leafLanes) >= PPG_LEAF_VOTE_TAIL || nodeLanes == 0ull;
        if (cur >= 0) {
            const Bvh4Hits hc = bvh4_children(S.bvh4 + cur, o, id, mint, fminf(maxt, best.t));
            if (hc.m > 0) {
                st.push_children(hc.m, hc.c1, hc.c2, hc.c3);
                cur = hc.c0;
            } else {
                if (st.sp == 0) break;
                cur = st.pop();
            }
        } else if (doLeaves) {
            const int code = ~cur;
            const int first = code >> 3, cnt = (code & 7) + 1;
            for (int q = first; q < first + cnt; ++q) {
                float tt, uu, vv;
                int orig;
                if (tri_hit(S.accel + 3 * q, o, d, mint, fminf(maxt, best.t), tt, uu, vv, orig)) {
                    if (tt < best.t || (tt == best.t && orig < bestOrig)) { best.t = tt; best.u = uu; best.v = vv; best.prim = q; bestOrig = orig; }
                }
            }
            if (st.sp == 0) break;
            cur = st.pop();
        }
    }
    if (SPH && S.n_spheres) sphere_pass<false>(S, o, d, mint, maxt, best);
    return true;
}

// ONE ray traversed by a whole WAVE (k_tail, when a wave carries only a handful of live paths).  A lone lane's traversal is a chain of
// ~17 dependent node / leaf fetches, almost every one an L2 miss (the BVH and the triangle records are 70 MB against 4 MB of L2 per XCD):
// measured 36 k cycles of a lone path's 54 k-cycle bounce.  Here the wave works on up to 16 stack entries at once — four lanes
// per entry, one child box or one triangle each — so the dependent chain is as long as the tree is deep, not as the traversal is long,
// and all fetches of a level are in flight together.  Entries are node indices / leaf codes on a wave-wide stack laid over the wave's 64
// LDS stack columns (entry k = row k / 64, column k % 64).  No front-to-back order is kept: more nodes are visited than by the ordered
// per-lane walk, which costs nothing in a wave that would otherwise idle.  Same box arithmetic, same triangle test, same (t, original
// index) minimum as trace_closest4, so the hit is the same bit for bit.  Must be called by all 64 lanes with identical arguments.
#ifndef PPG_COOP_MAX
#define PPG_COOP_MAX 16  // live lanes of a wave up to which k_tail traces cooperatively (measured per-iteration cycles, §7: 1 lane 36 k per-lane vs ~14 k).
                         // 6 until round 6; with the suspended traversals (which need more than 2 x PPG_TAIL_SUSPEND lanes) 10 and 16 measured +0.7 / +0.9 %
#endif
D Hit trace_closest4_wave(const DevScene &S, int *wave_stack /* this wave's column 0 */, int stride, F3 o, F3 d, float mint, float maxt,
                          int *probe_steps = nullptr) {
    const int lane = threadIdx.x & 63, g = lane >> 2, c = lane & 3;
    Hit best;
    best.t = __builtin_inff(); best.u = 0; best.v = 0; best.prim = -1;
    int bestOrig = 0x7fffffff;
    const F3 id = f3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    auto slot = [&](int k) -> int & { return wave_stack[(k >> 6) * stride + (k & 63)]; };
    int count = 1;
    if (S.n_top > 0) {
        // first step: the tree's top cut — up to 64 subtrees that partition the scene, each with the box its parent's node would decode for
        // it — tested by one lane each, instead of the first three levels one dependent step after the other
        int push = PPG_BVH4_EMPTY;
        if (lane < S.n_top) {
            float4 lo = S.bvh_top[2 * lane], hi = S.bvh_top[2 * lane + 1];
            PPG_PIN4(lo); PPG_PIN4(hi);
            const float ax = (lo.x - o.x) * id.x, bx = (hi.x - o.x) * id.x;
            const float ay = (lo.y - o.y) * id.y, by = (hi.y - o.y) * id.y;
            const float az = (lo.z - o.z) * id.z, bz = (hi.z - o.z) * id.z;
            const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), mint));
            const float f = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000008f, maxt);
            if (n <= f) push = __float_as_int(lo.w);
        }
        const unsigned long long pm = __ballot(push != PPG_BVH4_EMPTY);
        if (push != PPG_BVH4_EMPTY) slot((int)__popcll(pm & ((1ull << lane) - 1ull))) = push;
        count = (int)__popcll(pm);
    } else if (lane == 0) slot(0) = 0;
    for (;;) {
        if (count == 0) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the entries other lanes pushed in the previous step (LDS, wave-synchronous)
        const int take = count < 16 ? count : 16, base = count - take;
        if (probe_steps) ++*probe_steps;
        const int entry = g < take ? slot(base + g) : PPG_BVH4_EMPTY;
        count = base;
        const float tlim = fminf(maxt, best.t);
        int push = PPG_BVH4_EMPTY;
        float tt = __builtin_inff(), uu = 0, vv = 0;
        int tq = -1, torig = 0x7fffffff;
        // A step is ONE memory round trip for the whole wave: every lane issues all the loads of its entry — the node's four 16-byte words,
        // or its triangle's three — back to back and waits once (PPG_PIN).  Left to itself the compiler sinks each load to its first use
        // behind the early-outs of the tests, which made a triangle four dependent round trips (projection axis, t, barycentrics, original
        // index) and a node up to three: a lone path's traversal, a chain of such steps, spent 2100 cycles per step (cycle probe).
        if (entry >= 0 && entry != PPG_BVH4_EMPTY) {  // interior node: this lane tests child c (the arithmetic of bvh4q_load / bvh4_children)
            const uint4 *q = reinterpret_cast<const uint4 *>(S.bvh4 + entry);
            uint4 a = q[0], b = q[1], cc = q[2], dd = q[3];
            PPG_PIN4(a); PPG_PIN4(b); PPG_PIN4(cc); PPG_PIN4(dd);
            const float ox = __uint_as_float(a.x), oy = __uint_as_float(a.y), oz = __uint_as_float(a.z);
            const float sx = __uint_as_float((a.w & 255u) << 23), sy = __uint_as_float(((a.w >> 8) & 255u) << 23), sz = __uint_as_float(((a.w >> 16) & 255u) << 23);
            const int sh = 8 * c;
            const float lx = ox + (float)((b.x >> sh) & 255u) * sx, ly = oy + (float)((b.y >> sh) & 255u) * sy, lz = oz + (float)((b.z >> sh) & 255u) * sz;
            const float hx = ox + (float)((b.w >> sh) & 255u) * sx, hy = oy + (float)((cc.x >> sh) & 255u) * sy, hz = oz + (float)((cc.y >> sh) & 255u) * sz;
            const int ch = c == 0 ? (int)cc.z : (c == 1 ? (int)cc.w : (c == 2 ? (int)dd.x : (int)dd.y));
            const float ax = (lx - o.x) * id.x, bx = (hx - o.x) * id.x;
            const float ay = (ly - o.y) * id.y, by = (hy - o.y) * id.y;
            const float az = (lz - o.z) * id.z, bz = (hz - o.z) * id.z;
            const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), mint));
            const float f = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000008f, tlim);
            if (n <= f && ch != PPG_BVH4_EMPTY) push = ch;
        } else if (entry != PPG_BVH4_EMPTY) {  // leaf: this lane tests triangle c (and c + 4 of a leaf of more than four)
            const int code = ~entry, first = code >> 3, cnt = (code & 7) + 1;
            for (int k = c; k < cnt; k += 4) {
                const float4 *T = S.accel + 3 * (first + k);
                float t1, u1, v1;
                int orig;
                if (tri_hit(T, o, d, mint, tlim, t1, u1, v1, orig)) {
                    if (t1 < tt || (t1 == tt && orig < torig)) { tt = t1; uu = u1; vv = v1; tq = first + k; torig = orig; }
                }
            }
        }
        // children that were hit go onto the stack (any order).  The stack is PPG_LDS_STACK rows of 64: a traversal that would overrun it
        // (never observed; heavily overlapping boxes could) is abandoned — prim = -2 — and the caller walks that ray with one lane instead.
        const unsigned long long pm = __ballot(push != PPG_BVH4_EMPTY);
        if (count + (int)__popcll(pm) > PPG_LDS_STACK * 64) { best.prim = -2; return best; }
        if (push != PPG_BVH4_EMPTY) slot(count + (int)__popcll(pm & ((1ull << lane) - 1ull))) = push;
        count += (int)__popcll(pm);
        // the triangle hits of this step, folded into the closest hit so far by (t, original index): a step rarely has more than one, so
        // the lanes that hit are read one after the other (v_readlane, no LDS round trips) instead of two 64-lane shuffle reductions
        for (unsigned long long hm = __ballot(tq >= 0); hm; hm &= hm - 1ull) {
            const int src = __ffsll((long long)hm) - 1;
            const float ts = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt), src));
            const int os = __builtin_amdgcn_readlane(torig, src);
            if (ts < best.t || (ts == best.t && os < bestOrig)) {
                best.t = ts; bestOrig = os;
                best.u = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uu), src));
                best.v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), src));
                best.prim = __builtin_amdgcn_readlane(tq, src);
            }
        }
    }
    if (S.n_spheres) sphere_pass<false>(S, o, d, mint, maxt, best);
    return best;
}

// Intersection record: fillIntersectionRecord (skdtree.h:343-430) + computeShadingFrame (util.cpp:603-608)
struct Isect {
    F3 p, geoN, s, t, n, wi;
    int material, emitter;
};
D F3 to_local(const Isect &I, F3 v) { return f3(dot3(v, I.s), dot3(v, I.t), dot3(v, I.n)); }
D F3 to_world(const Isect &I, F3 v) { return I.s * v.x + I.t * v.y + I.n * v.z; }

// Sphere::fillIntersectionRecord (sphere.cpp:213-263): position re-projected onto the sphere, its normal, dpdu of the (theta, phi)
// parameterisation as the tangent; worldToObject of a vector taken as the transposed rotation.  `o` = the ray origin.
D void fill_isect_sphere(const DevScene &S, const Hit &h, F3 o, F3 d, Isect &I) {
    const float4 *Q = S.spheres + 4 * (h.prim - S.n_tris);
    const float4 c4 = Q[0], r0 = Q[1], r1 = Q[2], r2 = Q[3];
    const F3 c = f3(c4.x, c4.y, c4.z);
    F3 p = o + d * h.t;
    p = c + norm3(p - c) * c4.w;
    const F3 v = p - c;
    const F3 local = f3(r0.x * v.x + r1.x * v.y + r2.x * v.z, r0.y * v.x + r1.y * v.y + r2.y * v.z, r0.z * v.x + r1.z * v.y + r2.z * v.z);
    const F3 du = f3(-local.y, local.x, 0.0f) * (2 * PPG_PI_F);
    const F3 dpdu = f3(r0.x * du.x + r0.y * du.y + r0.z * du.z, r1.x * du.x + r1.y * du.y + r1.z * du.z, r2.x * du.x + r2.y * du.y + r2.z * du.z);
    F3 n = norm3(p - c);
    if (__float_as_int(r2.w)) n = n * -1.0f;
    I.p = p; I.geoN = n; I.n = n;
    I.s = norm3(dpdu - n * dot3(n, dpdu));
    I.t = cross3(n, I.s);
    I.wi = to_local(I, -d);
    I.material = __float_as_int(r0.w);
    I.emitter = __float_as_int(r1.w);
}
D void fill_isect(const DevScene &S, const Hit &h, F3 d, Isect &I) {
    const float4 *T = S.tris + 3 * h.prim;
    float4 q0 = T[0], q1 = T[1];
    F3 p0 = f3(q0.x, q0.y, q0.z), p1 = f3(q1.x, q1.y, q1.z), p2 = ld3(T + 2);
    F3 b = f3(1 - h.u - h.v, h.u, h.v);
    I.p = p0 * b.x + p1 * b.y + p2 * b.z;
    F3 side1 = p1 - p0, side2 = p2 - p0;
    F3 fn = cross3(side1, side2);
    float len = len3(fn);
    if (!(fn.x == 0 && fn.y == 0 && fn.z == 0)) fn = div3(fn, len);
    F3 shN;
    if (S.normals) {
        const float4 *Nn = S.normals + 3 * h.prim;
        shN = norm3(ld3(Nn) * b.x + ld3(Nn + 1) * b.y + ld3(Nn + 2) * b.z);
        if (dot3(fn, shN) < 0) fn = -fn;
    } else {
        shN = fn;
    }
    I.geoN = fn;
    I.n = shN;
    I.s = norm3(side1 - shN * dot3(shN, side1));
    I.t = cross3(shN, I.s);
    I.wi = to_local(I, -d);
    I.material = __float_as_int(q0.w);
    I.emitter = __float_as_int(q1.w);
}

// ---- bitmap textures (FULL kernels; include/ppg.h ppg_texture) ----
// What a textured BSDF needs of the intersection besides the frame: its.uv, its.dpdu / dpdv, the material's texture word.
struct TexInfo {
    unsigned int tex;
    float u, v;
    F3 dpdu, dpdv;
};
// fillIntersectionRecord for a triangle whose BSDF may read a bitmap: as fill_isect, plus the texture coordinates (skdtree.h:403-410)
// and — on meshes that carry them — TriMesh::computeUVTangents' tangents (trimesh.cpp:683-735) as dpdu / dpdv (skdtree.h:374-381),
// which then also span the shading frame.
D void fill_isect_tex(const DevScene &S, const Hit &h, F3 d, Isect &I, TexInfo &X) {
    const float4 *T = S.tris + 3 * h.prim;
    float4 q0 = T[0], q1 = T[1];
    F3 p0 = f3(q0.x, q0.y, q0.z), p1 = f3(q1.x, q1.y, q1.z), p2 = ld3(T + 2);
    F3 b = f3(1 - h.u - h.v, h.u, h.v);
    I.p = p0 * b.x + p1 * b.y + p2 * b.z;
    F3 side1 = p1 - p0, side2 = p2 - p0;
    F3 fn = cross3(side1, side2);
    float len = len3(fn);
    const F3 nraw = fn;
    if (!(fn.x == 0 && fn.y == 0 && fn.z == 0)) fn = div3(fn, len);
    F3 shN;
    if (S.normals) {
        const float4 *Nn = S.normals + 3 * h.prim;
        shN = norm3(ld3(Nn) * b.x + ld3(Nn + 1) * b.y + ld3(Nn + 2) * b.z);
        if (dot3(fn, shN) < 0) fn = -fn;
    } else {
        shN = fn;
    }
    I.geoN = fn;
    I.n = shN;
    I.material = __float_as_int(q0.w);
    I.emitter = __float_as_int(q1.w);
    X.tex = __float_as_uint(S.materials[PPG_MAT_STRIDE * (size_t)I.material + 5].x);
    F3 dpdu = side1;
    if (X.tex) {
        X.u = b.y; X.v = b.z;
        X.dpdu = side1; X.dpdv = side2;
        if (S.uvs) {
            const float2 t0 = S.uvs[3 * (size_t)h.prim], t1 = S.uvs[3 * (size_t)h.prim + 1], t2 = S.uvs[3 * (size_t)h.prim + 2];
            if (!(t0.x != t0.x) && !(t1.x != t1.x) && !(t2.x != t2.x)) {
                X.u = t0.x * b.x + t1.x * b.y + t2.x * b.z;
                X.v = t0.y * b.x + t1.y * b.y + t2.y * b.z;
                const float dUV1x = t1.x - t0.x, dUV1y = t1.y - t0.y, dUV2x = t2.x - t0.x, dUV2y = t2.y - t0.y;
                if (len != 0) {
                    const float determinant = dUV1x * dUV2y - dUV1y * dUV2x;
                    if (determinant == 0) {
                        const F3 a = div3(nraw, len);  // coordinateSystem(n / length, dpdu, dpdv), util.cpp:592-601
                        if (ppg_abs(a.x) > ppg_abs(a.y)) {
                            float invLen = 1.0f / __builtin_sqrtf(a.x * a.x + a.z * a.z);
                            X.dpdv = f3(a.z * invLen, 0.0f, -a.x * invLen);
                        } else {
                            float invLen = 1.0f / __builtin_sqrtf(a.y * a.y + a.z * a.z);
                            X.dpdv = f3(0.0f, a.z * invLen, -a.y * invLen);
                        }
                        X.dpdu = cross3(X.dpdv, a);
                    } else {
                        const float invDet = 1.0f / determinant;
                        X.dpdu = (side1 * dUV2y - side2 * dUV1y) * invDet;
                        X.dpdv = (side1 * -dUV2x + side2 * dUV1x) * invDet;
                    }
                }
            }
        }
        dpdu = X.dpdu;
    }
    I.s = norm3(dpdu - shN * dot3(shN, dpdu));
    I.t = cross3(shN, I.s);
    I.wi = to_local(I, -d);
}

D int tex_modulo(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }  // math.h:67-70
// MIPMap::evalTexel (mipmap.h:503-563) on level 0
D F3 tex_texel(const DevTex &t, int x, int y) {
    if (x < 0 || x >= t.w) {
        switch (t.wrap_u) {
            case PPG_WRAP_REPEAT: x = tex_modulo(x, t.w); break;
            case PPG_WRAP_CLAMP: x = x < 0 ? 0 : t.w - 1; break;
            case PPG_WRAP_MIRROR: x = tex_modulo(x, 2 * t.w); if (x >= t.w) x = 2 * t.w - x - 1; break;
            case PPG_WRAP_ZERO: return f3s(0.0f);
            default: return f3s(1.0f);
        }
    }
    if (y < 0 || y >= t.h) {
        switch (t.wrap_v) {
            case PPG_WRAP_REPEAT: y = tex_modulo(y, t.h); break;
            case PPG_WRAP_CLAMP: y = y < 0 ? 0 : t.h - 1; break;
            case PPG_WRAP_MIRROR: y = tex_modulo(y, 2 * t.h); if (y >= t.h) y = 2 * t.h - y - 1; break;
            case PPG_WRAP_ZERO: return f3s(0.0f);
            default: return f3s(1.0f);
        }
    }
    const float4 v = t.texels[(size_t)y * t.w + x];
    return f3(v.x, v.y, v.z);
}
// Texture2D::eval(its) without UV partials (texture.cpp:112-121) → BitmapTexture::eval(uv) (bitmap.cpp:431-452)
D F3 tex_eval(const DevTex &t, float iu, float iv) {
    const float ux = iu * t.su + t.ou, uy = iv * t.sv + t.ov;
    if (t.nearest) return tex_texel(t, (int)__builtin_floorf(ux * t.w), (int)__builtin_floorf(uy * t.h));
    if (!ppg_isfinite(ux) || !ppg_isfinite(uy)) return f3s(0.0f);
    const float u = ux * t.w - 0.5f, v = uy * t.h - 0.5f;
    const int xPos = (int)__builtin_floorf(u), yPos = (int)__builtin_floorf(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    return tex_texel(t, xPos, yPos) * dx2 * dy2 + tex_texel(t, xPos, yPos + 1) * dx2 * dy1 + tex_texel(t, xPos + 1, yPos) * dx1 * dy2 +
           tex_texel(t, xPos + 1, yPos + 1) * dx1 * dy1;
}
// Texture2D::evalGradient(its) (texture.cpp:123-131) over MIPMap::evalGradientBilinear(0, uv) (mipmap.h:601-626): luminance of d/du, d/dv
D void tex_gradient_lum(const DevTex &t, float iu, float iv, float &gu, float &gv) {
    const float ux = iu * t.su + t.ou, uy = iv * t.sv + t.ov;
    gu = 0.0f; gv = 0.0f;
    if (t.nearest || !ppg_isfinite(ux) || !ppg_isfinite(uy)) return;
    const float u = ux * t.w - 0.5f, v = uy * t.h - 0.5f;
    const int xPos = (int)__builtin_floorf(u), yPos = (int)__builtin_floorf(v);
    const float dx = u - xPos, dy = v - yPos;
    const F3 p00 = tex_texel(t, xPos, yPos), p10 = tex_texel(t, xPos + 1, yPos), p01 = tex_texel(t, xPos, yPos + 1), p11 = tex_texel(t, xPos + 1, yPos + 1);
    const F3 tmp = p01 + p10 - p11;
    F3 g0 = (p10 + p00 * (dy - 1) - tmp * dy) * (float)t.w;
    F3 g1 = (p01 + p00 * (dx - 1) - tmp * dx) * (float)t.h;
    g0 = g0 * t.su;
    g1 = g1 * t.sv;
    gu = g0.x * 0.212671f + g0.y * 0.715160f + g0.z * 0.072169f;
    gv = g1.x * 0.212671f + g1.y * 0.715160f + g1.z * 0.072169f;
}
// BumpMap::getFrame (bumpmap.cpp:135-160): the perturbed frame (s, t, n)
D void bump_frame(const DevScene &S, const Isect &I, const TexInfo &X, F3 &ps, F3 &pt, F3 &pn) {
    float dDispDu, dDispDv;
    tex_gradient_lum(S.textures[(X.tex >> 16) - 1], X.u, X.v, dDispDu, dDispDv);
    const F3 dpdu = X.dpdu + I.n * (dDispDu - dot3(I.n, X.dpdu));
    const F3 dpdv = X.dpdv + I.n * (dDispDv - dot3(I.n, X.dpdv));
    pn = norm3(cross3(dpdu, dpdv));
    ps = norm3(dpdu - pn * dot3(pn, dpdu));
    pt = cross3(pn, ps);
    if (dot3(pn, I.geoN) < 0) pn = pn * -1.0f;
}

// AreaLight::eval (area.cpp:104-109)
D F3 eval_Le(const DevScene &S, const Isect &I, F3 dir) {
    if (I.emitter < 0) return f3s(0.0f);
    if (dot3(I.n, dir) <= 0) return f3s(0.0f);
    float4 r = S.emitters[I.emitter];
    return f3(r.x, r.y, r.z);
}

// DiscreteDistribution::sample (pmf.h:124-136) on a normalised cdf with `entries` values (cdf[0] = 0)
D int pmf_sample(const float *cdf, int entries, float v) {
    int lo = 0, hi = entries;  // std::lower_bound: first position with cdf[pos] >= v
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] < v) lo = mid + 1; else hi = mid;
    }
    int index = lo - 1;
    if (index < 0) index = 0;
    if (index > entries - 2) index = entries - 2;
    while (index < entries - 1 && cdf[index + 1] - cdf[index] == 0) ++index;
    return index;
}

// Direct illumination sample on an area emitter — Scene::sampleAttenuatedEmitterDirect (scene.cpp:876-897) up to,
// not including, the transmittance test: emitter choice, AreaLight::sampleDirect (area.cpp:158-173),
// Shape::sampleDirect (shape.cpp:102-115), TriMesh::samplePosition (trimesh.cpp:412-423), Triangle::sample
// (triangle.cpp:24-58).  Returns radiance / pdf (solid angle, before the emitter-choice probability), or zero
// with ds.pdf = 0.
struct DirectSample {
    F3 n, d;
    float dist, pdf, em_pdf;
    bool is_env;
    F3 sd;        // direction and length of the shadow ray: Scene::evalTransmittance recomputes them from the two end points
    float sdist;  // (scene.cpp:621-623) — identical to (d, dist) for area emitters, a rounding apart for the environment emitter
};

D F3 cosine_hemisphere(float sx, float sy);  // defined below

// BSphere::rayIntersect (bsphere.h:88-95) + solveQuadratic (util.cpp:447-485) against the environment emitter's sphere
D bool bsphere_intersect(const DevScene &S, F3 o_, F3 d, float &nearHit, float &farHit) {
    F3 o = o_ - f3(S.bsphere.x, S.bsphere.y, S.bsphere.z);
    float A = dot3(d, d), B = 2 * dot3(o, d), Cq = dot3(o, o) - S.bsphere.w * S.bsphere.w;
    if (A == 0) {
        if (B != 0) { nearHit = farHit = -Cq / B; return true; }
        return false;
    }
    float discrim = B * B - 4.0f * A * Cq;
    if (discrim < 0) return false;
    float temp, sqrtDiscrim = __builtin_sqrtf(discrim);
    if (B < 0) temp = -0.5f * (B - sqrtDiscrim);
    else temp = -0.5f * (B + sqrtDiscrim);
    nearHit = temp / A;
    farHit = Cq / temp;
    if (nearHit > farHit) { float t = nearHit; nearHit = farHit; farHit = t; }
    return true;
}
// ConstantBackgroundEmitter::fillDirectSamplingRecord (constant.cpp:240-254): can a ray that left the scene be attributed to it?
D bool env_fill_direct(const DevScene &S, F3 o, F3 d) {
    float nearT, farT;
    return bsphere_intersect(S, o, d, nearT, farT) && !(nearT > 0) && !(farT < 0);
}
// ConstantBackgroundEmitter::pdfDirect (constant.cpp:216-231), solid angle; cos_refn = dot(d, refN), or < -1.5 when refN = 0
D float env_pdf_direct(float cos_refn) { return cos_refn < -1.5f ? PPG_INV_PI_F * 0.25f : PPG_INV_PI_F * ppg_max(0.0f, cos_refn); }
D float lum3(F3 s);  // defined below
// ---- EnvironmentMap (emitters/envmap.cpp), level-0 bilinear lookups (mipmap.h:503-596: u repeats, v clamps) ----
D F3 envmap_to_world(const DevScene &S, F3 v) {
    return f3(S.em_R[0] * v.x + S.em_R[1] * v.y + S.em_R[2] * v.z, S.em_R[3] * v.x + S.em_R[4] * v.y + S.em_R[5] * v.z, S.em_R[6] * v.x + S.em_R[7] * v.y + S.em_R[8] * v.z);
}
D F3 envmap_to_local(const DevScene &S, F3 v) {
    return f3(S.em_R[0] * v.x + S.em_R[3] * v.y + S.em_R[6] * v.z, S.em_R[1] * v.x + S.em_R[4] * v.y + S.em_R[7] * v.z, S.em_R[2] * v.x + S.em_R[5] * v.y + S.em_R[8] * v.z);
}
D F3 envmap_texel(const DevScene &S, int x, int y) {
    if (x < 0 || x >= S.em_w) { x %= S.em_w; if (x < 0) x += S.em_w; }
    if (y < 0 || y >= S.em_h) y = y < 0 ? 0 : S.em_h - 1;
    const float4 t = S.em_texels[(size_t)y * S.em_w + x];
    return f3(t.x, t.y, t.z);
}
D int envmap_clamp_row(const DevScene &S, int y) { return y < 0 ? 0 : (y > S.em_h - 1 ? S.em_h - 1 : y); }
D bool finitef(float v) { return (ppg_f2u(v) & 0x7f800000u) != 0x7f800000u; }
// evalEnvironment (envmap.cpp:381-407) of a ray without differentials travelling along the world direction d
D F3 envmap_eval(const DevScene &S, F3 dWorld) {
    const F3 v = envmap_to_local(S, dWorld);
    const float uvx = dm_atan2(v.x, -v.z) * (PPG_INV_PI_F * 0.5f), uvy = dm_acos(v.y) * PPG_INV_PI_F;
    if (!finitef(uvx) || !finitef(uvy)) return f3s(0.0f);
    const float u = uvx * S.em_w - 0.5f, vv = uvy * S.em_h - 0.5f;
    const int xPos = (int)__builtin_floorf(u), yPos = (int)__builtin_floorf(vv);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = vv - yPos, dy2 = 1.0f - dy1;
    const F3 value = envmap_texel(S, xPos, yPos) * dx2 * dy2 + envmap_texel(S, xPos, yPos + 1) * dx2 * dy1 + envmap_texel(S, xPos + 1, yPos) * dx1 * dy2 +
                     envmap_texel(S, xPos + 1, yPos + 1) * dx1 * dy1;
    return value * S.em_scale;
}
// the radiance a ray travelling along d sees once it has left the scene: constant or image-based environment emitter
D F3 env_radiance(const DevScene &S, F3 d) { return S.env.w == 2.0f ? envmap_eval(S, d) : f3(S.env.x, S.env.y, S.env.z); }
// sampleReuse (envmap.cpp:659-664): std::lower_bound over size + 1 floats
D int envmap_sample_reuse(const float *cdf, int size, float &sample) {
    int lo = 0, n = size + 1;
    while (n > 0) {
        const int half = n >> 1;
        if (cdf[lo + half] < sample) { lo += half + 1; n -= half + 1; }
        else n = half;
    }
    int index = lo - 1;
    index = index < 0 ? 0 : index;
    index = index > size - 1 ? size - 1 : index;
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
D float interval_to_tent(float sample) {  // warp.cpp:143-155
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; }
    else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - __builtin_sqrtf(sample));
}
// internalSampleDirection (envmap.cpp:557-595): direction in the emitter's frame
D void envmap_sample_direction(const DevScene &S, float sx, float sy, F3 &d, F3 &value, float &pdf) {
    const int row = envmap_sample_reuse(S.em_cdf_rows, S.em_h, sy);
    const int col = envmap_sample_reuse(S.em_cdf_cols + (size_t)row * (S.em_w + 1), S.em_w, sx);
    const float posX = (float)col + interval_to_tent(sx), posY = (float)row + interval_to_tent(sy);
    const int xPos = (int)__builtin_floorf(posX), yPos = (int)__builtin_floorf(posY);
    const float dx1 = posX - xPos, dx2 = 1.0f - dx1, dy1 = posY - yPos, dy2 = 1.0f - dy1;
    const F3 value1 = envmap_texel(S, xPos, yPos) * dx2 * dy2 + envmap_texel(S, xPos + 1, yPos) * dx1 * dy2;
    const F3 value2 = envmap_texel(S, xPos, yPos + 1) * dx2 * dy1 + envmap_texel(S, xPos + 1, yPos + 1) * dx1 * dy1;
    value = (value1 + value2) * S.em_scale;
    pdf = (lum3(value1) * S.em_row_weights[envmap_clamp_row(S, yPos)] + lum3(value2) * S.em_row_weights[envmap_clamp_row(S, yPos + 1)]) * S.em_norm;
    float sinPhi, cosPhi, sinTheta, cosTheta;
    dm_sincos(S.em_px * (posX + 0.5f), &sinPhi, &cosPhi);
    dm_sincos(S.em_py * (posY + 0.5f), &sinTheta, &cosTheta);
    d = f3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= ppg_max(ppg_abs(sinTheta), PPG_EPSILON);
}
// internalPdfDirection (envmap.cpp:598-633): direction in the emitter's frame
D float envmap_pdf_direction(const DevScene &S, F3 d) {
    const float uvx = dm_atan2(d.x, -d.z) * (PPG_INV_PI_F * 0.5f), uvy = dm_acos(d.y) * PPG_INV_PI_F;
    if (!finitef(uvx) || !finitef(uvy)) return 0.0f;
    const float u = uvx * S.em_w - 0.5f, v = uvy * S.em_h - 0.5f;
    const int xPos = (int)__builtin_floorf(u), yPos = (int)__builtin_floorf(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    const F3 value1 = envmap_texel(S, xPos, yPos) * dx2 * dy2 + envmap_texel(S, xPos + 1, yPos) * dx1 * dy2;
    const F3 value2 = envmap_texel(S, xPos, yPos + 1) * dx2 * dy1 + envmap_texel(S, xPos + 1, yPos + 1) * dx1 * dy1;
    const float sinTheta = __builtin_sqrtf(ppg_max(0.0f, 1 - d.y * d.y));
    return (lum3(value1) * S.em_row_weights[envmap_clamp_row(S, yPos)] + lum3(value2) * S.em_row_weights[envmap_clamp_row(S, yPos + 1)]) * S.em_norm /
           ppg_max(ppg_abs(sinTheta), PPG_EPSILON);
}
// ConstantBackgroundEmitter::sampleDirect (constant.cpp:176-214)
D F3 env_sample_direct(const DevScene &S, F3 ref, F3 refN, float sx, float sy, DirectSample &ds) {
    if (S.env.w == 2.0f) {  // EnvironmentMap::sampleDirect, envmap.cpp:510-538
        F3 dl, value;
        float pdfM;
        envmap_sample_direction(S, sx, sy, dl, value, pdfM);
        const F3 dw = envmap_to_world(S, dl);
        float nearT, farT;
        ds.pdf = 0.0f;
        if (iszero3(value) || pdfM == 0 || !bsphere_intersect(S, ref, dw, nearT, farT) || nearT >= 0 || farT <= 0) return f3s(0.0f);
        ds.pdf = pdfM;
        const F3 p = ref + dw * farT;
        ds.n = norm3(f3(S.bsphere.x, S.bsphere.y, S.bsphere.z) - p);
        ds.dist = farT;
        ds.d = dw;
        const F3 pd = p - ref;
        ds.sdist = len3(pd);
        ds.sd = div3(pd, ds.sdist);
        return div3(value, pdfM);
    }
    F3 d;
    float pdf;
    const bool hasRefN = !(refN.x == 0 && refN.y == 0 && refN.z == 0);
    if (hasRefN) {
        const F3 l = cosine_hemisphere(sx, sy);
        pdf = PPG_INV_PI_F * l.z;
        F3 sF, tF;  // Frame(dRec.refN): coordinateSystem, util.cpp:592-601
        if (ppg_abs(refN.x) > ppg_abs(refN.y)) {
            float invLen = 1.0f / __builtin_sqrtf(refN.x * refN.x + refN.z * refN.z);
            tF = f3(refN.z * invLen, 0.0f, -refN.x * invLen);
        } else {
            float invLen = 1.0f / __builtin_sqrtf(refN.y * refN.y + refN.z * refN.z);
            tF = f3(0.0f, refN.z * invLen, -refN.y * invLen);
        }
        sF = cross3(tF, refN);
        d = sF * l.x + tF * l.y + refN * l.z;
    } else {
        float z = 1.0f - 2.0f * sy;  // warp::squareToUniformSphere, warp.cpp:25-31
        float r = __builtin_sqrtf(ppg_max(0.0f, 1.0f - z * z));
        float sinPhi, cosPhi;
        dm_sincos(2.0f * PPG_PI_F * sx, &sinPhi, &cosPhi);
        d = f3(r * cosPhi, r * sinPhi, z);
        pdf = PPG_INV_PI_F * 0.25f;
    }
    float nearT, farT;
    ds.pdf = 0.0f;
    if (!bsphere_intersect(S, ref, d, nearT, farT)) return f3s(0.0f);
    if (!(nearT < 0 && farT > 0)) return f3s(0.0f);
    const F3 p = ref + d * farT;
    ds.n = norm3(f3(S.bsphere.x, S.bsphere.y, S.bsphere.z) - p);
    ds.d = d;
    ds.dist = farT;
    ds.pdf = pdf;
    const F3 pd = p - ref;
    ds.sdist = len3(pd);
    ds.sd = div3(pd, ds.sdist);
    if (hasRefN && dot3(ds.d, refN) <= 0) return f3s(0.0f);
    return div3(f3(S.env.x, S.env.y, S.env.z), pdf);
}
// Sphere::sampleDirect (sphere.cpp:291-355) and pdfDirect (:357-378), solid-angle measure
D void sphere_sample_direct(const float4 *Q, F3 ref, float sx, float sy, DirectSample &ds) {
    const float4 c4 = Q[0];
    const F3 c = f3(c4.x, c4.y, c4.z);
    const float radius = c4.w;
    const float invSurfaceArea = 1 / (4 * PPG_PI_F * radius * radius);
    const F3 refToCenter = c - ref;
    const float refDist2 = dot3(refToCenter, refToCenter);
    const float invRefDist = 1.0f / __builtin_sqrtf(refDist2);
    const float sinAlpha = radius * invRefDist;
    if (sinAlpha < 1 - PPG_EPSILON) {  // outside: the cone subtended by the sphere
        const float cosAlpha = __builtin_sqrtf(ppg_max(0.0f, 1.0f - sinAlpha * sinAlpha));
        const float cosTheta = (1 - sx) + sx * cosAlpha;  // warp::squareToUniformCone, warp.cpp:54-63
        const float sinTheta = __builtin_sqrtf(ppg_max(0.0f, 1.0f - cosTheta * cosTheta));
        float sinPhi, cosPhi;
        dm_sincos(2.0f * PPG_PI_F * sy, &sinPhi, &cosPhi);
        const F3 lv = f3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
        const F3 a = refToCenter * invRefDist;  // Frame(a): coordinateSystem, util.cpp:592-601
        F3 sF, tF;
        if (ppg_abs(a.x) > ppg_abs(a.y)) {
            const float invLen = 1.0f / __builtin_sqrtf(a.x * a.x + a.z * a.z);
            tF = f3(a.z * invLen, 0.0f, -a.x * invLen);
        } else {
            const float invLen = 1.0f / __builtin_sqrtf(a.y * a.y + a.z * a.z);
            tF = f3(0.0f, a.z * invLen, -a.y * invLen);
        }
        sF = cross3(tF, a);
        ds.d = sF * lv.x + tF * lv.y + a * lv.z;
        ds.pdf = (PPG_INV_PI_F * 0.5f) / (1 - cosAlpha);  // squareToUniformConePdf, warp.h:74-76
        const float projDist = dot3(refToCenter, ds.d);
        const float baseT = refDist2 / projDist;
        const F3 query = ref + ds.d * baseT;
        const F3 queryToCenter = c - query;
        const float queryDist2 = dot3(queryToCenter, queryToCenter);
        const float queryProjDist = dot3(queryToCenter, ds.d);
        const float B = -2 * queryProjDist, Cq = queryDist2 - radius * radius;  // solveQuadratic(1, B, C), util.cpp:447-485
        float nearT;
        const float discrim = B * B - 4.0f * 1.0f * Cq;
        if (discrim < 0) nearT = queryProjDist;
        else {
            const float sqrtDiscrim = __builtin_sqrtf(discrim);
            float temp;
            if (B < 0) temp = -0.5f * (B - sqrtDiscrim);
            else temp = -0.5f * (B + sqrtDiscrim);
            const float x0 = temp / 1.0f, x1 = Cq / temp;
            nearT = x0 > x1 ? x1 : x0;
        }
        ds.dist = baseT + nearT;
        ds.n = norm3(ds.d * nearT - queryToCenter);
        const F3 pd = (c + ds.n * radius) - ref;  // the shadow ray goes to dRec.p (scene.cpp:889-893), not along dRec.d
        ds.sdist = len3(pd);
        ds.sd = div3(pd, ds.sdist);
    } else {  // inside: uniform over the sphere (warp::squareToUniformSphere, warp.cpp:25-31)
        const float z = 1.0f - 2.0f * sy;
        const float r = __builtin_sqrtf(ppg_max(0.0f, 1.0f - z * z));
        float sinPhi, cosPhi;
        dm_sincos(2.0f * PPG_PI_F * sx, &sinPhi, &cosPhi);
        const F3 dv = f3(r * cosPhi, r * sinPhi, z);
        const F3 p = c + dv * radius;
        ds.n = dv;
        F3 dd = p - ref;
        const float dist2 = dot3(dd, dd);
        ds.dist = __builtin_sqrtf(dist2);
        ds.d = div3(dd, ds.dist);
        ds.pdf = invSurfaceArea * dist2 / ppg_abs(dot3(ds.d, ds.n));
        ds.sd = ds.d; ds.sdist = ds.dist;
    }
    if (__float_as_int(Q[3].w)) ds.n = ds.n * -1.0f;
}
D float sphere_pdf_direct(const float4 *Q, F3 ref, F3 d, F3 n, float dist) {
    const float4 c4 = Q[0];
    const F3 refToCenter = f3(c4.x, c4.y, c4.z) - ref;
    const float invRefDist = 1.0f / len3(refToCenter);
    const float sinAlpha = c4.w * invRefDist;
    if (sinAlpha < 1 - PPG_EPSILON) {
        const float cosAlpha = __builtin_sqrtf(ppg_max(0.0f, 1 - sinAlpha * sinAlpha));
        return (PPG_INV_PI_F * 0.5f) / (1 - cosAlpha);
    }
    const float invSurfaceArea = 1 / (4 * PPG_PI_F * c4.w * c4.w);
    return invSurfaceArea * dist * dist / ppg_abs(dot3(d, n));
}
D F3 emitter_sample_direct(const DevScene &S, F3 ref, F3 refN, float sx, float sy, DirectSample &ds) {
    ds.pdf = 0; ds.em_pdf = 0; ds.dist = 0; ds.n = f3s(0.0f); ds.d = f3s(0.0f); ds.is_env = false; ds.sd = f3s(0.0f); ds.sdist = 0;
    const int n_sel = S.n_emitters + (S.env.w != 0 ? 1 : 0);
    if (n_sel == 0) return f3s(0.0f);
    const int e = pmf_sample(S.em_sel_cdf, n_sel + 1, sx);
    const float c0 = S.em_sel_cdf[e], c1 = S.em_sel_cdf[e + 1];
    ds.em_pdf = c1 - c0;
    sx = (sx - c0) / (c1 - c0);  // sampleReuse, pmf.h:183-188
    if (e == S.n_emitters) {  // the environment emitter is the last one
        ds.is_env = true;
        return env_sample_direct(S, ref, refN, sx, sy, ds);
    }
    const int4 info = S.em_info[e];
    if (info.y == 0) return f3s(0.0f);
    if (info.y < 0) {  // the emitter is an analytic sphere: AreaLight::sampleDirect (area.cpp:158-173) on Sphere::sampleDirect
        sphere_sample_direct(S.spheres + 4 * (-info.y - 1), ref, sx, sy, ds);
        if (!(dot3(ds.d, refN) >= 0 && dot3(ds.d, ds.n) < 0 && ds.pdf != 0)) {
            ds.pdf = 0.0f;
            return f3s(0.0f);
        }
        const float4 r = S.emitters[e];
        return div3(f3(r.x, r.y, r.z), ds.pdf);
    }
    const float *acdf = S.em_area_cdf + info.z;
    const int ti = pmf_sample(acdf, info.y + 1, sy);
    const float a0 = acdf[ti], a1 = acdf[ti + 1];
    sy = (sy - a0) / (a1 - a0);
    const float4 *T = S.em_tris + 3 * (size_t)(info.x + ti);
    const F3 p0 = ld3(T), p1 = ld3(T + 1), p2 = ld3(T + 2);
    const float a = __builtin_sqrtf(ppg_max(0.0f, 1.0f - sx));  // warp::squareToUniformTriangle, warp.cpp:76-79
    const float bx = 1 - a, by = a * sy;
    const F3 sideA = p1 - p0, sideB = p2 - p0;
    const F3 p = p0 + sideA * bx + sideB * by;
    if (S.em_normals) {
        const float4 *Nn = S.em_normals + 3 * (size_t)(info.x + ti);
        ds.n = norm3(ld3(Nn) * (1.0f - bx - by) + ld3(Nn + 1) * bx + ld3(Nn + 2) * by);
    } else {
        ds.n = norm3(cross3(sideA, sideB));
    }
    ds.pdf = __int_as_float(info.w);  // invSurfaceArea
    F3 d = p - ref;
    const float distSquared = dot3(d, d);
    ds.dist = __builtin_sqrtf(distSquared);
    ds.d = div3(d, ds.dist);
    ds.sd = ds.d; ds.sdist = ds.dist;
    const float dp = ppg_abs(dot3(ds.d, ds.n));
    ds.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    if (!(dot3(ds.d, refN) >= 0 && dot3(ds.d, ds.n) < 0 && ds.pdf != 0)) {
        ds.pdf = 0.0f;
        return f3s(0.0f);
    }
    const float4 r = S.emitters[e];
    return div3(f3(r.x, r.y, r.z), ds.pdf);
}

// Transform::operator()(Point) (transform.h:108-125) and operator()(Vector) (:175-183)
D F3 xf_point(const float *m, F3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w == 1.0f) return f3(x, y, z);
    return div3(f3(x, y, z), w);
}
D F3 xf_vec(const float *m, F3 v) {
    return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// warp.cpp:81-102, 43-52
D void disk_concentric(float sx, float sy, float &px, float &py) {
    float r1 = 2.0f * sx - 1.0f;
    float r2 = 2.0f * sy - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = (PPG_PI_F / 4.0f) * (r2 / r1);
    } else {
        r = r2;
        phi = (PPG_PI_F / 2.0f) - (r1 / r2) * (PPG_PI_F / 4.0f);
    }
    float c, s;
    dm_sincos(phi, &s, &c);
    px = r * c; py = r * s;
}
D F3 cosine_hemisphere(float sx, float sy) {
    float px, py;
    disk_concentric(sx, sy, px, py);
    float z = __builtin_sqrtf(ppg_max(0.0f, 1.0f - px * px - py * py));
    if (z == 0) z = 1e-10f;
    return f3(px, py, z);
}

// SmoothDiffuse (diffuse.cpp:110-150)
D F3 diffuse_eval(F3 refl, F3 wi, F3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return f3s(0.0f);
    return refl * (PPG_INV_PI_F * wo.z);
}
D float diffuse_pdf(F3 wi, F3 wo) {
    if (wi.z <= 0 || wo.z <= 0) return 0.0f;
    return PPG_INV_PI_F * wo.z;
}

// BSDF dispatch over the supported plugins (include/ppg.h PPG_BSDF_*): diffuse.cpp:110-150, twosided.cpp:100-180
// (one diffuse BRDF on both sides), conductor.cpp:220-290 with material "none" (ideal mirror, delta)
D bool bsdf_is_smooth(int type) { return type != PPG_BSDF_MIRROR; }
D F3 bsdf_eval(int type, F3 refl, F3 wi, F3 wo) {
    if (type == PPG_BSDF_DIFFUSE) return diffuse_eval(refl, wi, wo);
    if (type == PPG_BSDF_TWOSIDED_DIFFUSE) {
        if (wi.z > 0) return diffuse_eval(refl, wi, wo);
        wi.z *= -1; wo.z *= -1;
        return diffuse_eval(refl, wi, wo);
    }
    return f3s(0.0f);
}
D float bsdf_pdf(int type, F3 wi, F3 wo) {
    if (type == PPG_BSDF_DIFFUSE) return diffuse_pdf(wi, wo);
    if (type == PPG_BSDF_TWOSIDED_DIFFUSE) {
        if (wi.z > 0) return diffuse_pdf(wi, wo);
        wi.z *= -1; wo.z *= -1;
        return diffuse_pdf(wi, wo);
    }
    return 0.0f;
}
// returns the sampling weight f·cos/pdf; wo, pdf, delta are outputs
D F3 bsdf_sample(int type, F3 refl, F3 wi, float sx, float sy, F3 &wo, float &pdf, bool &delta) {
    delta = false;
    wo = f3s(0.0f);
    if (type == PPG_BSDF_MIRROR) {
        if (wi.z <= 0) { pdf = 0.0f; return f3s(0.0f); }
        wo = f3(-wi.x, -wi.y, wi.z);  // reflect(wi)
        pdf = 1;
        delta = true;
        return refl;
    }
    bool flipped = false;
    if (type == PPG_BSDF_TWOSIDED_DIFFUSE && wi.z < 0) { wi.z *= -1; flipped = true; }
    if (wi.z <= 0) { pdf = 0.0f; return f3s(0.0f); }
    wo = cosine_hemisphere(sx, sy);
    pdf = PPG_INV_PI_F * wo.z;
    if (flipped && !iszero3(refl) && pdf != 0) wo.z *= -1;
    return refl;
}

// ------------------------------------------------------------------------------------------------
// Full material set (kernel variants instantiated with FULL = true; scenes with only diffuse / two-sided diffuse / mirror
// materials keep the lean functions above).  Expression order follows the cited reference code, like the oracle's.
// ------------------------------------------------------------------------------------------------
struct Mat {
    int type, flags;          // type normalised: TWOSIDED_DIFFUSE → DIFFUSE + PPG_MAT_TWOSIDED
    F3 refl, spec, eta, k, opacity;
    float alpha, fdr_int;
    const float *rt;          // roughplastic: its rough-transmittance slice, rt_n samples
    int rt_n;
    float refl_lum;           // luminance of the material record's reflectance = of the texture's average when `refl` was read from a bitmap:
                              // what the plug-ins' configure() derives the component sampling weights from (plastic.cpp:191-204)
};
D Mat load_material(const DevScene &S, int id) {
    const float4 *m = S.materials + PPG_MAT_STRIDE * (size_t)id;
    const float4 a = m[0], b = m[1], c = m[2], d = m[3];
    const float4 e = m[4];
    Mat M;
    M.type = (int)a.w; M.flags = __float_as_int(c.w);
    if (M.type == PPG_BSDF_TWOSIDED_DIFFUSE) { M.type = PPG_BSDF_DIFFUSE; M.flags |= PPG_MAT_TWOSIDED; }
    M.refl = f3(a.x, a.y, a.z); M.spec = f3(b.x, b.y, b.z); M.eta = f3(c.x, c.y, c.z); M.k = f3(d.x, d.y, d.z);
    M.alpha = b.w; M.fdr_int = d.w;
    M.opacity = f3(e.x, e.y, e.z);
    M.rt_n = S.rtrans_n;
    M.rt = S.rtrans + (size_t)__float_as_int(e.w) * (size_t)(S.rtrans_n + 1);
    M.refl_lum = a.x * 0.212671f + a.y * 0.715160f + a.z * 0.072169f;
    return M;
}
D bool mat_is_smooth(const Mat &M) {
    return M.type == PPG_BSDF_DIFFUSE || M.type == PPG_BSDF_ROUGHCONDUCTOR || M.type == PPG_BSDF_PLASTIC || M.type == PPG_BSDF_ROUGHDIELECTRIC ||
           M.type == PPG_BSDF_ROUGHPLASTIC;
}
D bool mat_two_sided(const Mat &M) {
    return (M.flags & PPG_MAT_TWOSIDED) && M.type != PPG_BSDF_DIELECTRIC && M.type != PPG_BSDF_THINDIELECTRIC && M.type != PPG_BSDF_ROUGHDIELECTRIC;
}
D bool mat_masked(const Mat &M) { return (M.flags & PPG_MAT_MASK) != 0; }
D bool mat_backside_or_transmission(const Mat &M) {
    return (M.flags & (PPG_MAT_TWOSIDED | PPG_MAT_MASK)) || M.type == PPG_BSDF_DIELECTRIC || M.type == PPG_BSDF_THINDIELECTRIC ||
           M.type == PPG_BSDF_ROUGHDIELECTRIC;
}
D bool mat_has_null(const Mat &M) { return mat_masked(M) || M.type == PPG_BSDF_THINDIELECTRIC; }  // getType() & ENull

D F3 cdiv3(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
D F3 safe_sqrt3(F3 s) { return f3(__builtin_sqrtf(ppg_max(0.0f, s.x)), __builtin_sqrtf(ppg_max(0.0f, s.y)), __builtin_sqrtf(ppg_max(0.0f, s.z))); }
D float lum3(F3 s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }
D float fsignum(float v) { return (ppg_f2u(v) >> 31) ? -1.0f : 1.0f; }  // math::signum: the FP sign, never zero

// util.cpp:651-681
D float fresnel_dielectric_ext(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = ppg_abs(cosThetaI_);
    float cosThetaT = __builtin_sqrtf(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
D float fresnel_dielectric_ext(float cosThetaI, float eta) { float t; return fresnel_dielectric_ext(cosThetaI, t, eta); }

// util.cpp:739-761
D F3 fresnel_conductor_exact(float cosThetaI, F3 eta, F3 k) {
    float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    F3 temp1 = mul3(eta, eta) - mul3(k, k) - f3s(sinThetaI2);
    F3 a2pb2 = safe_sqrt3(mul3(temp1, temp1) + mul3(mul3(mul3(k, k), eta), eta) * 4.0f);
    F3 a = safe_sqrt3((a2pb2 + temp1) * 0.5f);
    F3 term1 = a2pb2 + f3s(cosThetaI2), term2 = a * (2 * cosThetaI);
    F3 Rs2 = cdiv3(term1 - term2, term1 + term2);
    F3 term3 = a2pb2 * cosThetaI2 + f3s(sinThetaI4), term4 = term2 * sinThetaI2;
    F3 Rp2 = cdiv3(mul3(Rs2, term3 - term4), term3 + term4);
    return (Rp2 + Rs2) * 0.5f;
}

// math.cpp:25-72
D float mts_erfinv(float x) {
    float w = -dm_log((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = __builtin_sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}
D float mts_erf(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    const float sign = (ppg_f2u(x) >> 31) ? -1.0f : 1.0f;
    x = ppg_abs(x);
    float t = 1.0f / (1.0f + p * x);
    float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * dm_exp(-x * x);
    return sign * y;
}

// MicrofacetDistribution, isotropic GGX or Beckmann, visible-normal sampling (microfacet.h:191-237, 425-540, 565-690)
struct Mfd { float alpha; bool beckmann; };
D float ggx_eval(Mfd d, F3 m) {
    const float alpha = d.alpha;
    if (m.z <= 0) return 0.0f;
    float cosTheta2 = m.z * m.z;
    float beckmannExponent = ((m.x * m.x) / (alpha * alpha) + (m.y * m.y) / (alpha * alpha)) / cosTheta2;
    float result;
    if (d.beckmann) {
        result = dm_exp(-beckmannExponent) / (PPG_PI_F * alpha * alpha * cosTheta2 * cosTheta2);
    } else {
        float root = (1.0f + beckmannExponent) * cosTheta2;
        result = 1.0f / (PPG_PI_F * alpha * alpha * root * root);
    }
    if (result * m.z < 1e-20f) result = 0;
    return result;
}
D float hypot2f(float a, float b) {  // math.cpp:74-86
    float r;
    if (ppg_abs(a) > ppg_abs(b)) { r = b / a; r = ppg_abs(a) * __builtin_sqrtf(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = ppg_abs(b) * __builtin_sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return r;
}
D float ggx_smith_g1(Mfd d, F3 v, F3 m) {
    if (dot3(v, m) * v.z <= 0) return 0.0f;
    float temp = 1 - v.z * v.z;
    float tanTheta = temp <= 0.0f ? 0.0f : ppg_abs(__builtin_sqrtf(temp) / v.z);
    if (tanTheta == 0.0f) return 1.0f;
    if (d.beckmann) {
        float a = 1.0f / (d.alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
    }
    float root = d.alpha * tanTheta;
    return 2.0f / (1.0f + hypot2f(1.0f, root));
}
D float ggx_pdf_visible(Mfd d, F3 wi, F3 m) {
    if (wi.z == 0) return 0.0f;
    return ggx_smith_g1(d, wi, m) * ppg_abs(dot3(wi, m)) * ggx_eval(d, m) / ppg_abs(wi.z);
}
D void beckmann_sample_visible11(float thetaI, float u, float v, float &sx, float &sy) {  // microfacet.h:565-642
    const float SQRT_PI_INV = 1 / __builtin_sqrtf(PPG_PI_F);
    if (thetaI < 1e-4f) {
        float sinPhi, cosPhi;
        float r = __builtin_sqrtf(-dm_log(1.0f - u));
        dm_sincos(2 * PPG_PI_F * v, &sinPhi, &cosPhi);
        sx = r * cosPhi; sy = r * sinPhi;
        return;
    }
    float tanThetaI = dm_tan(thetaI);
    float cotThetaI = 1 / tanThetaI;
    float a = -1, c = mts_erf(cotThetaI);
    float sample_x = ppg_max(u, 1e-6f);
    float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
    float b = c - (1 + c) * dm_pow(1 - sample_x, fit);
    float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * dm_exp(-cotThetaI * cotThetaI));
    int it = 0;
    while (++it < 10) {
        if (!(b >= a && b <= c)) b = 0.5f * (a + c);
        float invErf = mts_erfinv(b);
        float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * dm_exp(-invErf * invErf)) - sample_x;
        float derivative = normalization * (1 - invErf * tanThetaI);
        if (ppg_abs(value) < 1e-5f) break;
        if (value > 0) c = b;
        else a = b;
        b -= value / derivative;
    }
    sx = mts_erfinv(b);
    sy = mts_erfinv(2.0f * ppg_max(v, 1e-6f) - 1.0f);
}
D void ggx_sample_visible11(float thetaI, float u, float v, float &sx, float &sy) {
    if (thetaI < 1e-4f) {
        float sinPhi, cosPhi;
        float r = __builtin_sqrtf(ppg_max(0.0f, u / (1 - u)));
        dm_sincos(2 * PPG_PI_F * v, &sinPhi, &cosPhi);
        sx = r * cosPhi; sy = r * sinPhi;
        return;
    }
    float tanThetaI = dm_tan(thetaI);
    float a = 1 / tanThetaI;
    float G1 = 2.0f / (1.0f + __builtin_sqrtf(ppg_max(0.0f, 1.0f + 1.0f / (a * a))));
    float A = 2.0f * u / G1 - 1.0f;
    if (ppg_abs(A) == 1) A -= (A < 0 ? -1.0f : 1.0f) * PPG_EPSILON;
    float tmp = 1.0f / (A * A - 1.0f);
    float B = tanThetaI;
    float Dq = __builtin_sqrtf(ppg_max(0.0f, B * B * tmp * tmp - (A * A - B * B) * tmp));
    float slope_x_1 = B * tmp - Dq;
    float slope_x_2 = B * tmp + Dq;
    sx = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
    float S;
    if (v > 0.5f) { S = 1.0f; v = 2.0f * (v - 0.5f); }
    else { S = -1.0f; v = 2.0f * (0.5f - v); }
    float z = (v * (v * (v * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
              (v * (v * (v * (v * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
    sy = S * z * __builtin_sqrtf(1.0f + sx * sx);
}
D F3 ggx_sample_visible(Mfd d, F3 _wi, float u, float v) {
    const float alpha = d.alpha;
    F3 wi = norm3(f3(alpha * _wi.x, alpha * _wi.y, _wi.z));
    float theta = 0, phi = 0;
    if (wi.z < 0.99999f) {
        theta = dm_acos(wi.z);
        phi = dm_atan2(wi.y, wi.x);
    }
    float sinPhi, cosPhi;
    dm_sincos(phi, &sinPhi, &cosPhi);
    float sx, sy;
    if (d.beckmann) beckmann_sample_visible11(theta, u, v, sx, sy);
    else ggx_sample_visible11(theta, u, v, sx, sy);
    float rx = cosPhi * sx - sinPhi * sy, ry = sinPhi * sx + cosPhi * sy;
    rx *= alpha; ry *= alpha;
    float normalization = 1.0f / __builtin_sqrtf(rx * rx + ry * ry + 1.0f);
    return f3(-rx * normalization, -ry * normalization, normalization);
}

// plastic.cpp:191-204 (configure) — recomputed per call from the material record, same arithmetic as the oracle's configure()
D float plastic_prob_specular(const Mat &M, float Fi) {
    float dAvg = M.refl_lum, sAvg = lum3(M.spec);
    float w = sAvg / (dAvg + sAvg);
    return (Fi * w) / (Fi * w + (1 - Fi) * (1 - w));
}
D F3 plastic_diff_term(const Mat &M) {
    if (M.flags & PPG_MAT_NONLINEAR) return cdiv3(M.refl, f3s(1.0f) - M.refl * M.fdr_int);
    return div3(M.refl, 1 - M.fdr_int);
}

// evalCubicInterp1D (spline.cpp:23-60) on [0, 1], no extrapolation
D float cubic_interp_1d(float x, const float *values, int size) {
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    float t = ((x - 0.0f) * (float)(size - 1)) / (1.0f - 0.0f);
    int k = (int)t;
    k = k < size - 2 ? k : size - 2;
    k = k > 0 ? k : 0;
    const float f0 = values[k], f1 = values[k + 1];
    float d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]);
    else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]);
    else d1 = values[k + 1] - values[k];
    t = t - (float)k;
    const float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}
// roughplastic: m_externalRoughTransmittance->eval(cosTheta, alpha), eta and alpha fixed (rtrans.h:185-196, 233)
D float rough_T(const Mat &M, float cosTheta) {
    const float warped = dm_pow(ppg_abs(cosTheta), 0.25f);
    if (!(cosTheta >= 0)) return 0.0f;
    return ppg_min(1.0f, ppg_max(0.0f, cubic_interp_1d(warped, M.rt, M.rt_n)));
}
D float roughplastic_prob_specular(const Mat &M, float cosThetaI) {  // roughplastic.cpp:406-412
    const float dAvg = M.refl_lum, sAvg = lum3(M.spec);
    const float w = sAvg / (dAvg + sAvg);
    const float pS = 1 - rough_T(M, cosThetaI);
    return (pS * w) / (pS * w + (1 - pS) * (1 - w));
}

// thindielectric.cpp:160-164: slab reflectance incl. internal reflections R' = R + TRT + TR^3T + ..
D float thin_R(const Mat &M, float cosThetaI) {
    float Rr = fresnel_dielectric_ext(ppg_abs(cosThetaI), M.eta.x), Tt = 1 - Rr;
    if (Rr < 1) Rr += Tt * Tt * Rr / (1 - Rr * Rr);
    return Rr;
}
// BSDF::eval(BSDFSamplingRecord(its, -wo, wo), EDiscrete) restricted to the null component: what a ray going straight through keeps
D F3 mat_eval_null(const Mat &M, float cosThetaI) {
    if (mat_masked(M)) return f3s(1.0f) - M.opacity;  // mask.cpp:115-116
    return M.spec * (1 - thin_R(M, cosThetaI));
}

// one-sided plugins, solid-angle measure
D F3 mat_eval_one(const Mat &M, F3 wi, F3 wo) {
    if (M.type == PPG_BSDF_DIFFUSE) return diffuse_eval(M.refl, wi, wo);
    if (M.type == PPG_BSDF_ROUGHCONDUCTOR) {  // roughconductor.cpp:247-282
        if (wi.z <= 0 || wo.z <= 0) return f3s(0.0f);
        F3 H = norm3(wo + wi);
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        const float Dm = ggx_eval(distr, H);
        if (Dm == 0) return f3s(0.0f);
        const F3 F = mul3(fresnel_conductor_exact(dot3(wi, H), M.eta, M.k), M.refl);
        const float G = ggx_smith_g1(distr, wi, H) * ggx_smith_g1(distr, wo, H);
        float model = Dm * G / (4.0f * wi.z);
        return F * model;
    }
    if (M.type == PPG_BSDF_ROUGHDIELECTRIC) {  // roughdielectric.cpp:268-340
        if (wi.z == 0) return f3s(0.0f);
        const float m_eta = M.eta.x, m_invEta = 1 / m_eta;
        const bool reflect = wi.z * wo.z > 0;
        F3 H;
        if (reflect) H = norm3(wo + wi);
        else {
            float eta = wi.z > 0 ? m_eta : m_invEta;
            H = norm3(wi + wo * eta);
        }
        H = H * fsignum(H.z);
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        const float Dm = ggx_eval(distr, H);
        if (Dm == 0) return f3s(0.0f);
        const float F = fresnel_dielectric_ext(dot3(wi, H), m_eta);
        const float G = ggx_smith_g1(distr, wi, H) * ggx_smith_g1(distr, wo, H);
        if (reflect) {
            float value = F * Dm * G / (4.0f * ppg_abs(wi.z));
            return M.refl * value;
        }
        float eta = wi.z > 0.0f ? m_eta : m_invEta;
        float sqrtDenom = dot3(wi, H) + eta * dot3(wo, H);
        float value = ((1 - F) * Dm * G * eta * eta * dot3(wi, H) * dot3(wo, H)) / (wi.z * sqrtDenom * sqrtDenom);
        float factor = wi.z > 0 ? m_invEta : m_eta;
        return M.spec * ppg_abs(value * factor * factor);
    }
    if (M.type == PPG_BSDF_ROUGHPLASTIC) {  // roughplastic.cpp:330-384
        if (wi.z <= 0 || wo.z <= 0) return f3s(0.0f);
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        const F3 H = norm3(wo + wi);
        const float Dm = ggx_eval(distr, H);
        const float F = fresnel_dielectric_ext(dot3(wi, H), M.eta.x);
        const float G = ggx_smith_g1(distr, wi, H) * ggx_smith_g1(distr, wo, H);
        const float value = F * Dm * G / (4.0f * wi.z);
        const F3 result = M.spec * value;
        const float T12 = rough_T(M, wi.z), T21 = rough_T(M, wo.z);
        const float invEta2 = 1.0f / (M.eta.x * M.eta.x);
        return result + plastic_diff_term(M) * (PPG_INV_PI_F * wo.z * T12 * T21 * invEta2);
    }
    if (M.type == PPG_BSDF_PLASTIC) {  // plastic.cpp:247-281, diffuse component
        if (wo.z <= 0 || wi.z <= 0) return f3s(0.0f);
        float Fi = fresnel_dielectric_ext(wi.z, M.eta.x);
        float Fo = fresnel_dielectric_ext(wo.z, M.eta.x);
        const float invEta2 = 1 / (M.eta.x * M.eta.x);
        return plastic_diff_term(M) * ((PPG_INV_PI_F * wo.z) * invEta2 * (1 - Fi) * (1 - Fo));
    }
    return f3s(0.0f);
}
D float mat_pdf_one(const Mat &M, F3 wi, F3 wo) {
    if (M.type == PPG_BSDF_DIFFUSE) return diffuse_pdf(wi, wo);
    if (M.type == PPG_BSDF_ROUGHCONDUCTOR) {  // roughconductor.cpp:284-307
        if (wi.z <= 0 || wo.z <= 0) return 0.0f;
        F3 H = norm3(wo + wi);
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        return ggx_eval(distr, H) * ggx_smith_g1(distr, wi, H) / (4.0f * wi.z);
    }
    if (M.type == PPG_BSDF_ROUGHDIELECTRIC) {  // roughdielectric.cpp:342-418
        const float m_eta = M.eta.x, m_invEta = 1 / m_eta;
        const bool reflect = wi.z * wo.z > 0;
        F3 H;
        float dwh_dwo;
        if (reflect) {
            H = norm3(wo + wi);
            dwh_dwo = 1.0f / (4.0f * dot3(wo, H));
        } else {
            float eta = wi.z > 0 ? m_eta : m_invEta;
            H = norm3(wi + wo * eta);
            float sqrtDenom = dot3(wi, H) + eta * dot3(wo, H);
            dwh_dwo = (eta * eta * dot3(wo, H)) / (sqrtDenom * sqrtDenom);
        }
        H = H * fsignum(H.z);
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        float prob = ggx_pdf_visible(distr, wi * fsignum(wi.z), H);
        float F = fresnel_dielectric_ext(dot3(wi, H), m_eta);
        prob *= reflect ? F : (1 - F);
        return ppg_abs(prob * dwh_dwo);
    }
    if (M.type == PPG_BSDF_ROUGHPLASTIC) {  // roughplastic.cpp:386-437
        if (wi.z <= 0 || wo.z <= 0) return 0.0f;
        const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
        const F3 H = norm3(wo + wi);
        const float pS = roughplastic_prob_specular(M, wi.z), pD = 1 - pS;
        const float dwh_dwo = 1.0f / (4.0f * dot3(wo, H));
        const float prob = ggx_pdf_visible(distr, wi, H);
        float result = prob * dwh_dwo * pS;
        result += pD * (PPG_INV_PI_F * wo.z);
        return result;
    }
    if (M.type == PPG_BSDF_PLASTIC) {  // plastic.cpp:283-311
        if (wo.z <= 0 || wi.z <= 0) return 0.0f;
        float Fi = fresnel_dielectric_ext(wi.z, M.eta.x);
        return (PPG_INV_PI_F * wo.z) * (1 - plastic_prob_specular(M, Fi));
    }
    return 0.0f;
}
D F3 mat_sample_one(const Mat &M, F3 wi, float sx, float sy, F3 &wo, float &pdf, bool &delta, float &eta, bool &isnull, unsigned int key,
                    unsigned int &dim) {
    delta = false; eta = 1.0f; pdf = 0.0f; wo = f3s(0.0f); isnull = false;
    switch (M.type) {
        case PPG_BSDF_DIFFUSE:
            if (wi.z <= 0) return f3s(0.0f);
            wo = cosine_hemisphere(sx, sy);
            pdf = PPG_INV_PI_F * wo.z;
            return M.refl;
        case PPG_BSDF_MIRROR:
        case PPG_BSDF_CONDUCTOR:  // conductor.cpp:268-284
            if (wi.z <= 0) return f3s(0.0f);
            wo = f3(-wi.x, -wi.y, wi.z);
            delta = true;
            pdf = 1;
            return mul3(M.refl, fresnel_conductor_exact(wi.z, M.eta, M.k));
        case PPG_BSDF_ROUGHCONDUCTOR: {  // roughconductor.cpp:367-415
            if (wi.z < 0) return f3s(0.0f);
            const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
            F3 m = ggx_sample_visible(distr, wi, sx, sy);
            pdf = ggx_pdf_visible(distr, wi, m);
            if (pdf == 0) return f3s(0.0f);
            wo = m * (2 * dot3(wi, m)) - wi;
            if (wo.z <= 0) return f3s(0.0f);
            F3 F = mul3(fresnel_conductor_exact(dot3(wi, m), M.eta, M.k), M.refl);
            float weight = ggx_smith_g1(distr, wo, m);
            pdf /= 4.0f * dot3(wo, m);
            return F * weight;
        }
        case PPG_BSDF_ROUGHPLASTIC: {  // roughplastic.cpp:439-501
            if (wi.z <= 0) return f3s(0.0f);
            const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
            const float pS = roughplastic_prob_specular(M, wi.z);
            if (sy < pS) {
                sy /= pS;
                const F3 m = ggx_sample_visible(distr, wi, sx, sy);
                wo = m * (2 * dot3(wi, m)) - wi;
                if (wo.z <= 0) return f3s(0.0f);
            } else {
                sy = (sy - pS) / (1 - pS);
                wo = cosine_hemisphere(sx, sy);
            }
            pdf = mat_pdf_one(M, wi, wo);
            if (pdf == 0) return f3s(0.0f);
            return div3(mat_eval_one(M, wi, wo), pdf);
        }
        case PPG_BSDF_PLASTIC: {  // plastic.cpp:368-425
            if (wi.z <= 0) return f3s(0.0f);
            float Fi = fresnel_dielectric_ext(wi.z, M.eta.x);
            float pS = plastic_prob_specular(M, Fi);
            if (sx < pS) {
                delta = true;
                wo = f3(-wi.x, -wi.y, wi.z);
                pdf = pS;
                return div3(M.spec * Fi, pS);
            }
            wo = cosine_hemisphere((sx - pS) / (1 - pS), sy);
            float Fo = fresnel_dielectric_ext(wo.z, M.eta.x);
            pdf = (1 - pS) * (PPG_INV_PI_F * wo.z);
            const float invEta2 = 1 / (M.eta.x * M.eta.x);
            return plastic_diff_term(M) * (invEta2 * (1 - Fi) * (1 - Fo) / (1 - pS));
        }
        case PPG_BSDF_DIELECTRIC: {  // dielectric.cpp:271-312, ERadiance
            float cosThetaT;
            const float e = M.eta.x, invEta = 1 / e;
            float F = fresnel_dielectric_ext(wi.z, cosThetaT, e);
            delta = true;
            if (sx <= F) {
                wo = f3(-wi.x, -wi.y, wi.z);
                pdf = F;
                return M.refl;
            }
            float scale = -(cosThetaT < 0 ? invEta : e);
            wo = f3(scale * wi.x, scale * wi.y, cosThetaT);
            eta = cosThetaT < 0 ? e : invEta;
            pdf = 1 - F;
            float factor = cosThetaT < 0 ? invEta : e;
            return M.spec * (factor * factor);
        }
        case PPG_BSDF_ROUGHDIELECTRIC: {  // roughdielectric.cpp:514-606
            const float m_eta = M.eta.x, m_invEta = 1 / m_eta;
            const Mfd distr{M.alpha, (M.flags & PPG_MAT_BECKMANN) != 0};
            const F3 wis = wi * fsignum(wi.z);
            const F3 m = ggx_sample_visible(distr, wis, sx, sy);
            const float microfacetPDF = ggx_pdf_visible(distr, wis, m);
            if (microfacetPDF == 0) return f3s(0.0f);
            pdf = microfacetPDF;
            float cosThetaT;
            const float F = fresnel_dielectric_ext(dot3(wi, m), cosThetaT, m_eta);
            F3 weight = f3s(1.0f);
            bool sampleReflection = true;
            if (ppg_rand(key, dim++) > F) { sampleReflection = false; pdf *= 1 - F; }  // bRec.sampler->next1D(), roughdielectric.cpp:553
            else pdf *= F;
            float dwh_dwo;
            if (sampleReflection) {
                wo = m * (2 * dot3(wi, m)) - wi;
                if (wi.z * wo.z <= 0) return f3s(0.0f);
                weight = mul3(weight, M.refl);
                dwh_dwo = 1.0f / (4.0f * dot3(wo, m));
            } else {
                if (cosThetaT == 0) return f3s(0.0f);
                float e2 = m_eta;  // refract(wi, m, eta, cosThetaT), util.cpp:767-772
                if (cosThetaT < 0) e2 = 1 / e2;
                wo = m * (dot3(wi, m) * e2 + cosThetaT) - wi * e2;
                eta = cosThetaT < 0 ? m_eta : m_invEta;
                if (wi.z * wo.z >= 0) return f3s(0.0f);
                float factor = cosThetaT < 0 ? m_invEta : m_eta;
                weight = mul3(weight, M.spec * (factor * factor));
                float sqrtDenom = dot3(wi, m) + eta * dot3(wo, m);
                dwh_dwo = (eta * eta * dot3(wo, m)) / (sqrtDenom * sqrtDenom);
            }
            weight = weight * ggx_smith_g1(distr, wo, m);
            pdf *= ppg_abs(dwh_dwo);
            return weight;
        }
        case PPG_BSDF_THINDIELECTRIC: {  // thindielectric.cpp:203-232
            const float Rr = thin_R(M, wi.z);
            delta = true;
            if (sx <= Rr) {
                wo = f3(-wi.x, -wi.y, wi.z);
                pdf = Rr;
                return M.refl;
            }
            isnull = true;
            wo = -wi;  // transmit()
            pdf = 1 - Rr;
            return M.spec;
        }
    }
    return f3s(0.0f);
}
// + the TwoSided adapter (twosided.cpp:120-180) and, outside it, the Mask adapter (mask.cpp:108-214)
D F3 mat_eval(const Mat &M, F3 wi, F3 wo) {
    if (mat_two_sided(M) && !(wi.z > 0)) { wi.z *= -1; wo.z *= -1; }
    F3 r = mat_eval_one(M, wi, wo);
    return mat_masked(M) ? mul3(r, M.opacity) : r;
}
D float mat_pdf(const Mat &M, F3 wi, F3 wo) {
    if (mat_two_sided(M) && !(wi.z > 0)) { wi.z *= -1; wo.z *= -1; }
    float r = mat_pdf_one(M, wi, wo);
    return mat_masked(M) ? r * lum3(M.opacity) : r;
}
D F3 mat_sample_ts(const Mat &M, F3 wi, float sx, float sy, F3 &wo, float &pdf, bool &delta, float &eta, bool &isnull, unsigned int key,
                   unsigned int &dim);
// key / dim: the path's sampler, for plug-ins whose sample() draws from it (roughdielectric)
D F3 mat_sample(const Mat &M, F3 wi, float sx, float sy, F3 &wo, float &pdf, bool &delta, float &eta, bool &isnull, unsigned int key,
                unsigned int &dim) {
    // (ONE call of the nested BSDF's sample(): see the note on code size in shade_one)
    const bool masked = mat_masked(M);
    const float prob = masked ? lum3(M.opacity) : 1.0f;
    if (masked && !(sx < prob)) {
        wo = -wi;
        eta = 1.0f;
        delta = true;
        isnull = true;
        pdf = 1 - prob;
        return div3(f3s(1.0f) - M.opacity, pdf);
    }
    if (masked) sx /= prob;
    F3 result = mat_sample_ts(M, wi, sx, sy, wo, pdf, delta, eta, isnull, key, dim);
    if (masked) {
        result = div3(mul3(result, M.opacity), prob);
        pdf *= prob;
    }
    return result;
}
D F3 mat_sample_ts(const Mat &M, F3 wi, float sx, float sy, F3 &wo, float &pdf, bool &delta, float &eta, bool &isnull, unsigned int key,
                   unsigned int &dim) {
    bool flipped = false;
    if (mat_two_sided(M) && wi.z < 0) { wi.z *= -1; flipped = true; }
    F3 result = mat_sample_one(M, wi, sx, sy, wo, pdf, delta, eta, isnull, key, dim);
    if (flipped && !iszero3(result) && pdf != 0) wo.z *= -1;
    return result;
}

// ------------------------------------------------------------------------------------------------
// SD-tree
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(32))) SNode {  // sampling quadtree node, 32 B
    float sum[4];
    unsigned short child[4];
    unsigned int pad[2];
};

struct __attribute__((aligned(16))) LeafHdr {  // one per S-tree node, 64 B
    unsigned int s_base, s_num;  // sampling D-tree: first node in the SNode pool, node count
    float s_sum, s_statw;        // DTree::m_atomic {sum, statisticalWeight} of the sampling tree (GP:538-557)
    unsigned int b_base, b_num;  // building D-tree block in bchild / bacc
    int s_depth, b_depth;        // m_maxDepth
    float theta;                 // AdamOptimizer::State (GP:116-124)
    int adam_iter;
    float adam_m, adam_v;
    float b_statw;               // statisticalWeightBuilding() as float (valid after build; halved by refine)
    float adam_bg, adam_ba;      // AdamOptimizer::State::batchGradient / batchAccumulation (GP:121-122): the partial batch of append()
    unsigned int pad;
};

// One deferred DTreeWrapper::optimizeBsdfSamplingFraction call (= ppg_adam_record of include/ppg.h)
struct __attribute__((aligned(16))) AdamRec {
    unsigned long long key;  // leaf << PPG_ADAM_LEAF_SHIFT | path << PPG_ADAM_CODE_BITS | code
    float product, woPdf, bsdfPdf, dTreePdf, weight, pad;
};

struct DevTree {
    const int4 *stree;            // {axis, child0, child1, -}
    LeafHdr *hdr;                 // per S-tree node
    const SNode *snodes;          // sampling pool
    const ushort4 *bchild;        // building pool topology
    unsigned long long *bacc;     // building pool accumulators [node*4 + slot], 2^-24 fixed point
    unsigned long long *bweight;  // per S-tree node: building statistical weight accumulator (folded, see *_rep)
    // Records of the round for the sampling-fraction optimiser (include/ppg.h "Learning the BSDF sampling fraction"): written at
    // position adam_base[path] + vertex when adam_base != nullptr (every record's place is known in advance: nearest / stochastic
    // spatial filter without next-event estimation — the buffer is then already ordered by (path, code) and only a stable sort
    // by leaf remains), otherwise appended through adam_count[0] (adam_count[1] = overflow flag) and sorted by the whole key.
    unsigned long long *adam_keys;
    AdamRec *adam_recs;
    unsigned int *adam_count;
    const unsigned int *adam_base;
    unsigned int adam_cap;
    // Replicated accumulation targets [node * PPG_REPLICAS + r]: a popular S-tree leaf receives several percent of all
    // records of a pass and one address sustains only ~90 atomics/µs; workgroups spread over the replicas,
    // k_fold_replicas adds them into the compact arrays above (integer sums: exact).
    unsigned long long *bweight_rep;
    float aabb_min[3], aabb_ext[3];  // cubified AABB (GP:857-859)
    float aabb_max[3];
    int is_built;
    // Lookup accelerator: the S-tree cycles its split axis x,y,z, so the first 3*GRID_BITS levels of any descent
    // are determined by the top GRID_BITS bits of each normalised coordinate.  grid[cell] = node reached after
    // min(3*GRID_BITS, depth of the leaf) levels | levels << 27.  One read replaces up to 18 dependent ones.
    const unsigned int *grid;
    // The optimiser's variable per S-tree node as it was when the round began (nullptr: read it in the leaf header).  A round's stragglers are
    // finished BESIDE the application of that round's records and the next round's (ppg_hip.hip "Stragglers"), which rewrite the headers;
    // they must go on sampling with the fractions of their own round (include/ppg.h: all paths of a round see the fractions in effect at its start).
    const float *theta_frozen;
};
#define PPG_REPLICAS 32
#define PPG_GRID_BITS 6
#define PPG_GRID_DIM (1 << PPG_GRID_BITS)
#define PPG_GRID_LEVELS (3 * PPG_GRID_BITS)

// STree::dTreeWrapper (GP:897-905) + STreeNode::dTreeWrapper (GP:761-769): leaf index and voxel size.
// The reference halves p[axis] / size[axis] level by level; every one of those operations is exact in
// binary floating point (×2, −0.5 on [0.5,1), ÷2), so jumping 18 levels through the grid and computing
// the voxel size as extent · 2^-n gives bit-identical results (tests compare against the literal descent
// of the oracle).
D int stree_lookup(const DevTree &T, F3 pw, F3 &size) {
    size = f3(T.aabb_ext[0], T.aabb_ext[1], T.aabb_ext[2]);
    float p[3];
    p[0] = (pw.x - T.aabb_min[0]) / size.x;
    p[1] = (pw.y - T.aabb_min[1]) / size.y;
    p[2] = (pw.z - T.aabb_min[2]) / size.z;
    int cell[3];
    float rem[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float s = p[a] * (float)PPG_GRID_DIM;
        int c = (s < (float)PPG_GRID_DIM) ? ((s < 0.0f) ? 0 : (int)s) : (PPG_GRID_DIM - 1);  // out-of-cube points follow child 0 / child 1 forever
        cell[a] = c;
        rem[a] = s - (float)c;
    }
    const unsigned int e = T.grid[(cell[2] * PPG_GRID_DIM + cell[1]) * PPG_GRID_DIM + cell[0]];
    int idx = (int)(e & 0x07ffffffu);
    int depth = (int)(e >> 27);
    if (depth == PPG_GRID_LEVELS) {  // possibly deeper than the grid: continue the literal descent on the remainders
        for (;;) {
            int4 n = T.stree[idx];
            if (n.y == 0) break;
            int a = n.x;
            ++depth;
            if (rem[a] < 0.5f) {
                rem[a] *= 2;
                idx = n.y;
            } else {
                rem[a] = (rem[a] - 0.5f) * 2;
                idx = n.z;
            }
        }
    }
    // levels along x, y, z of a leaf at depth D (root splits x, then y, z, x, ...)
    size = f3(size.x * ppg_exp2i(-((depth + 2) / 3)), size.y * ppg_exp2i(-((depth + 1) / 3)), size.z * ppg_exp2i(-(depth / 3)));
    return idx;
}

// DTreeWrapper::canonicalToDir / dirToCanonical (GP:586-608)
D F3 canonical_to_dir(float px, float py) {
    const float cosTheta = 2 * px - 1;
    const float phi = 2 * PPG_PI_F * py;
    const float sinTheta = __builtin_sqrtf(1 - cosTheta * cosTheta);
    float sinPhi, cosPhi;
    dm_sincos(phi, &sinPhi, &cosPhi);
    return f3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
D void dir_to_canonical(F3 d, float &px, float &py) {
    if (!ppg_isfinite(d.x) || !ppg_isfinite(d.y) || !ppg_isfinite(d.z)) { px = 0; py = 0; return; }
    const float cosTheta = ppg_min(ppg_max(d.z, -1.0f), 1.0f);
    float phi = dm_atan2(d.y, d.x);
    while (phi < 0) phi = (float)((double)phi + 2.0 * (double)PPG_PI_F);
    px = (cosTheta + 1) / 2;
    py = phi / (2 * PPG_PI_F);
}

// QuadTreeNode::childIndex (GP:205-217)
D int quad_child_index(float &px, float &py) {
    int res = 0;
    if (px < 0.5f) px *= 2; else { px = (px - 0.5f) * 2; res |= 1; }
    if (py < 0.5f) py *= 2; else { py = (py - 0.5f) * 2; res |= 2; }
    return res;
}

// One sampling node in registers: sums as float4, children as packed u16x4.  Dynamic indexing of a struct
// would send it through LDS/scratch and serialise its two 16-byte loads; selects keep it in VGPRs.
struct NodeR {
    float4 s;
    uint2 c;
};
D NodeR load_node(const SNode *p) {
    NodeR n;
    const float4 *q = reinterpret_cast<const float4 *>(p);
    n.s = q[0];
    float4 t = q[1];
    n.c = make_uint2(__float_as_uint(t.x), __float_as_uint(t.y));
    return n;
}
D float node_sum(const NodeR &n, int i) { return i == 0 ? n.s.x : (i == 1 ? n.s.y : (i == 2 ? n.s.z : n.s.w)); }
D unsigned int node_child(const NodeR &n, int i) {
    unsigned int w = (i & 2) ? n.c.y : n.c.x;
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

// DTree::mean (GP:387-393)
D float dtree_mean(float sum, float statw) {
    if (statw == 0) return 0;
    const float factor = 1 / (PPG_PI_F * 4 * statw);
    return factor * sum;
}

// what sampling / pdf evaluation need of a leaf's header
struct DTreeRef {
    unsigned int s_base;
    float s_sum, s_statw;
};
D DTreeRef dtree_ref(const LeafHdr &h) { DTreeRef r; r.s_base = h.s_base; r.s_sum = h.s_sum; r.s_statw = h.s_statw; return r; }

// DTree::pdf + QuadTreeNode::pdf (GP:415-421, 232-245), iterative
template <typename Stack>
D float dtree_pdf(const DevTree &T, const DTreeRef &h, float px, float py, Stack factors) {
    if (!(dtree_mean(h.s_sum, h.s_statw) > 0)) return 1 / (4 * PPG_PI_F);
    // the recursion multiplies factors from the leaf upwards: f1 * (f2 * (f3 * ...)); keep that order
    int nf = 0;
    unsigned int node = 0;
    float result;
    for (;;) {
        const NodeR n = load_node(T.snodes + h.s_base + node);
        const int index = quad_child_index(px, py);
        const float si = node_sum(n, index);
        if (!(si > 0)) { result = 0; break; }
        const float factor = 4 * si / (n.s.x + n.s.y + n.s.z + n.s.w);
        const unsigned int c = node_child(n, index);
        if (c == 0) { result = factor; break; }
        factors[nf++] = factor;
        node = c;
    }
    for (int i = nf - 1; i >= 0; --i) result = factors[i] * result;
    return result / (4 * PPG_PI_F);
}

// DTree::sample + QuadTreeNode::sample (GP:431-442, 257-301), iterative.
// The recursion returns origin + 0.5 * child.sample(); unrolled as a stack of origins.
D void dtree_sample(const DevTree &T, const DTreeRef &h, uint32_t key, uint32_t &dim, float &ox, float &oy) {
    if (!(dtree_mean(h.s_sum, h.s_statw) > 0)) {
        ox = ppg_rand(key, dim++);
        oy = ppg_rand(key, dim++);
        return;
    }
    unsigned int orgx = 0, orgy = 0;  // origin.x / origin.y of level i is 0.5 iff bit i is set (GP:271, 282, 291)
    int depth = 0;
    unsigned int node = 0;
    float rx, ry;
    for (;;) {
        const NodeR n = load_node(T.snodes + h.s_base + node);
        int index = 0;
        float topLeft = n.s.x;
        float topRight = n.s.y;
        float partial = topLeft + n.s.z;
        float total = partial + topRight + n.s.w;
        if (!(total > 0.0f)) {
            rx = ppg_rand(key, dim++);
            ry = ppg_rand(key, dim++);
            break;
        }
        float boundary = partial / total;
        float ogx = 0.0f, ogy = 0.0f;
        float sample = ppg_rand(key, dim++);
        if (sample < boundary) {
            sample /= boundary;
            boundary = topLeft / partial;
        } else {
            partial = total - partial;
            ogx = 0.5f;
            sample = (sample - boundary) / (1.0f - boundary);
            boundary = topRight / partial;
            index |= 1 << 0;
        }
        if (sample < boundary) {
            sample /= boundary;
        } else {
            ogy = 0.5f;
            sample = (sample - boundary) / (1.0f - boundary);
            index |= 1 << 1;
        }
        if (ogx != 0.0f) orgx |= 1u << depth;
        if (ogy != 0.0f) orgy |= 1u << depth;
        ++depth;
        const unsigned int c = node_child(n, index);
        if (c == 0) {
            rx = ppg_rand(key, dim++);
            ry = ppg_rand(key, dim++);
            break;
        }
        node = c;
    }
    for (int i = depth - 1; i >= 0; --i) {
        rx = (((orgx >> i) & 1u) ? 0.5f : 0.0f) + 0.5f * rx;
        ry = (((orgy >> i) & 1u) ? 0.5f : 0.0f) + 0.5f * ry;
    }
    ox = ppg_min(ppg_max(rx, 0.0f), 1.0f);
    oy = ppg_min(ppg_max(ry, 0.0f), 1.0f);
}

// per-thread column of a [depth][threads] LDS array: conflict-free, costs no registers
struct LdsColumn {
    float *base;
    int stride;
    D float &operator[](int i) const { return base[i * stride]; }
};
struct RegColumn {
    float v[24];
    D float &operator[](int i) { return v[i]; }
};

D float logistic(float x) { return 1 / (1 + dm_exp(-x)); }  // GP:64-66

#endif
