/*
 * ppg_testhooks.h — entry points of libppg_hip.so that exist for the tests only.  Not part of the drop-in boundary (include/ppg.h): nothing
 * a host of the integrator needs, no reference counterpart.
 */
#ifndef PPG_TESTHOOKS_H
#define PPG_TESTHOOKS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Host only (no GPU is touched): the quantised 4-wide BVH that ppg_set_scene builds for the traversal kernels, from a triangle
   soup.  nodes_out receives up to nodes_cap 64-byte nodes — { float origin[3]; uint32 exps (byte a = biased exponent of the power-of-two cell
   size along axis a); uint32 qlo[3] (x, y, z: child k's lower plane in byte k); uint32 qhi[3]; int32 child[4] (>= 0 node, < 0 leaf:
   ~child = first << 3 | count - 1 in leaf order, 0x7fffffff unused); int32 pad[2] } —, *n_nodes their number (also when it exceeds the
   capacity), order_out[n_triangles] the triangles in leaf order.  tests/test_bvh_host.py checks with it, on the CPU, that no child box on
   the way to a triangle a ray hits is ever missed under the kernels' float arithmetic (csrc/ppg_device.h bvh4_children). */
int ppg_debug_build_bvh(const float *positions, const uint32_t *indices, uint32_t n_triangles, float pad_abs, int32_t max_leaf,
                        void *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes, uint32_t *order_out);

/* The depth beyond which a path is a STRAGGLER (include/ppg.h: PPG_ADAM_DEFER_DEPTH = 64, part of the result).  One path in 10^4 gets there
   in a real scene and none in most test scenes; the parity tests lower it (1 .. 64; the oracle has the same switch, ppgo_debug_set_defer_depth)
   so that thousands of paths go through the stragglers' machinery — hand-over inside k_tail, the second launch beside the next round, the
   records applied one round late — and must still come out bit-equal to the oracle. */
struct ppg_ctx;
int ppg_debug_set_defer_depth(struct ppg_ctx *ctx, int32_t depth);

#ifdef __cplusplus
}
#endif
#endif /* PPG_TESTHOOKS_H */
