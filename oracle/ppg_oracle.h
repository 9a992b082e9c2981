/*
 * ppg_oracle.h — C interface of the CPU oracle (libppg_oracle.so).  TEST INFRASTRUCTURE ONLY.
 *
 * Same shape as include/ppg.h (prefix ppgo_ instead of ppg_) so that the parity tests drive the
 * HIP product and the oracle with identical call sequences.  See ppg_oracle.cpp for what each
 * function restates and for the parity-pin status.
 */
#ifndef PPG_ORACLE_H
#define PPG_ORACLE_H

#include "../include/ppg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppgo_ctx ppgo_ctx;

enum { PPGO_ACC_FIXED = 0, PPGO_ACC_FLOAT = 1 };
/* ROUND: the product's rule (include/ppg.h "Learning the BSDF sampling fraction": records applied at the end of every round in
   key order).  SEQUENTIAL: GP:672-697 literally — every record is applied the moment its path commits it (single thread). */
enum { PPGO_ADAM_ROUND = 0, PPGO_ADAM_SEQUENTIAL = 1, PPGO_ADAM_HALF = 2 /* measurement only: half-pass rounds in iteration 1 */,
       /* measurement only (round 6): PPGO_ADAM_REGIONS + R — in the iterations of up to 16 passes every pass is rendered in R
          groups of image blocks, consecutive in the spiral order in which the reference's scheduler hands blocks out (imageproc.cpp:29-80), and
          the optimiser is applied after every group: the variable then follows the image REGION being rendered, as the reference's does */
       PPGO_ADAM_REGIONS = 16 };

int ppgo_create(const ppg_config *cfg, ppgo_ctx **out);
void ppgo_destroy(ppgo_ctx *ctx);
const char *ppgo_last_error(const ppgo_ctx *ctx);
/* acc_mode: PPGO_ACC_*, adam_mode: PPGO_ADAM_*, threads: OpenMP threads for the pixel loop (FIXED mode
   stays bit-identical for any thread count; FLOAT / SEQUENTIAL force 1). */
int ppgo_set_modes(ppgo_ctx *ctx, int32_t acc_mode, int32_t adam_mode, int32_t threads);
int ppgo_set_scene(ppgo_ctx *ctx, const ppg_scene *scene);
int ppgo_set_shard(ppgo_ctx *ctx, int32_t rank, int32_t world, int32_t tile_size);

int ppgo_render(ppgo_ctx *ctx);
int ppgo_begin_render(ppgo_ctx *ctx);
int ppgo_begin_iteration(ppgo_ctx *ctx, int32_t is_final);
int ppgo_set_final(ppgo_ctx *ctx, int32_t is_final);
int ppgo_set_do_nee(ppgo_ctx *ctx, int32_t do_nee);
int ppgo_render_passes(ppgo_ctx *ctx, int32_t n_passes, ppg_pass_stats *stats);
int ppgo_render_passes_nostat(ppgo_ctx *ctx, int32_t n_passes);
int ppgo_finish_passes(ppgo_ctx *ctx, ppg_pass_stats *stats);
int ppgo_build_sdtree(ppgo_ctx *ctx, ppg_tree_stats *stats);
int ppgo_end_iteration(ppgo_ctx *ctx);
int ppgo_end_render(ppgo_ctx *ctx);
int ppgo_cancel(ppgo_ctx *ctx);
int ppgo_set_stop_hook(ppgo_ctx *ctx, ppg_stop_hook hook, void *user);
int32_t ppgo_final_group_passes(int32_t n_passes);  /* include/ppg.h "Final iteration: groups of passes" */
int ppgo_final_partials(ppgo_ctx *ctx, void **data, uint64_t *n_floats);
int ppgo_final_partials_commit(ppgo_ctx *ctx);
int ppgo_read_film(ppgo_ctx *ctx, float *rgb);
int ppgo_read_variance(ppgo_ctx *ctx, float *rgb);
int ppgo_dump_sdtree(ppgo_ctx *ctx, const char *path);

int ppgo_sdtree_info_get(ppgo_ctx *ctx, ppg_sdtree_info *info);
int ppgo_sdtree_read_stree(ppgo_ctx *ctx, int32_t *axis, uint32_t *children);
int ppgo_sdtree_read_dtree_headers(ppgo_ctx *ctx, int32_t which, uint64_t *offset, uint32_t *num_nodes,
                                   int32_t *max_depth, float *sum, double *stat_weight);
int ppgo_sdtree_read_dtree_nodes(ppgo_ctx *ctx, int32_t which, float *sums, uint16_t *children, uint64_t *fixed_sums);
int ppgo_sdtree_read_adam(ppgo_ctx *ctx, float *theta);

/* host-memory counterparts of ppg_sdtree_stat_buffers / ppg_film_buffers / ppg_image_buffers for the
   gloo tests of the sharded driver: export copies the accumulators out, import overwrites them. */
int ppgo_stat_export(ppgo_ctx *ctx, uint64_t *sums, uint64_t n_sums, uint64_t *weights, uint64_t n_weights);
int ppgo_stat_import(ppgo_ctx *ctx, const uint64_t *sums, uint64_t n_sums, const uint64_t *weights, uint64_t n_weights);
int ppgo_stat_sizes(ppgo_ctx *ctx, uint64_t *n_sums, uint64_t *n_weights);
/* levels visited since create: {S-tree levels, lookups, D-tree sample levels, calls, pdf levels, calls, record levels, calls} */
int ppgo_work_counters(ppgo_ctx *ctx, uint64_t *out8);
/* number of paths by final depth since create, PPGO_LEN_HIST bins (the last one collects everything longer) */
#define PPGO_LEN_HIST 4096
int ppgo_path_length_histogram(ppgo_ctx *ctx, uint64_t *out);
/* the tests' switch of include/ppg_testhooks.h (ppg_debug_set_defer_depth), same meaning */
int ppgo_debug_set_defer_depth(ppgo_ctx *ctx, int32_t depth);
int ppgo_set_adam_regions(ppgo_ctx *ctx, int32_t regions);  /* = ppg_set_adam_regions */
int ppgo_set_pass_hook(ppgo_ctx *ctx, ppg_pass_hook hook, void *user);
/* host-memory counterparts of ppg_adam_records / ppg_adam_records_replace (valid inside the round hook) */
int ppgo_adam_records(ppgo_ctx *ctx, void **records, uint64_t *n);
int ppgo_adam_records_replace(ppgo_ctx *ctx, const void *records, uint64_t n);
/* one owner per D-tree (include/ppg.h "Sharded optimiser"): same calls as the product, on host memory */
int ppgo_hook_phase(ppgo_ctx *ctx, int32_t *phase);
int ppgo_adam_records_by_owner(ppgo_ctx *ctx, int32_t world, void **records, uint64_t *counts);
int ppgo_adam_state(ppgo_ctx *ctx, int32_t world, void **state, uint64_t *segment);
int ppgo_adam_state_commit(ppgo_ctx *ctx);
int ppgo_film_ptrs(ppgo_ctx *ctx, float **rgb_sum, float **weight);
int ppgo_image_ptrs(ppgo_ctx *ctx, float **image, float **sq_image);
int ppgo_image_weight_ptr(ppgo_ctx *ctx, float **w);

int ppgo_query_pdf(ppgo_ctx *ctx, uint32_t n, const float *positions, const float *dirs, float *pdf_out);
int ppgo_query_sample(ppgo_ctx *ctx, uint32_t n, const float *positions, uint64_t seed, float *dirs_out);

/* ---- unit-level hooks for the known-answer tests ---- */
/* fresh DTreeWrapper: reset(20, rho) from an empty tree; returns numNodes, depth, pdf of +z */
int ppgo_ka_fresh_reset(float rho, uint32_t *num_nodes, int32_t *depth, float *pdf);
/* STree over the unit cube, root building weight W, refine(thr): returns leaf count */
int ppgo_ka_refine(float W, uint64_t thr, uint32_t *n_leaves, uint32_t *n_nodes);
/* n identical records through the literal Adam path (GP:672-697); returns bsdfSamplingFraction() */
int ppgo_ka_adam(int32_t n, float product, float wo_pdf, float bsdf_pdf, float dtree_pdf, float weight,
                 int32_t loss /*1 = kl, 2 = var*/, float *fraction);
/* canonicalToDir / dirToCanonical (GP:586-608) */
void ppgo_canonical_to_dir(float x, float y, float *d);
void ppgo_dir_to_canonical(const float *d, float *xy);
/* element-wise ppg_detmath.h / ppg_rng.h evaluation: op 0 sincos(a), 1 atan2(a,b), 2 exp(a), 3 fixed round trip, 4 rand, 5 powi, 6 log, 7 pow,
   8 draw of a path: ppg_rand(ppg_path_key(seed, pixel, sample), dim) with a = pixel, b = seed << 16 | sample << 4 | dim as bit patterns */
int ppgo_math_eval(int32_t op, uint32_t n, const float *a, const float *b, float *out0, float *out1);
int ppgo_bsdf_eval(const ppg_material *mat, uint32_t n, const float *wi, const float *wo, float *f_out, float *pdf_out);
int ppgo_bsdf_sample(const ppg_material *mat, uint32_t n, const float *wi, const float *sample_xy, float *wo_out, float *weight_out,
                     float *pdf_out, float *eta_out, int32_t *delta_out);
int ppgo_bsdf_flags(const ppg_material *mat, int32_t *is_smooth, int32_t *all_delta, int32_t *backside_or_transmission);
/* D-tree exercise: record `n` (canonical xy, irradiance, weight) samples into a fresh wrapper with the
   given filter / acc mode, build, reset(rho), record again, build; then evaluate pdf at `m` query
   points and draw `m` samples keyed (seed, i).  Outputs node arrays of the final sampling tree. */
int ppgo_dtree_exercise(int32_t acc_mode, int32_t directional_filter, float rho, uint32_t n, const float *xy,
                        const float *irradiance, const float *weight, uint32_t m, const float *query_xy,
                        uint64_t seed, float *pdf_out, float *sample_xy_out, uint32_t *num_nodes_out,
                        float *node_sums_out /*cap 65536*4*/, uint16_t *node_children_out, float *stat_weight_out,
                        float *tree_sum_out);

#ifdef __cplusplus
}
#endif
#endif
