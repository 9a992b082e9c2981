/*
 * ppg.h — C-ABI of libppg_hip.so, the MI355X-native GuidedPathTracer hot path.
 *
 * Every entry point replaces a piece of the reference integrator plugin
 * (mitsuba/src/integrators/path/guided_path.cpp, cited as GP:line) or of the Mitsuba Integrator
 * interface it implements (mitsuba/include/mitsuba/render/integrator.h, cited as IH:line).
 * The reference-side binding a Mitsuba maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are host pointers unless the name says `dev`.
 *   - every function returns PPG_OK (0) or a negative error code; ppg_last_error() gives the text.
 *     No exception crosses this boundary (the reference uses Assert/Log(EError) → throw; GP:1023).
 *   - one context per GPU, driven by one host thread; ppg_cancel() is the only call that may come
 *     from another thread (mirrors Integrator::cancel(), IH:84 / GP:1643-1648).
 *   - the caller owns every buffer it passes in; ppg_set_scene() copies.
 */
#ifndef PPG_H
#define PPG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPG_OK 0
#define PPG_ERR_INVALID (-1)   /* bad argument / unknown enum string (reference: Assert(false), GP:1023...) */
#define PPG_ERR_DEVICE (-2)    /* HIP error */
#define PPG_ERR_STATE (-3)     /* call out of order (no scene, render not begun, ...) */
#define PPG_ERR_CANCELLED (-4) /* ppg_cancel() was called; render() returns false in the reference (GP:1584) */
#define PPG_ERR_NOMEM (-5)

typedef struct ppg_ctx ppg_ctx;

/* ------------------------------------------------------------------------------------------------
 * Integrator properties.  Names, meaning and defaults are the reference's:
 *   GuidedPathTracer(const Properties&)      GP:1014-1085
 *   MonteCarloIntegrator(const Properties&)  mitsuba/src/librender/integrator.cpp:190-225
 * String-valued properties stay strings so that an unknown value is rejected exactly where the
 * reference asserts.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ppg_config {
    const char *nee;                      /* "never" | "kickstart" | "always"          GP:1015-1024 (default "never") */
    const char *sampleCombination;        /* "discard" | "automatic" | "inversevar"    GP:1026-1035 (default "automatic") */
    const char *spatialFilter;            /* "nearest" | "stochastic" | "box"          GP:1037-1046 (default "nearest") */
    const char *directionalFilter;        /* "nearest" | "box"                         GP:1048-1055 (default "nearest") */
    const char *bsdfSamplingFractionLoss; /* "none" | "kl" | "var"                     GP:1057-1066 (default "none") */
    int32_t sdTreeMaxMemory;              /* MB, -1 = unlimited                        GP:1068 */
    int32_t sTreeThreshold;               /* 12000                                     GP:1069 */
    float dTreeThreshold;                 /* 0.01                                      GP:1070 */
    float bsdfSamplingFraction;           /* 0.5                                       GP:1071 */
    int32_t sppPerPass;                   /* 4                                         GP:1072 */
    const char *budgetType;               /* "spp" | "seconds" (default "seconds")     GP:1074-1081 */
    float budget;                         /* 300                                       GP:1083 */
    int32_t dumpSDTree;                   /* 0                                         GP:1084 */
    /* MonteCarloIntegrator */
    int32_t rrDepth;                      /* 5   integrator.cpp:192 */
    int32_t maxDepth;                     /* -1  integrator.cpp:197 */
    int32_t strictNormals;                /* 0   integrator.cpp:213 */
    int32_t hideEmitters;                 /* 0   integrator.cpp:218 */
    /* Build-specific (no reference counterpart) */
    uint64_t seed;                        /* key of the counter-based sampler that replaces the per-thread SFMT
                                             streams of samplers/independent.cpp (XML `seed` is ignored there) */
    int32_t device;                       /* HIP device ordinal */
    const char *dumpPrefix;               /* "<dest>" of the "<dest>-NN.sdt" dumps, GP:1192-1195; may be NULL */
} ppg_config;

/* Fills *cfg with the reference defaults listed above. */
void ppg_config_default(ppg_config *cfg);

/* ------------------------------------------------------------------------------------------------
 * Scene: what GuidedPathTracer::render() reaches through `Scene*` (GP:1516-1527, 1784, 1934, 2193),
 * flattened.  Triangles only; per-triangle BSDF and emitter indices.
 * ---------------------------------------------------------------------------------------------- */
enum {
    PPG_BSDF_DIFFUSE = 0,          /* mitsuba/src/bsdfs/diffuse.cpp:110-150 (one-sided Lambertian) */
    PPG_BSDF_TWOSIDED_DIFFUSE = 1, /* = PPG_BSDF_DIFFUSE with PPG_MAT_TWOSIDED (kept for the round-1 callers) */
    PPG_BSDF_MIRROR = 2,           /* conductor.cpp:220-290 with material "none" (eta = 0, k = 1: Fresnel = 1): ideal specular
                                      reflection, reflectance = specularReflectance; a delta BSDF — never guided (GP:1942-1944, 1654) */
    PPG_BSDF_CONDUCTOR = 3,        /* conductor.cpp:220-290: smooth conductor, fresnelConductorExact(eta, k) (util.cpp:739-761) */
    PPG_BSDF_ROUGHCONDUCTOR = 4,   /* roughconductor.cpp:247-415 with an isotropic GGX (or, PPG_MAT_BECKMANN, Beckmann) distribution and visible-normal
                                      sampling (microfacet.h:191-276, 425-540, 645-690): glossy ⇒ smooth ⇒ guided */
    PPG_BSDF_PLASTIC = 5,          /* plastic.cpp:247-455: delta specular coat over a diffuse base — the mixed delta/smooth
                                      case of sampleMat (GP:1672-1676) */
    PPG_BSDF_DIELECTRIC = 6,       /* dielectric.cpp:229-400: smooth glass, delta reflection + delta refraction, tracks eta */
    PPG_BSDF_THINDIELECTRIC = 7    /* thindielectric.cpp:152-252: delta reflection + a NULL (pass-through) component — exercises Li's
                                      null branch (GP:2045-2075) and the look-through of rayIntersectAndLookForEmitter /
                                      evalTransmittance (GP:2184-2245, scene.cpp:619-679) */
    ,
    PPG_BSDF_ROUGHDIELECTRIC = 8   /* roughdielectric.cpp:268-606: microfacet reflection + refraction (GGX / Beckmann, visible normals) — a smooth,
                                      TRANSMISSIVE BSDF: guided on both sides; its sample() draws one extra number from the path's sampler */,
    PPG_BSDF_ROUGHPLASTIC = 9      /* roughplastic.cpp:330-501: rough dielectric coating (GGX / Beckmann, visible normals) over a diffuse base; the
                                      energy balance between the two comes from Mitsuba's precomputed rough-transmittance tables
                                      (data/microfacet/{ggx,beckmann}.dat, rtrans.h), handed over as the per-material slice `ppg_scene.rtrans[material.rtrans]` */
};
#define PPG_BSDF_LAST PPG_BSDF_ROUGHPLASTIC
enum {
    PPG_MAT_TWOSIDED = 1,          /* wrap the (one-sided) BRDF in twosided.cpp:100-180, same BRDF on both sides */
    PPG_MAT_NONLINEAR = 2,         /* plastic: nonlinear = true (plastic.cpp:164) */
    PPG_MAT_BECKMANN = 8,          /* roughconductor: Beckmann distribution (Mitsuba's default) instead of GGX — microfacet.h:199-205,
                                      487-498, 565-642 (visible-normal sampling by numerical inversion), math.cpp:25-72 (erf, erfinv) */
    PPG_MAT_MASK = 4               /* wrap the BSDF (outside a two-sided adapter, if any) in mask.cpp:108-214 with constant `opacity`: a
                                      smooth/null hybrid — its sampled pass-through is recorded for the sampling-fraction optimiser
                                      (GP:2047-2068) */
};
enum { PPG_WRAP_REPEAT = 0, PPG_WRAP_MIRROR = 1, PPG_WRAP_CLAMP = 2, PPG_WRAP_ZERO = 3, PPG_WRAP_ONE = 4 }; /* ReconstructionFilter::EBoundaryCondition, mipmap.h:503-560 */

typedef struct ppg_texture {
    /* A bitmap texture as THIS integrator sees it (textures/bitmap.cpp).  Li fetches the BSDF with its.getBSDF() (GP:1934), not
       getBSDF(ray), so Intersection::computePartials never runs, its.hasUVPartials stays false and Texture2D::eval (texture.cpp:112-121)
       always takes the unfiltered branch: BitmapTexture::eval(uv) (bitmap.cpp:431-452) = MIPMap::evalBilinear(0, uv) (mipmap.h:575-596)
       on the full-resolution image — or evalBox for filterType "nearest" — whatever filterType the scene asks for; bump maps read
       evalGradientBilinear(0, uv) (mipmap.h:601-626).  uv = its.uv * (uscale, vscale) + (uoffset, voffset). */
    uint32_t width, height;
    const float *rgb;          /* [height * width * 3] linear RGB: the image after the loader's gamma / sRGB decoding (bitmap.cpp:182-260),
                                  row 0 = first row of the file; luminance images replicated to three channels */
    float uv_scale[2], uv_offset[2];
    int32_t wrap_u, wrap_v;    /* PPG_WRAP_*: wrapModeU / wrapModeV */
    int32_t nearest;           /* filterType = nearest */
} ppg_texture;


typedef struct ppg_material {
    int32_t type;         /* PPG_BSDF_* */
    float reflectance[3]; /* linear RGB (SPECTRUM_SAMPLES=3 build of the reference): diffuse `reflectance`; plastic
                             `diffuseReflectance`; conductors and dielectric `specularReflectance` */
    float specular[3];    /* plastic `specularReflectance`; dielectric / thindielectric `specularTransmittance` */
    float alpha;          /* roughconductor / roughdielectric: roughness `alpha` (clamped to >= 1e-4 like microfacet.h:135) */
    float eta[3];         /* conductors: eta per channel (already divided by extEta, roughconductor.cpp:185-186);
                             plastic / (rough / thin) dielectric: eta[0] = intIOR / extIOR */
    float k[3];           /* conductors: k per channel */
    int32_t flags;        /* PPG_MAT_* */
    int32_t rtrans;       /* roughplastic: index of this material's slice in ppg_scene.rtrans; otherwise 0 */
    float opacity[3];     /* PPG_MAT_MASK: opacity (mask.cpp, default 0.5) */
    uint32_t texture;     /* bits 0..15: 1 + index (ppg_scene.textures) of the bitmap on the diffuse reflectance — `reflectance` of diffuse,
                             `diffuseReflectance` of plastic / roughplastic — in which case reflectance[] holds the texture's average
                             (Texture::getAverage, used by the plug-ins' configure() for the component sampling weights, plastic.cpp:191-204);
                             bits 16..31: 1 + index of the displacement texture of a `bumpmap` adapter around this BSDF (bumpmap.cpp:135-219:
                             shading frame perturbed by the texture's gradient); 0 = none */
} ppg_material;           /* 80 bytes */

typedef struct ppg_emitter {
    float radiance[3]; /* area light, mitsuba/src/emitters/area.cpp:104-109 */
    float _pad;
} ppg_emitter;

typedef struct ppg_sphere {
    /* An analytic sphere (shapes/sphere.cpp:106-372): double-precision quadratic for the ray test (:164-189), intersection record and
       (theta, phi) tangent frame of :213-263, Shirley et al. cone sampling / uniform area sampling for next-event estimation
       (:291-378).  Spheres are primitives number n_triangles, n_triangles + 1, .. for the closest-hit tie rule. */
    float center[3];
    float radius;         /* > 0 (sphere.cpp:129-130); the scale of `toWorld` is folded in (:113-121) */
    float to_world[9];    /* row-major rotation left of `toWorld` once its scale is removed (m_objectToWorld's linear part, :116-120);
                             identity for an untransformed sphere.  The inverse is taken as the transpose. */
    uint32_t material;    /* index into materials */
    int32_t emitter;      /* index into emitters, -1 = none; an emitter id carried by a sphere must not be used by any other shape */
    int32_t flip_normals; /* sphere.cpp:124, 256-257, 351-352 */
} ppg_sphere;             /* 64 bytes */

typedef struct ppg_envmap {
    /* Image-based environment emitter (emitters/envmap.cpp): a latitude-longitude radiance map, bilinearly interpolated
       (mipmap.h:575-596, repeat in u / clamp in v), importance-sampled by luminance x sin(theta) with the tent-filtered pixel
       sampling of envmap.cpp:557-595 and its pdf (:598-633); the bounding sphere / emitter numbering of the constant emitter.
       Deviation: a camera ray that leaves the scene is looked up bilinearly at level 0 as well (the reference filters it with
       EWA over the ray differentials, envmap.cpp:389-404) — this only smooths the directly visible background. */
    uint32_t width, height;   /* <= 65535 */
    const float *rgb;         /* [height * width * 3] linear RGB, row 0 = +y (theta = 0), column 0 = -z turning towards +x */
    float scale;              /* envmap.cpp:189 */
    float to_world[9];        /* row-major rotation of the emitter's toWorld (inverse taken as transpose); identity if none */
} ppg_envmap;

typedef struct ppg_camera {
    /* row-major 4x4, exactly the matrices of mitsuba/src/sensors/perspective.cpp:150-164 (m_sampleToCamera)
       and the sensor's world transform; rays follow perspective.cpp:271-298 */
    float sample_to_camera[16];
    float camera_to_world[16];
    float near_clip, far_clip;
    int32_t width, height; /* film / crop size in pixels */
} ppg_camera;

typedef struct ppg_scene {
    uint32_t n_vertices;
    const float *positions;       /* [n_vertices*3] */
    const float *normals;         /* [n_vertices*3] or NULL → face normals (skdtree.h:388-401) */
    uint32_t n_triangles;
    const uint32_t *indices;      /* [n_triangles*3] */
    const uint32_t *tri_material; /* [n_triangles] index into materials */
    const int32_t *tri_emitter;   /* [n_triangles] index into emitters, -1 = not an emitter */
    uint32_t n_materials;
    const ppg_material *materials;
    uint32_t n_emitters;
    const ppg_emitter *emitters;
    ppg_camera camera;
    const float *environment;     /* NULL, or float[3]: radiance of a constant environment emitter (emitters/constant.cpp) — what rays
                                     that leave the scene see (GP:1902-1914, 2236-2243) and one more emitter for luminaire sampling */
    /* Rough-transmittance slices for PPG_BSDF_ROUGHPLASTIC (0 / 0 / NULL without such materials).  One slice per (distribution, alpha,
       eta) = what RoughPlastic::configure() (roughplastic.cpp:285-305) leaves in its two RoughTransmittance objects:
         [0 .. rtrans_samples)  m_externalRoughTransmittance after setEta(eta), setAlpha(alpha) (rtrans.h:299-400): transmittance over
                                the warped incident cosine |cos|^(1/4) in [0, 1], read by eval() through evalCubicInterp1D;
         [rtrans_samples]       m_internalRoughTransmittance->evalDiffuse(alpha) after setEta(1/eta) (the diffuse transmittance from
                                inside: 1 - Fdr, roughplastic.cpp:372). */
    uint32_t n_rtrans;
    uint32_t rtrans_samples;      /* thetaSamples of the data file (100 in Mitsuba's tables), >= 2 */
    const float *rtrans;          /* [n_rtrans * (rtrans_samples + 1)] */
    uint32_t n_spheres;
    const ppg_sphere *spheres;    /* [n_spheres] or NULL */
    const ppg_envmap *envmap;     /* NULL, or the image-based environment emitter (not together with `environment`) */
    const float *texcoords;       /* NULL, or [n_vertices * 2] per-vertex texture coordinates (TriMesh::m_texcoords; its.uv is their barycentric
                                     interpolation, skdtree.h:403-410 — without them its.uv = the barycentrics (b1, b2)).  A triangle whose three
                                     vertices all carry NaN has no texture coordinates (meshes with and without `vt` share the vertex array).
                                     Triangles WITH coordinates whose BSDF uses a texture also get the reference's UV tangents as dpdu / dpdv
                                     (TriMesh::computeUVTangents, trimesh.cpp:683-735; skdtree.h:374-381) */
    uint32_t n_textures;
    const ppg_texture *textures;  /* [n_textures] or NULL */
} ppg_scene;

/* ------------------------------------------------------------------------------------------------
 * Statistics returned by the stepwise calls (the numbers the reference logs).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ppg_pass_stats {   /* GP:1321-1326 "… seconds, Total passes, Var, TTUV, STUV" */
    double seconds;
    int32_t passes_rendered_total;
    int32_t passes_rendered_local;
    float variance;
    uint64_t samples;             /* pixels × spp rendered by this call (this shard) */
    uint64_t rays;                /* "Normal rays traced" (skdtree.cpp:46,123) */
    uint64_t path_length_sum;     /* Σ rRec.depth, avgPathLength GP:2147-2148 */
    uint64_t vertices_committed;  /* records that reached DTreeWrapper::record */
} ppg_pass_stats;

typedef struct ppg_tree_stats {   /* GP:1176-1186 "Distribution statistics" */
    int32_t min_depth, max_depth;  float avg_depth;
    float min_mean_radiance, avg_mean_radiance, max_mean_radiance;
    uint64_t min_nodes, max_nodes; float avg_nodes;
    float min_stat_weight, avg_stat_weight, max_stat_weight;
    uint32_t n_leaves;            /* S-tree leaves (nPoints, GP:1135) */
    uint32_t n_stree_nodes;
    uint64_t n_dtree_nodes;       /* Σ numNodes over the sampling D-trees */
} ppg_tree_stats;

/* ------------------------------------------------------------------------------------------------
 * Life cycle.
 * ---------------------------------------------------------------------------------------------- */
/* Replaces CreateInstance(props) → new GuidedPathTracer(props) (cobject.h:99-107, GP:1014, 2421-2422). */
int ppg_create(const ppg_config *cfg, ppg_ctx **out);
void ppg_destroy(ppg_ctx *ctx);
const char *ppg_last_error(const ppg_ctx *ctx); /* ctx may be NULL: error of the last failed ppg_create */
const char *ppg_description(void);              /* GetDescription(), cobject.h:107: "Guided path tracer" */

/* Replaces Scene::preprocess/initialize as far as this path needs it (scene.cpp:322-384): copies the
   triangles, builds the BVH that stands in for the SAH kd-tree (skdtree.cpp:112-142), uploads. */
int ppg_set_scene(ppg_ctx *ctx, const ppg_scene *scene);

/* Multi-GPU image sharding (no reference counterpart; BlockedRenderProcess hands 32×32 blocks to
   worker threads, renderproc.cpp:69-87).  Tile t (row-major, tile_size² pixels) is rendered by
   rank t % world.  Default: rank 0 of 1. */
int ppg_set_shard(ppg_ctx *ctx, int32_t rank, int32_t world, int32_t tile_size);

/* ------------------------------------------------------------------------------------------------
 * Rendering.  ppg_render() is GuidedPathTracer::render() (GP:1516-1585, IH:74-75) in one call.
 * The stepwise calls expose its phases so a multi-GPU driver can all-reduce the SD-tree statistics
 * between ppg_render_passes() and ppg_build_sdtree(); ppg_render() is exactly their composition
 * (renderSPP GP:1342-1426 / renderTime GP:1434-1514).
 * ---------------------------------------------------------------------------------------------- */
int ppg_render(ppg_ctx *ctx);

int ppg_begin_render(ppg_ctx *ctx);                       /* GP:1519-1550: new STree(aabb), buffers, m_iter = 0 */
int ppg_begin_iteration(ppg_ctx *ctx, int32_t is_final);  /* GP:1378-1381: m_isFinalIter, film->clear(), resetSDTree() GP:1108-1113 */
int ppg_set_final(ppg_ctx *ctx, int32_t is_final);        /* GP:1409 / 1492: m_isFinalIter = true before the FINAL passes */
int ppg_set_do_nee(ppg_ctx *ctx, int32_t do_nee);         /* GP:1362 m_doNee */
int ppg_render_passes(ppg_ctx *ctx, int32_t n_passes, ppg_pass_stats *stats); /* performRenderPasses GP:1210-1329 */
int ppg_build_sdtree(ppg_ctx *ctx, ppg_tree_stats *stats);                    /* buildSDTree GP:1115-1189 */
int ppg_end_iteration(ppg_ctx *ctx);                      /* GP:1417-1422: optional dump, ++m_iter */
int ppg_end_render(ppg_ctx *ctx);                         /* GP:1567-1582: inverse-variance combination into the film */

/* Integrator::cancel() (IH:84, GP:1643-1648).  Thread-safe; the running ppg_render / ppg_render_passes
   call returns PPG_ERR_CANCELLED.  The request is sticky: with no render under way — the context exists and its scene is still being set
   up, seconds of BVH build — it cancels the NEXT one (ppg_begin_render / ppg_render return PPG_ERR_CANCELLED at once and consume it).  A
   request a render has acted on is spent and does not reach the render after it. */
int ppg_cancel(ppg_ctx *ctx);
/* Device blocks released by contexts (buffer growth, ppg_destroy) are kept by the library — at most a third of the device's memory —
   and handed to the next allocation that fits, of this or another context (hipMalloc / hipFree of hundreds of MB are synchronous,
   millisecond-scale calls).  This gives the kept blocks of `device` back to the driver; the library also does so by itself when an
   allocation fails. */
int ppg_release_cached_memory(int32_t device);

/* Film read-back: weight-normalised RGB, row-major [height][width][3] (hdrfilm develop, fmtconv.cpp:1036-1044). */
int ppg_read_film(ppg_ctx *ctx, float *rgb);
/* Per-pixel variance estimate of the last ppg_render_passes (m_varianceBuffer, GP:1298-1314), [h][w][3]. */
int ppg_read_variance(ppg_ctx *ctx, float *rgb);

/* dumpSDTree (GP:1191-1208) in the byte format of DTreeWrapper::dump (GP:699-711), readable by
   visualizer/src/main.cpp:142-176.  camera_matrix = row-major 4x4 world transform. */
int ppg_dump_sdtree(ppg_ctx *ctx, const char *path);

/* ------------------------------------------------------------------------------------------------
 * SD-tree access (parity tests, multi-GPU reduction, visualisation).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ppg_sdtree_info {
    uint32_t n_stree_nodes;
    uint32_t n_leaves;
    uint64_t n_sampling_nodes; /* total quadtree nodes of all sampling D-trees (shared blocks counted once) */
    uint64_t n_building_nodes; /* total quadtree nodes of all building D-trees */
    float aabb_min[3], aabb_max[3]; /* cubified, GP:857-859 */
    int32_t iter;
    int32_t is_built;
} ppg_sdtree_info;

int ppg_sdtree_info_get(ppg_ctx *ctx, ppg_sdtree_info *info);

/* S-tree nodes in the reference's numbering (STreeNode, GP:740-845): per node axis, children[2]
   (0,0 for a leaf).  Arrays sized n_stree_nodes. */
int ppg_sdtree_read_stree(ppg_ctx *ctx, int32_t *axis, uint32_t *children /* [n*2] */);

/* Per S-tree node D-tree headers (only meaningful for leaves).  which: 0 = sampling, 1 = building.
   offset = first node of that tree in the node arrays read by ppg_sdtree_read_dtree_nodes. */
int ppg_sdtree_read_dtree_headers(ppg_ctx *ctx, int32_t which, uint64_t *offset, uint32_t *num_nodes,
                                  int32_t *max_depth, float *sum, double *stat_weight);

/* Quadtree nodes (QuadTreeNode, GP:158-371) of all D-trees of one kind, concatenated.
   sums [n*4]: sampling → the float sums; building → the accumulated sums converted to float.
   children [n*4] (uint16, 0 = leaf slot).  fixed_sums may be NULL; building only: the raw 2^-PPG_FIXED_SHIFT
   fixed-point accumulators. */
int ppg_sdtree_read_dtree_nodes(ppg_ctx *ctx, int32_t which, float *sums, uint16_t *children, uint64_t *fixed_sums);

/* learned BSDF sampling fraction state per S-tree node (AdamOptimizer::State, GP:116-124): theta only */
int ppg_sdtree_read_adam(ppg_ctx *ctx, float *theta);

/* Device pointers to the accumulate-only statistics of the *building* SD-tree, for the per-iteration
   all-reduce(sum) of a multi-GPU render.  Both arrays are int64/uint64 fixed-point,
   so an integer sum is exact and order independent.  Valid until the next ppg_begin_iteration. */
int ppg_sdtree_stat_buffers(ppg_ctx *ctx, void **dev_sums, uint64_t *n_sums, void **dev_weights, uint64_t *n_weights);
/* Device pointers to the film accumulators (float RGB sums + sample counts), for the final gather. */
int ppg_film_buffers(ppg_ctx *ctx, void **dev_rgb_sum /* float[h*w*3] */, void **dev_weight /* float[h*w] */);
/* Device pointers to image / squared image of the current ppg_render_passes for the variance reduction:
   call between ppg_render_passes_nostat() and ppg_finish_passes() when sharded. */
int ppg_image_buffers(ppg_ctx *ctx, void **dev_image /* float[h*w*3] */, void **dev_sq_image /* float[h*w*3] */,
                      void **dev_weight /* float[h*w] */);
int ppg_render_passes_nostat(ppg_ctx *ctx, int32_t n_passes); /* GP:1217-1286 only */
int ppg_finish_passes(ppg_ctx *ctx, ppg_pass_stats *stats);   /* GP:1288-1328 only */

/* ------------------------------------------------------------------------------------------------
 * Final iteration: groups of passes.
 *
 * In the final iteration nothing is recorded (GP:2150-2154, !m_isFinalIter) and it holds at least half of a render's samples
 * (GP:1367-1374); the sampler is keyed by (pixel, sample index), so its passes are independent of each other.  A sharded render therefore
 * deals them to the ranks WHOLE — all pixels — instead of splitting every pass by tiles: the tail of a batch of unbounded paths lasts as
 * long as its longest path (a guided path survives a bounce with probability 0.99, GP:2124-2139), a term that does not shrink with the
 * number of ranks when every rank runs every batch, and does when a rank runs every world-th batch.
 * So that the film is the same on one GPU and on N, bit for bit, the float sums are given one association:
 *   - the n passes of a ppg_render_passes() call made with the final flag set (budgetType = spp) form groups of
 *     ppg_final_group_passes(n) = 16 * ceil(n / 1024) consecutive passes (at most 64 groups);
 *   - a group's samples are summed per pixel in sample order, from zero, into the group's partial (image, squared image, weight);
 *   - the call's image / squared image / weights and the iteration's film are the partials added in group order.
 * Sharded: WHO renders a group does not enter the sums, so it is chosen by count.  With at least two groups per rank (groups >= 2 * world)
 * group g is rendered by rank g % world over the whole film; with fewer — a 13-pass final iteration is ONE group, one of 64 passes four —
 * every rank renders every group on its own tiles (ppg_set_shard), or most ranks would idle through half of the render's samples.  Either
 * way, between ppg_render_passes_nostat() and ppg_finish_passes() the host all-reduces (sum, float) the n_floats at `dev` —
 * [film 3 n][film weights n][groups x (image 3 n, squared image 3 n, weights n)], n = pixels; every pixel of a slot is non-zero on one rank
 * only and the film head (the tile-sharded training passes an `automatic` render adds to the same iteration, GP:1400-1405) has disjoint
 * supports, so the sums are exact — and calls ppg_final_partials_commit().  Afterwards image and film are complete on every rank: no
 * ppg_image_buffers / ppg_film_buffers exchange for this iteration.  n_floats = 0: not a sharded final iteration (one rank, or a time
 * budget), exchange the image buffers as usual.  n_floats depends only on the film size and the pass count (4 n + groups * 7 n): a rank
 * that failed can still join the collective, with zeros.
 * ---------------------------------------------------------------------------------------------- */
/* budgetType = seconds in a sharded render: every decision the reference takes by its clock (GP:1259-1262 inside performRenderPasses,
   GP:1434-1514 in renderTime) must come out the same on all ranks, or they would render different numbers of passes and their collectives
   would stop matching.  The stop hook is asked after every batch of passes with this rank's own decision (elapsed whole seconds > budget) and
   returns the one all ranks follow — rank 0's.  The exchange behind it also carries the ranks' status words (host/rccl_reducer.h
   stopDecision: one all-reduce of {decision, status}): a rank that was cancelled asks the hook once more before it leaves the batch loop —
   with local_stop = PPG_STOP_CANCELLED (2) instead of a 0 / 1 decision, so that the host sets its status word even when the cancel did not
   come through the host's own cancel() —, so it meets the others in the exchange they are in, every rank stops there, and the image exchange
   that follows aborts the render on all of them.  The host loop passes the times it measures through a broadcast of rank 0's. */
#define PPG_STOP_CANCELLED 2
typedef int (*ppg_stop_hook)(void *user, int local_stop);
int ppg_set_stop_hook(ppg_ctx *ctx, ppg_stop_hook hook, void *user);
/* The HIP stream (hipStream_t) the context's kernels run on.  A reducer that enqueues its collectives on it needs no host synchronisation
   between an exchange and the library call that consumes its result: the round hook's all-to-all and all-gather are followed, in stream
   order, by the sort / apply / commit kernels. */
int ppg_exchange_stream(ppg_ctx *ctx, void **hip_stream);
int32_t ppg_final_group_passes(int32_t n_passes);
int ppg_final_partials(ppg_ctx *ctx, void **dev /* float[n_floats] */, uint64_t *n_floats);
int ppg_final_partials_commit(ppg_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * Learning the BSDF sampling fraction (bsdfSamplingFractionLoss = "kl" | "var"; AdamOptimizer GP:69-133,
 * DTreeWrapper::optimizeBsdfSamplingFraction GP:672-697).
 *
 * The reference calls optimizeBsdfSamplingFraction() for every committed record under a per-D-tree spin-lock, in the order
 * in which its worker threads happen to arrive: AdamOptimizer::append() accumulates gradient * weight and takes one step
 * whenever the accumulated weight exceeds batchSize = 1 (GP:85-95), the gradient being evaluated at the variable's value at
 * that moment.  That order is not reproducible (not even between two runs of the reference).  This build keeps the rule and
 * fixes the order:
 *   - the n passes of a ppg_render_passes() call (= the training passes of an iteration) are rendered in ROUNDS of
 *     ppg_adam_round_passes(.., n) consecutive passes — at least two rounds per iteration from the second iteration on, so that
 *     the fractions learned from the first half already steer the second (measured on cbox-improved: variance of iterations
 *     2..6 within noise of the literal rule).  All paths of a round are sampled with the fractions in effect
 *     at its start;
 *   - at the end of the round every record that reached DTreeWrapper::record with product > 0 is applied to its D-tree with
 *     exactly the reference's arithmetic (float, gradient at the current variable, append(), step()), D-tree by D-tree, in
 *     ascending order of the 64-bit key
 *         leaf << PPG_ADAM_LEAF_SHIFT | path << PPG_ADAM_CODE_BITS | code
 *     leaf = S-tree node of the D-tree; path = (sample index within the round) * width * height + pixel index;
 *     code = PPG_ADAM_CODE_VERTEX + i for path vertex i (Vertex::commit, GP:2150-2154), min(rRec.depth, PPG_ADAM_CODE_VERTEX - 1)
 *     for the direct-light vertex of next-event estimation (GP:1994-2010).  Keys are unique, so the order is total: a valid
 *     serialisation of the reference's critical sections that does not depend on wave scheduling, thread or GPU count.
 * What the fixed order costs (measured against the reference's own render logs): under the reference's rule the
 * variable follows the records of the last ~100 paths — of the image region being rendered —, which a value frozen for a round cannot; the
 * variance estimate of the EARLY iterations is up to twice the reference's on spaceship-improved (equal from iteration 6, 64 passes, on),
 * unaffected from iteration 2 on on kitchen-improved.  The oracle implements the literal rule too (PPGO_ADAM_SEQUENTIAL) and reproduces the
 * reference's logs with it to 1 - 4 %.
 * AdamOptimizer::State (incl. the partial batch) survives rounds, iterations and STree subdivision (GP:890) as in the reference.
 *
 * STRAGGLERS (round 6).  With maxDepth = -1 a guided path survives Russian roulette with probability 0.99 (GP:2124-2139): one path in
 * ten thousand runs for hundreds of bounces, and a round that waits for it before its records may be applied idles the GPU for
 * milliseconds — on every GPU of a sharded render alike.  The reference applies a path's records when that path happens to finish
 * (Vertex::commit at the end of Li, GP:2150-2154, under the lock of GP:719-737): records of long paths arrive late there too.  The rule
 * here, deterministic and independent of wave scheduling, batch size, thread and GPU count:
 *   - in a round whose record positions are known in advance (spatialFilter != box, no kick-start luminaire sampling in effect) of a
 *     render with maxDepth = -1, a path whose FINAL rRec.depth (the value Li adds to avgPathLength, GP:2147-2148) exceeds
 *     PPG_ADAM_DEFER_DEPTH is a straggler;
 *   - a straggler's optimiser records are applied with the NEXT round of the same ppg_render_passes() call — per D-tree after that
 *     round's own records: their keys carry PPG_ADAM_DEFER_PATH_BIT in the path field, so "ascending key order" says exactly that —,
 *     and those of the call's last round in a round of their own at the end of the call (the sharded render's round hook is called for
 *     it on every rank, with or without records);
 *   - nothing else moves: the straggler's splats reach the building tree before ppg_build_sdtree (integer sums: the same bits whenever
 *     they land), its radiance reaches image and film in its sample's place in the sums.
 * On KITCHEN one path in 10^4 is a straggler (16-pass round at 1280x720: 1 481 of 14.7 M); the product finishes them on a side stream
 * beside the next round's paths (ppg_hip.hip "Stragglers").  The oracle implements the same rule.
 * Limits: width * height * sppPerPass <= 2^(PPG_ADAM_PATH_BITS - 1) and at most 2^24 S-tree nodes while a loss is set.
 * ---------------------------------------------------------------------------------------------- */
#define PPG_ADAM_ROUND_MAX_PASSES 16
#define PPG_ADAM_ROUND_MAX_PATHS (1u << 24)
#define PPG_ADAM_CODE_BITS 13
#define PPG_ADAM_PATH_BITS 27
#define PPG_ADAM_LEAF_SHIFT (PPG_ADAM_CODE_BITS + PPG_ADAM_PATH_BITS)
#define PPG_ADAM_CODE_VERTEX 4096
#define PPG_ADAM_DEFER_DEPTH 64
#define PPG_ADAM_DEFER_PATH_BIT (1u << (PPG_ADAM_PATH_BITS - 1))   /* in the path field of the key: a straggler's record, applied one round late */
/* passes per round for a call that renders n_passes: the largest power of two <= min(PPG_ADAM_ROUND_MAX_PASSES, n_passes / 2)
   whose paths (passes * sppPerPass * pixels of the whole image, not of a shard) stay within PPG_ADAM_ROUND_MAX_PATHS; at least 1 */
int32_t ppg_adam_round_passes(int32_t spp_per_pass, int32_t width, int32_t height, int32_t n_passes);

/* Rounds by image REGION — an extension, off by default (regions = 0), for renders whose EARLY iterations must follow the reference's.
   Under the reference's rule the optimiser's variable follows the records of the last ~100 paths, i.e. of the image region its 16 threads are
   rendering; a variable frozen for a whole pass cannot, and the variance estimate of iterations 1 - 4 is up to twice the reference's
   (above).  With regions = R >= 2, in every ppg_render_passes() call of at most PPG_ADAM_REGION_MAX_PASSES passes that is rendered in rounds,
   a round is ONE pass over ONE of R groups of 32x32-pixel blocks — consecutive in the spiral order in which the reference's scheduler hands
   blocks out (ppg_spiral_block_ranks, include/ppg_detmath.h; group of block b = rank(b) * R / blocks) — and the optimiser is applied after
   every group: R rounds per pass.  Measured against the reference's own log of spaceship-improved: variance estimate
   of iterations 1 - 4 within 2 - 13 % with R = 16 (2x with R = 0).  The price: R rounds of pixels / R paths each instead of one round of
   1 - 8 passes — on one MI355X the early iterations of KITCHEN at 1280x720 take several times longer.  Keys, stragglers,
   sharding (a rank renders its tiles of a group; R hook calls per pass on every rank) as for any round.  R is clamped to the number of blocks. */
#define PPG_ADAM_REGION_MAX_PASSES 16
int ppg_set_adam_regions(ppg_ctx *ctx, int32_t regions);

typedef struct ppg_adam_record { /* one deferred optimizeBsdfSamplingFraction() call: DTreeRecord's fields it reads (GP:562-568) */
    uint64_t key;
    float product, wo_pdf, bsdf_pdf, dtree_pdf, statistical_weight, _pad;
} ppg_adam_record;             /* 32 bytes */

/* Round hook for multi-GPU rendering: called at the end of every round after this rank's records were collected and before
   they are applied.  The hook gathers the records of all ranks (ppg_adam_records → exchange → ppg_adam_records_replace) so that
   every rank applies the identical sequence and the learned fractions stay bit-identical across ranks — and equal to a
   single-GPU render, because the key order does not depend on the sharding.  Return non-zero to abort. */
typedef int (*ppg_pass_hook)(void *user);
int ppg_set_pass_hook(ppg_ctx *ctx, ppg_pass_hook hook, void *user);
/* Valid inside the hook only.  dev_records: this rank's records (device memory, unspecified order). */
int ppg_adam_records(ppg_ctx *ctx, void **dev_records /* ppg_adam_record[n] */, uint64_t *n);
/* Valid inside the hook only: the records to apply instead (device pointer, copied). */
int ppg_adam_records_replace(ppg_ctx *ctx, const void *dev_records, uint64_t n);

/* Sharded optimiser: ONE OWNER PER D-TREE.  The reference serialises the Adam steps per D-tree (its spin-lock, GP:719-737), so the
   D-trees can be dealt to the ranks: S-tree node `leaf` belongs to rank leaf / segment, segment = ceil(n_stree_nodes / world) —
   contiguous node ranges, so with the records in key order every owner's records are one contiguous run.  The hook is then called
   TWICE per round:
     phase 0  before the records are applied: ppg_adam_records_by_owner → all-to-all (every rank keeps 1 / world of the records
              instead of gathering all of them) → ppg_adam_records_replace with what this rank owns; it then sorts and applies only those;
     phase 1  after they were applied (only if phase 0 called ppg_adam_records_by_owner): ppg_adam_state → all-gather of the
              owners' segments, in place → ppg_adam_state_commit.
   Same records in the same key order at the owner as in the union ⇒ the same bits as the gather-everything scheme and as one GPU. */
int ppg_hook_phase(ppg_ctx *ctx, int32_t *phase);
/* phase 0: this rank's records in key order (device memory) and how many of them each of the `world` owners gets */
int ppg_adam_records_by_owner(ppg_ctx *ctx, int32_t world, void **dev_records /* ppg_adam_record[sum(counts)] */, uint64_t *counts /* [world] */);
/* phase 1: the optimiser's state of every S-tree node, 24 bytes each (theta, iter, m, v, batchGradient, batchAccumulation; GP:116-124),
   packed into a device array of world * segment entries; rank r's segment [r * segment, (r + 1) * segment) holds what it computed */
int ppg_adam_state(ppg_ctx *ctx, int32_t world, void **dev_state, uint64_t *segment);
int ppg_adam_state_commit(ppg_ctx *ctx); /* take the (gathered) array back into the tree */

/* Batched queries against the current *sampling* SD-tree (what Li does per vertex):
   pdf  = DTreeWrapper::pdf(dir) of the leaf containing p        (GP:623-625, 897-905)
   dirs = DTreeWrapper::sample(sampler) with the build's sampler keyed by (seed, i)  (GP:619-621) */
int ppg_query_pdf(ppg_ctx *ctx, uint32_t n, const float *positions, const float *dirs, float *pdf_out);
int ppg_query_sample(ppg_ctx *ctx, uint32_t n, const float *positions, uint64_t seed, float *dirs_out);

/* Kernel timing of the last ppg_render_passes, measured with HIP events on the stream the kernels
   run on: names[i] → accumulated milliseconds and launch count.  Used by bench.py's roofline block. */
typedef struct ppg_kernel_time { const char *name; double ms; uint64_t launches; uint64_t units; } ppg_kernel_time;
int ppg_kernel_times(ppg_ctx *ctx, ppg_kernel_time *out, uint32_t cap, uint32_t *n);
int ppg_enable_kernel_timing(ppg_ctx *ctx, int32_t enable);

#define PPG_FIXED_SHIFT 24 /* building sums / weights are accumulated as round(x * 2^24) in uint64 */

#ifdef __cplusplus
}
#endif
#endif /* PPG_H */
