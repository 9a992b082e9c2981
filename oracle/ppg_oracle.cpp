/*
 * ppg_oracle.cpp — CPU restatement of the reference GuidedPathTracer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libppg_oracle.so, and only as the
 * checker / the timed CPU baseline.  libppg_hip.so never links or calls into it.
 *
 * What it follows (GP = /root/reference/mitsuba/src/integrators/path/guided_path.cpp):
 *   AdamOptimizer GP:69-133 · QuadTreeNode GP:158-371 · DTree GP:374-560 · DTreeWrapper GP:570-738 ·
 *   STreeNode GP:740-845 · STree GP:848-1007 · GuidedPathTracer GP:1012-2419 (surface branch; the medium
 *   branch GP:1803-1893 is unreachable in every config) — plus the callees listed in
 *   diffuse.cpp:110-150, warp.cpp:43-52/81-102, area.cpp:104-109,
 *   perspective.cpp:271-298, skdtree.cpp:112-142 (ray epsilon), skdtree.h:343-430, util.cpp:592-608;
 *   for next-event estimation scene.cpp:619-679/876-897/949-952, area.cpp:158-183, shape.cpp:102-126,
 *   trimesh.cpp:388-423, triangle.cpp:24-67, pmf.h:101-188, records.inl:160-178.
 *   The code is written recursively / object-per-node like the reference, on purpose: the HIP product
 *   uses flat arrays and wavefront kernels, so the two share no implementation.
 *
 * PARITY PIN STATUS — read before trusting this file:
 *   The reference cannot be built in this environment (mitsuba.h:24 needs boost; scons, xerces-c,
 *   OpenEXR, Eigen are absent) and ships no tests for this path, so there is no
 *   bit-level pin: BIT-LEVEL PARITY WITH THE REFERENCE IS UNPINNED.  What pins this restatement is
 *     (1) the known answers of SURVEY.md §8(c) (85-node/depth-4 first reset, pdf 1/4π, refine →
 *         512 leaves, 10 KL Adam records → 0.512494385)         tests/test_oracle_known_answers.py
 *     (2) the reference's own shipped render logs and pixels (tests/golden/ref_logs.json,
 *         ref_cbox_images.npz, mined by tools/make_ref_fixtures.py): iteration schedules exactly,
 *         iteration-0/1 SD-tree statistics, average path length, variance sequence and the CBOX
 *         image statistically, for the default and the "improved" configuration
 *                                                               tests/test_oracle_reference_pins.py
 *     (3) next-event estimation (no reference log uses it): the analytic irradiance under a small lamp
 *         (within 1 %) and agreement of nee = never / kickstart / always on CBOX
 *                                                               tests/test_oracle_known_answers.py
 *
 * Deliberate, documented deviations (DESIGN.md §"numerical contract"):
 *   - sampler: counter-based (include/ppg_rng.h) instead of per-thread SFMT streams;
 *   - libm sincos/atan2/exp/pow → include/ppg_detmath.h (bit-reproducible on CPU and GPU);
 *   - ray/triangle test: the reference's own TriAccel projection test (triaccel.h), closest hit by (t, primitive
 *     index) over all triangles / an own BVH instead of the kd-tree's traversal order (ties only);
 *   - SD-tree statistics are accumulated in 2^-24 fixed point (order independent) when
 *     acc_mode = FIXED (default); acc_mode = FLOAT is the reference's sequential float adds
 *     (GP:59-62) and is what pin (1) exercises;
 *   - Adam: adam_mode = SEQUENTIAL is GP:672-697 literally (single thread, every record applied the moment its path
 *     commits it); adam_mode = ROUND (default, the product's rule) applies the same calls with the same arithmetic at the end
 *     of every round of render passes in a fixed key order (include/ppg.h "Learning the BSDF sampling fraction").
 */
#include "../include/ppg.h"
#include "../include/ppg_detmath.h"
#include "../include/ppg_rng.h"
#include "ppg_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stack>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef float Float;

// ------------------------------------------------------------------------------------------------
// Small math types (stand for mitsuba/core/{point,vector,frame}.h; arithmetic order as cited)
// ------------------------------------------------------------------------------------------------
struct Point2 {
    Float x, y;
    Float &operator[](int i) { return i == 0 ? x : y; }
    Float operator[](int i) const { return i == 0 ? x : y; }
};

struct Vec {
    Float x, y, z;
    Vec() : x(0), y(0), z(0) {}
    Vec(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    explicit Vec(Float v) : x(v), y(v), z(v) {}
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Vec operator+(const Vec &o) const { return Vec(x + o.x, y + o.y, z + o.z); }
    Vec operator-(const Vec &o) const { return Vec(x - o.x, y - o.y, z - o.z); }
    Vec operator*(Float f) const { return Vec(x * f, y * f, z * f); }
    Vec operator-() const { return Vec(-x, -y, -z); }
    // vector.h:535-542: division multiplies by the reciprocal
    Vec operator/(Float f) const { Float r = 1.0f / f; return Vec(x * r, y * r, z * r); }
};
typedef Vec Point;
typedef Vec Spectrum;  // SPECTRUM_SAMPLES = 3

inline Float dot(const Vec &a, const Vec &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec cross(const Vec &a, const Vec &b) {
    return Vec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Float length(const Vec &v) { return std::sqrt(dot(v, v)); }
inline Vec normalize(const Vec &v) { return v / length(v); }  // vector.h:625-627
inline Vec mul(const Vec &a, const Vec &b) { return Vec(a.x * b.x, a.y * b.y, a.z * b.z); }
inline bool isZero(const Spectrum &s) { return s.x == 0.0f && s.y == 0.0f && s.z == 0.0f; }  // spectrum.h:577
inline bool isValid(const Spectrum &s) {  // spectrum.h:467-472
    for (int i = 0; i < 3; ++i)
        if (!ppg_isfinite(s[i]) || s[i] < 0.0f) return false;
    return true;
}
inline Float average(const Spectrum &s) {  // spectrum.h:481-486
    Float r = 0.0f;
    for (int i = 0; i < 3; ++i) r += s[i];
    return r * (1.0f / 3);
}
inline Float specMax(const Spectrum &s) { return ppg_max(ppg_max(s.x, s.y), s.z); }  // spectrum.h:543-548
inline Float luminance(const Spectrum &s) {  // spectrum.h:725-727
    return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f;
}

struct Frame {
    Vec s, t, n;
    Vec toLocal(const Vec &v) const { return Vec(dot(v, s), dot(v, t), dot(v, n)); }
    Vec toWorld(const Vec &v) const { return s * v.x + t * v.y + n * v.z; }
};

struct AABB {
    Point min, max;
    Vec getExtents() const { return max - min; }
    Point clip(const Point &p) const {  // aabb.h clip(): clamp to the box
        Point r = p;
        for (int i = 0; i < 3; ++i) r[i] = ppg_min(ppg_max(r[i], min[i]), max[i]);
        return r;
    }
};

// Sampler: next1D / next2D of samplers/independent.cpp:95-103 over the counter-based generator.
struct Sampler {
    uint32_t key, dim;
    Float next1D() { return ppg_rand(key, dim++); }
    Point2 next2D() {
        Float a = ppg_rand(key, dim++);
        Float b = ppg_rand(key, dim++);
        return Point2{a, b};
    }
};

enum ENee { ENever, EKickstart, EAlways };
enum ESampleCombination { EDiscard, EDiscardWithAutomaticBudget, EInverseVariance };
enum ESpatialFilter { ESNearest, EStochasticBox, ESBox };
enum EDirectionalFilter { EDNearest, EDBox };
enum ELoss { ENone, EKL, EVariance };
enum EBudget { ESpp, ESeconds };

struct DTreeWrapper;
// One call of optimizeBsdfSamplingFraction deferred to the end of the round (adam_mode = ROUND): its inputs and its place in
// the canonical order (S-tree leaf, path of the round, record code) — see include/ppg.h "Learning the BSDF sampling fraction".
struct AdamRecord {
    DTreeWrapper *dTree;
    uint32_t path, code;
    float product, woPdf, bsdfPdf, dTreePdf, weight;
};
struct Modes {
    int acc = PPGO_ACC_FIXED;
    int adam = PPGO_ADAM_ROUND;
    // ROUND mode, per worker thread: where the records of the path being committed go, and that path's / vertex's identity
    std::vector<AdamRecord> *sink = nullptr;
    uint32_t path = 0, code = 0;
};

inline Float logistic(Float x) { return 1 / (1 + ppg_exp(-x)); }  // GP:64-66

// Work counters for the algorithmic-bytes accounting of SURVEY.md §8(d) (levels visited per operation).
enum { CNT_STREE_LEVELS, CNT_STREE_LOOKUPS, CNT_DSAMPLE_LEVELS, CNT_DSAMPLE_CALLS, CNT_DPDF_LEVELS, CNT_DPDF_CALLS, CNT_DRECORD_LEVELS, CNT_DRECORD_CALLS, CNT_N };
thread_local uint64_t t_cnt[CNT_N];

// ------------------------------------------------------------------------------------------------
// AdamOptimizer  GP:69-133
// ------------------------------------------------------------------------------------------------
class AdamOptimizer {
public:
    explicit AdamOptimizer(Float learningRate, int batchSize = 1, Float epsilon = 1e-08f, Float beta1 = 0.9f,
                           Float beta2 = 0.999f) {
        m_hparams = {learningRate, batchSize, epsilon, beta1, beta2};
    }

    void append(Float gradient, Float statisticalWeight) {  // GP:85-95
        m_state.batchGradient += gradient * statisticalWeight;
        m_state.batchAccumulation += statisticalWeight;
        if (m_state.batchAccumulation > m_hparams.batchSize) {
            step(m_state.batchGradient / m_state.batchAccumulation);
            m_state.batchGradient = 0;
            m_state.batchAccumulation = 0;
        }
    }

    void step(Float gradient) {  // GP:97-109
        ++m_state.iter;
        // GP:100: std::pow(float, int) promotes to double, the expression is evaluated in double and rounded once (ppg_detmath.h)
        Float actualLearningRate = ppg_adam_learning_rate(m_hparams.learningRate, m_hparams.beta1, m_hparams.beta2, m_state.iter);
        m_state.firstMoment = m_hparams.beta1 * m_state.firstMoment + (1 - m_hparams.beta1) * gradient;
        m_state.secondMoment = m_hparams.beta2 * m_state.secondMoment + (1 - m_hparams.beta2) * gradient * gradient;
        m_state.variable -= actualLearningRate * m_state.firstMoment / (std::sqrt(m_state.secondMoment) + m_hparams.epsilon);
        m_state.variable = ppg_min(ppg_max(m_state.variable, -20.0f), 20.0f);
    }

    Float variable() const { return m_state.variable; }

    // the complete state as six 32-bit words (theta, iter, m, v, batchGradient, batchAccumulation): what the owner of a D-tree hands to the
    // other ranks of a sharded render (include/ppg.h "Sharded optimiser")
    void exportState(uint32_t out[6]) const {
        memcpy(out + 0, &m_state.variable, 4); memcpy(out + 1, &m_state.iter, 4); memcpy(out + 2, &m_state.firstMoment, 4);
        memcpy(out + 3, &m_state.secondMoment, 4); memcpy(out + 4, &m_state.batchGradient, 4); memcpy(out + 5, &m_state.batchAccumulation, 4);
    }
    void importState(const uint32_t in[6]) {
        memcpy(&m_state.variable, in + 0, 4); memcpy(&m_state.iter, in + 1, 4); memcpy(&m_state.firstMoment, in + 2, 4);
        memcpy(&m_state.secondMoment, in + 3, 4); memcpy(&m_state.batchGradient, in + 4, 4); memcpy(&m_state.batchAccumulation, in + 5, 4);
    }

private:
    struct State {
        int iter = 0;
        Float firstMoment = 0;
        Float secondMoment = 0;
        Float variable = 0;
        Float batchAccumulation = 0;
        Float batchGradient = 0;
    } m_state;
    struct Hyperparameters {
        Float learningRate;
        int batchSize;
        Float epsilon;
        Float beta1;
        Float beta2;
    } m_hparams;
};

// ------------------------------------------------------------------------------------------------
// QuadTreeNode  GP:158-371
// ------------------------------------------------------------------------------------------------
class QuadTreeNode {
public:
    QuadTreeNode() {
        for (int i = 0; i < 4; ++i) { m_sum[i] = 0; m_acc[i] = 0; m_children[i] = 0; }
    }
    void setSum(int index, Float val) { m_sum[index] = val; }
    Float sum(int index) const { return m_sum[index]; }
    void setChild(int idx, uint16_t val) { m_children[idx] = val; }
    uint16_t child(int idx) const { return m_children[idx]; }
    void setSum(Float val) { for (int i = 0; i < 4; ++i) { m_sum[i] = val; m_acc[i] = 0; } }
    bool isLeaf(int index) const { return child(index) == 0; }
    uint64_t acc(int index) const { return m_acc[index]; }
    void setAcc(int index, uint64_t v) { m_acc[index] = v; }

    int childIndex(Point2 &p) const {  // GP:205-217
        int res = 0;
        for (int i = 0; i < 2; ++i) {
            if (p[i] < 0.5f) {
                p[i] *= 2;
            } else {
                p[i] = (p[i] - 0.5f) * 2;
                res |= 1 << i;
            }
        }
        return res;
    }

    Float pdf(Point2 &p, const std::vector<QuadTreeNode> &nodes) const {  // GP:232-245
        ++t_cnt[CNT_DPDF_LEVELS];
        const int index = childIndex(p);
        if (!(sum(index) > 0)) return 0;
        const Float factor = 4 * sum(index) / (sum(0) + sum(1) + sum(2) + sum(3));
        if (isLeaf(index)) return factor;
        return factor * nodes[child(index)].pdf(p, nodes);
    }

    int depthAt(Point2 &p, const std::vector<QuadTreeNode> &nodes) const {  // GP:247-255
        const int index = childIndex(p);
        if (isLeaf(index)) return 1;
        return 1 + nodes[child(index)].depthAt(p, nodes);
    }

    Point2 sample(Sampler *sampler, const std::vector<QuadTreeNode> &nodes) const {  // GP:257-301
        ++t_cnt[CNT_DSAMPLE_LEVELS];
        int index = 0;
        Float topLeft = sum(0);
        Float topRight = sum(1);
        Float partial = topLeft + sum(2);
        Float total = partial + topRight + sum(3);
        if (!(total > 0.0f)) return sampler->next2D();

        Float boundary = partial / total;
        Point2 origin = Point2{0.0f, 0.0f};
        Float sample = sampler->next1D();

        if (sample < boundary) {
            sample /= boundary;
            boundary = topLeft / partial;
        } else {
            partial = total - partial;
            origin.x = 0.5f;
            sample = (sample - boundary) / (1.0f - boundary);
            boundary = topRight / partial;
            index |= 1 << 0;
        }
        if (sample < boundary) {
            sample /= boundary;
        } else {
            origin.y = 0.5f;
            sample = (sample - boundary) / (1.0f - boundary);
            index |= 1 << 1;
        }
        Point2 r = isLeaf(index) ? sampler->next2D() : nodes[child(index)].sample(sampler, nodes);
        return Point2{origin.x + 0.5f * r.x, origin.y + 0.5f * r.y};
    }

    void add(int index, Float value, int accMode) {  // addToAtomicFloat GP:59-62 / fixed-point variant
        if (accMode == PPGO_ACC_FLOAT) {
            m_sum[index] += value;
        } else {
            __atomic_fetch_add(&m_acc[index], ppg_to_fixed(value), __ATOMIC_RELAXED);
        }
    }

    void record(Point2 &p, Float irradiance, std::vector<QuadTreeNode> &nodes, int accMode) {  // GP:303-312
        ++t_cnt[CNT_DRECORD_LEVELS];
        int index = childIndex(p);
        if (isLeaf(index)) add(index, irradiance, accMode);
        else nodes[child(index)].record(p, irradiance, nodes, accMode);
    }

    static Float computeOverlappingArea(const Point2 &min1, const Point2 &max1, const Point2 &min2, const Point2 &max2) {
        Float lengths[2];  // GP:314-320
        for (int i = 0; i < 2; ++i)
            lengths[i] = ppg_max(ppg_min(max1[i], max2[i]) - ppg_max(min1[i], min2[i]), 0.0f);
        return lengths[0] * lengths[1];
    }

    void record(const Point2 &origin, Float size, Point2 nodeOrigin, Float nodeSize, Float value,
                std::vector<QuadTreeNode> &nodes, int accMode) {  // GP:322-338
        Float childSize = nodeSize / 2;
        for (int i = 0; i < 4; ++i) {
            Point2 childOrigin = nodeOrigin;
            if (i & 1) childOrigin.x += childSize;
            if (i & 2) childOrigin.y += childSize;
            Float w = computeOverlappingArea(origin, Point2{origin.x + size, origin.y + size}, childOrigin,
                                             Point2{childOrigin.x + childSize, childOrigin.y + childSize});
            if (w > 0.0f) {
                if (isLeaf(i)) add(i, value * w, accMode);
                else nodes[child(i)].record(origin, size, childOrigin, childSize, value, nodes, accMode);
            }
        }
    }

    // FIXED mode: turn the integer accumulators of the leaf slots into the float sums build() works on.
    void resolveFixed() {
        for (int i = 0; i < 4; ++i)
            if (isLeaf(i)) m_sum[i] = ppg_from_fixed(m_acc[i]);
    }

    void build(std::vector<QuadTreeNode> &nodes) {  // GP:346-366
        for (int i = 0; i < 4; ++i) {
            if (isLeaf(i)) continue;
            QuadTreeNode &c = nodes[child(i)];
            c.build(nodes);
            Float sum = 0;
            for (int j = 0; j < 4; ++j) sum += c.sum(j);
            setSum(i, sum);
        }
    }

private:
    Float m_sum[4];
    uint64_t m_acc[4];
    uint16_t m_children[4];
};

// ------------------------------------------------------------------------------------------------
// DTree  GP:374-560
// ------------------------------------------------------------------------------------------------
class DTree {
public:
    DTree() {
        m_sum = 0; m_statisticalWeight = 0; m_statAcc = 0; m_maxDepth = 0;
        m_nodes.emplace_back();
        m_nodes.front().setSum(0.0f);
    }
    const QuadTreeNode &node(size_t i) const { return m_nodes[i]; }

    Float mean() const {  // GP:387-393
        if (m_statisticalWeight == 0) return 0;
        const Float factor = 1 / (PPG_PI_F * 4 * m_statisticalWeight);
        return factor * m_sum;
    }

    void recordIrradiance(Point2 p, Float irradiance, Float statisticalWeight, EDirectionalFilter directionalFilter,
                          int accMode) {  // GP:395-413
        if (ppg_isfinite(statisticalWeight) && statisticalWeight > 0) {
            if (accMode == PPGO_ACC_FLOAT) m_statisticalWeight += statisticalWeight;
            else __atomic_fetch_add(&m_statAcc, ppg_to_fixed(statisticalWeight), __ATOMIC_RELAXED);

            if (ppg_isfinite(irradiance) && irradiance > 0) {
                ++t_cnt[CNT_DRECORD_CALLS];
                if (directionalFilter == EDNearest) {
                    m_nodes[0].record(p, irradiance * statisticalWeight, m_nodes, accMode);
                } else {
                    int depth = depthAt(p);
                    Float size = ppg_exp2i(-depth);  // std::pow(0.5f, depth), exact
                    Point2 origin = p;
                    origin.x -= size / 2;
                    origin.y -= size / 2;
                    m_nodes[0].record(origin, size, Point2{0.0f, 0.0f}, 1.0f,
                                      irradiance * statisticalWeight / (size * size), m_nodes, accMode);
                }
            }
        }
    }

    Float pdf(Point2 p) const {  // GP:415-421
        ++t_cnt[CNT_DPDF_CALLS];
        if (!(mean() > 0)) return 1 / (4 * PPG_PI_F);
        return m_nodes[0].pdf(p, m_nodes) / (4 * PPG_PI_F);
    }
    int depthAt(Point2 p) const { return m_nodes[0].depthAt(p, m_nodes); }
    int depth() const { return m_maxDepth; }

    Point2 sample(Sampler *sampler) const {  // GP:431-442
        ++t_cnt[CNT_DSAMPLE_CALLS];
        if (!(mean() > 0)) return sampler->next2D();
        Point2 res = m_nodes[0].sample(sampler, m_nodes);
        res.x = ppg_min(ppg_max(res.x, 0.0f), 1.0f);  // math::clamp
        res.y = ppg_min(ppg_max(res.y, 0.0f), 1.0f);
        return res;
    }

    size_t numNodes() const { return m_nodes.size(); }
    Float statisticalWeight() const { return m_statisticalWeight; }
    void setStatisticalWeight(Float w) { m_statisticalWeight = w; }
    Float sumValue() const { return m_sum; }
    uint64_t statAcc() const { return m_statAcc; }

    void reset(const DTree &previousDTree, int newMaxDepth, Float subdivisionThreshold) {  // GP:456-514
        m_sum = 0; m_statisticalWeight = 0; m_statAcc = 0;
        m_maxDepth = 0;
        m_nodes.clear();
        m_nodes.emplace_back();

        struct StackNode {
            size_t nodeIndex;
            size_t otherNodeIndex;
            const DTree *otherDTree;
            int depth;
        };
        std::stack<StackNode> nodeIndices;
        nodeIndices.push({0, 0, &previousDTree, 1});
        const Float total = previousDTree.m_sum;

        while (!nodeIndices.empty()) {
            StackNode sNode = nodeIndices.top();
            nodeIndices.pop();
            m_maxDepth = std::max(m_maxDepth, sNode.depth);

            for (int i = 0; i < 4; ++i) {
                const QuadTreeNode &otherNode = sNode.otherDTree->m_nodes[sNode.otherNodeIndex];
                const Float fraction = total > 0 ? (otherNode.sum(i) / total) : ppg_exp2i(-2 * sNode.depth);  // pow(0.25f, depth)
                if (sNode.depth < newMaxDepth && fraction > subdivisionThreshold) {
                    if (!otherNode.isLeaf(i)) {
                        nodeIndices.push({m_nodes.size(), otherNode.child(i), &previousDTree, sNode.depth + 1});
                    } else {
                        nodeIndices.push({m_nodes.size(), m_nodes.size(), this, sNode.depth + 1});
                    }
                    m_nodes[sNode.nodeIndex].setChild(i, static_cast<uint16_t>(m_nodes.size()));
                    const Float seed = sNode.otherDTree->m_nodes[sNode.otherNodeIndex].sum(i) / 4;
                    m_nodes.emplace_back();
                    m_nodes.back().setSum(seed);
                    if (m_nodes.size() > std::numeric_limits<uint16_t>::max()) {
                        nodeIndices = std::stack<StackNode>();
                        break;
                    }
                }
            }
        }
        for (auto &node : m_nodes) node.setSum(0);
    }

    void build(int accMode) {  // GP:520-533
        if (accMode == PPGO_ACC_FIXED) {
            for (auto &n : m_nodes) n.resolveFixed();
            m_statisticalWeight = ppg_from_fixed(m_statAcc);
        }
        auto &root = m_nodes[0];
        root.build(m_nodes);
        Float sum = 0;
        for (int i = 0; i < 4; ++i) sum += root.sum(i);
        m_sum = sum;
    }

    size_t approxMemoryFootprint() const { return m_nodes.capacity() * 24 + 40; }  // GP:516-518 with the reference's sizeof

    // multi-rank reduction support (tests of the sharded driver): raw accumulators out / in
    void exportAcc(std::vector<uint64_t> &sums, std::vector<uint64_t> &weights) const {
        for (auto &n : m_nodes) for (int i = 0; i < 4; ++i) sums.push_back(n.acc(i));
        weights.push_back(m_statAcc);
    }
    void importAcc(const uint64_t *&sums, const uint64_t *&weights) {
        for (auto &n : m_nodes) for (int i = 0; i < 4; ++i) n.setAcc(i, *sums++);
        m_statAcc = *weights++;
    }

private:
    std::vector<QuadTreeNode> m_nodes;
    Float m_sum;
    Float m_statisticalWeight;
    uint64_t m_statAcc;
    int m_maxDepth;
};

// ------------------------------------------------------------------------------------------------
// DTreeRecord GP:562-568, DTreeWrapper GP:570-738
// ------------------------------------------------------------------------------------------------
struct DTreeRecord {
    Vec d;
    Float radiance, product;
    Float woPdf, bsdfPdf, dTreePdf;
    Float statisticalWeight;
    bool isDelta;
};

struct DTreeWrapper {
public:
    void record(const DTreeRecord &rec, EDirectionalFilter directionalFilter, ELoss bsdfSamplingFractionLoss,
                const Modes &modes) {  // GP:575-584
        if (!rec.isDelta) {
            Float irradiance = rec.radiance / rec.woPdf;
            building.recordIrradiance(dirToCanonical(rec.d), irradiance, rec.statisticalWeight, directionalFilter, modes.acc);
        }
        if (bsdfSamplingFractionLoss != ENone && rec.product > 0) {
            optimizeBsdfSamplingFraction(rec, bsdfSamplingFractionLoss == EKL ? 1.0f : 2.0f, modes);
        }
    }

    static Vec canonicalToDir(Point2 p) {  // GP:586-595
        const Float cosTheta = 2 * p.x - 1;
        const Float phi = 2 * PPG_PI_F * p.y;
        const Float sinTheta = std::sqrt(1 - cosTheta * cosTheta);
        Float sinPhi, cosPhi;
        ppg_sincos(phi, &sinPhi, &cosPhi);
        return Vec(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    }

    static Point2 dirToCanonical(const Vec &d) {  // GP:597-608
        if (!ppg_isfinite(d.x) || !ppg_isfinite(d.y) || !ppg_isfinite(d.z)) return Point2{0, 0};
        const Float cosTheta = ppg_min(ppg_max(d.z, -1.0f), 1.0f);
        Float phi = ppg_atan2(d.y, d.x);
        while (phi < 0) phi = (Float)((double)phi + 2.0 * (double)PPG_PI_F);  // `phi += 2.0 * M_PI` with float M_PI
        return Point2{(cosTheta + 1) / 2, phi / (2 * PPG_PI_F)};
    }

    void build(const Modes &modes) {  // GP:610-613
        building.build(modes.acc);
        sampling = building;
    }
    void reset(int maxDepth, Float subdivisionThreshold) { building.reset(sampling, maxDepth, subdivisionThreshold); }  // GP:615-617
    Vec sample(Sampler *sampler) const { return canonicalToDir(sampling.sample(sampler)); }  // GP:619-621
    Float pdf(const Vec &dir) const { return sampling.pdf(dirToCanonical(dir)); }              // GP:623-625
    int depth() const { return sampling.depth(); }
    size_t numNodes() const { return sampling.numNodes(); }
    Float meanRadiance() const { return sampling.mean(); }
    Float statisticalWeight() const { return sampling.statisticalWeight(); }
    Float statisticalWeightBuilding() const { return building.statisticalWeight(); }
    void setStatisticalWeightBuilding(Float w) { building.setStatisticalWeight(w); }
    size_t approxMemoryFootprint() const { return building.approxMemoryFootprint() + sampling.approxMemoryFootprint(); }

    Float bsdfSamplingFraction(Float variable) const { return logistic(variable); }  // GP:659-661
    Float dBsdfSamplingFraction_dVariable(Float variable) const {                    // GP:663-666
        Float fraction = bsdfSamplingFraction(variable);
        return fraction * (1 - fraction);
    }
    Float bsdfSamplingFraction() const { return bsdfSamplingFraction(bsdfSamplingFractionOptimizer.variable()); }

    void optimizeBsdfSamplingFraction(const DTreeRecord &rec, Float ratioPower, const Modes &modes) {  // GP:672-697
        if (modes.adam != PPGO_ADAM_SEQUENTIAL) {  // deferred to the end of the round, applied in canonical order by applyAdamRound()
            modes.sink->push_back(AdamRecord{this, modes.path, modes.code, rec.product, rec.woPdf, rec.bsdfPdf, rec.dTreePdf, rec.statisticalWeight});
            return;
        }
        optimizeBsdfSamplingFraction(rec.product, rec.woPdf, rec.bsdfPdf, rec.dTreePdf, rec.statisticalWeight, ratioPower);
    }
    void optimizeBsdfSamplingFraction(Float product, Float woPdf, Float bsdfPdf, Float dTreePdf, Float statisticalWeight, Float ratioPower) {
        Float variable = bsdfSamplingFractionOptimizer.variable();
        Float samplingFraction = bsdfSamplingFraction(variable);
        Float mixPdf = samplingFraction * bsdfPdf + (1 - samplingFraction) * dTreePdf;
        Float r = product / mixPdf;
        Float ratio = ratioPower == 1.0f ? r : r * r;  // std::pow(x, 1 | 2)
        Float dLoss_dSamplingFraction = -ratio / woPdf * (bsdfPdf - dTreePdf);
        Float dLoss_dVariable = dLoss_dSamplingFraction * dBsdfSamplingFraction_dVariable(variable);
        Float l2RegGradient = 0.01f * variable;
        Float lossGradient = l2RegGradient + dLoss_dVariable;
        bsdfSamplingFractionOptimizer.append(lossGradient, statisticalWeight);
    }

    void dump(FILE *f, const Point &p, const Vec &size) const {  // GP:699-711
        float hdr[7] = {p.x, p.y, p.z, size.x, size.y, size.z, sampling.mean()};
        fwrite(hdr, 4, 7, f);
        uint64_t sw = (uint64_t)sampling.statisticalWeight(), nn = (uint64_t)sampling.numNodes();
        fwrite(&sw, 8, 1, f);
        fwrite(&nn, 8, 1, f);
        for (size_t i = 0; i < sampling.numNodes(); ++i) {
            const auto &node = sampling.node(i);
            for (int j = 0; j < 4; ++j) {
                float s = node.sum(j);
                uint16_t c = node.child(j);
                fwrite(&s, 4, 1, f);
                fwrite(&c, 2, 1, f);
            }
        }
    }

    DTree building;
    DTree sampling;
    AdamOptimizer bsdfSamplingFractionOptimizer{0.01f};
};

// ------------------------------------------------------------------------------------------------
// STreeNode GP:740-845
// ------------------------------------------------------------------------------------------------
struct STreeNode {
    STreeNode() {
        children = {};
        isLeaf = true;
        axis = 0;
    }
    int childIndex(Point &p) const {  // GP:747-755
        if (p[axis] < 0.5f) {
            p[axis] *= 2;
            return 0;
        } else {
            p[axis] = (p[axis] - 0.5f) * 2;
            return 1;
        }
    }
    int nodeIndex(Point &p) const { return children[childIndex(p)]; }

    DTreeWrapper *dTreeWrapper(Point &p, Vec &size, std::vector<STreeNode> &nodes) {  // GP:761-769
        ++t_cnt[CNT_STREE_LEVELS];
        if (isLeaf) return &dTree;
        size[axis] /= 2;
        return nodes[nodeIndex(p)].dTreeWrapper(p, size, nodes);
    }

    void forEachLeaf(const std::function<void(const DTreeWrapper *, const Point &, const Vec &)> &func, Point p, Vec size,
                     const std::vector<STreeNode> &nodes) const {  // GP:796-813
        if (isLeaf) {
            func(&dTree, p, size);
        } else {
            size[axis] /= 2;
            for (int i = 0; i < 2; ++i) {
                Point childP = p;
                if (i == 1) childP[axis] += size[axis];
                nodes[children[i]].forEachLeaf(func, childP, size, nodes);
            }
        }
    }

    static Float computeOverlappingVolume(const Point &min1, const Point &max1, const Point &min2, const Point &max2) {
        Float lengths[3];  // GP:815-821
        for (int i = 0; i < 3; ++i)
            lengths[i] = ppg_max(ppg_min(max1[i], max2[i]) - ppg_max(min1[i], min2[i]), 0.0f);
        return lengths[0] * lengths[1] * lengths[2];
    }

    void record(const Point &min1, const Point &max1, Point min2, Vec size2, const DTreeRecord &rec,
                EDirectionalFilter directionalFilter, ELoss loss, std::vector<STreeNode> &nodes, const Modes &modes) {
        Float w = computeOverlappingVolume(min1, max1, min2, min2 + size2);  // GP:823-839
        if (w > 0) {
            if (isLeaf) {
                dTree.record({rec.d, rec.radiance, rec.product, rec.woPdf, rec.bsdfPdf, rec.dTreePdf,
                              rec.statisticalWeight * w, rec.isDelta},
                             directionalFilter, loss, modes);
            } else {
                size2[axis] /= 2;
                for (int i = 0; i < 2; ++i) {
                    if (i & 1) min2[axis] += size2[axis];
                    nodes[children[i]].record(min1, max1, min2, size2, rec, directionalFilter, loss, nodes, modes);
                }
            }
        }
    }

    bool isLeaf;
    DTreeWrapper dTree;
    int axis;
    std::array<uint32_t, 2> children;
};

// ------------------------------------------------------------------------------------------------
// STree GP:848-1007
// ------------------------------------------------------------------------------------------------
class STree {
public:
    explicit STree(const AABB &aabb) {  // GP:850-860
        clear();
        m_aabb = aabb;
        Vec size = m_aabb.max - m_aabb.min;
        Float maxSize = ppg_max(ppg_max(size.x, size.y), size.z);
        m_aabb.max = m_aabb.min + Vec(maxSize);
    }
    void clear() {
        m_nodes.clear();
        m_nodes.emplace_back();
    }

    void subdivide(int nodeIdx, std::vector<STreeNode> &nodes) {  // GP:876-895
        nodes.resize(nodes.size() + 2);
        STreeNode &cur = nodes[nodeIdx];
        for (int i = 0; i < 2; ++i) {
            uint32_t idx = (uint32_t)nodes.size() - 2 + i;
            cur.children[i] = idx;
            nodes[idx].axis = (cur.axis + 1) % 3;
            nodes[idx].dTree = cur.dTree;
            nodes[idx].dTree.setStatisticalWeightBuilding(nodes[idx].dTree.statisticalWeightBuilding() / 2);
        }
        cur.isLeaf = false;
        cur.dTree = {};
    }

    DTreeWrapper *dTreeWrapper(Point p, Vec &size) {  // GP:897-905
        size = m_aabb.getExtents();
        p = Point(p - m_aabb.min);
        p.x /= size.x;
        p.y /= size.y;
        p.z /= size.z;
        ++t_cnt[CNT_STREE_LOOKUPS];
        return m_nodes[0].dTreeWrapper(p, size, m_nodes);
    }
    DTreeWrapper *dTreeWrapper(Point p) {
        Vec size;
        return dTreeWrapper(p, size);
    }

    template <typename F> void forEachDTreeWrapper(F func) {  // GP:912-933 (leaf order = node order)
        for (auto &node : m_nodes)
            if (node.isLeaf) func(&node.dTree);
    }
    void forEachDTreeWrapperConstP(const std::function<void(const DTreeWrapper *, const Point &, const Vec &)> &func) const {
        m_nodes[0].forEachLeaf(func, m_aabb.min, m_aabb.max - m_aabb.min, m_nodes);  // GP:920-922
    }

    void record(const Point &p, const Vec &dTreeVoxelSize, DTreeRecord rec, EDirectionalFilter directionalFilter, ELoss loss,
                const Modes &modes) {  // GP:935-943
        Float volume = 1;
        for (int i = 0; i < 3; ++i) volume *= dTreeVoxelSize[i];
        rec.statisticalWeight /= volume;
        m_nodes[0].record(p - dTreeVoxelSize * 0.5f, p + dTreeVoxelSize * 0.5f, m_aabb.min, m_aabb.getExtents(), rec,
                          directionalFilter, loss, m_nodes, modes);
    }

    void dump(FILE *f) const {  // GP:945-951
        forEachDTreeWrapperConstP([f](const DTreeWrapper *dTree, const Point &p, const Vec &size) {
            if (dTree->statisticalWeight() > 0) dTree->dump(f, p, size);
        });
    }

    bool shallSplit(const STreeNode &node, int, size_t samplesRequired) {  // GP:953-955 (float > size_t → float compare)
        return m_nodes.size() < std::numeric_limits<uint32_t>::max() - 1 &&
               node.dTree.statisticalWeightBuilding() > (Float)samplesRequired;
    }

    void refine(size_t sTreeThreshold, int maxMB) {  // GP:957-998
        if (maxMB >= 0) {
            size_t approxMemoryFootprint = 0;
            for (const auto &node : m_nodes) approxMemoryFootprint += node.dTree.approxMemoryFootprint();
            if (approxMemoryFootprint / 1000000 >= (size_t)maxMB) return;
        }
        struct StackNode {
            size_t index;
            int depth;
        };
        std::stack<StackNode> nodeIndices;
        nodeIndices.push({0, 1});
        while (!nodeIndices.empty()) {
            StackNode sNode = nodeIndices.top();
            nodeIndices.pop();
            if (m_nodes[sNode.index].isLeaf) {
                if (shallSplit(m_nodes[sNode.index], sNode.depth, sTreeThreshold)) subdivide((int)sNode.index, m_nodes);
            }
            if (!m_nodes[sNode.index].isLeaf) {
                const STreeNode &node = m_nodes[sNode.index];
                for (int i = 0; i < 2; ++i) nodeIndices.push({node.children[i], sNode.depth + 1});
            }
        }
    }

    const AABB &aabb() const { return m_aabb; }
    std::vector<STreeNode> &nodes() { return m_nodes; }

private:
    std::vector<STreeNode> m_nodes;
    AABB m_aabb;
};

// BSDFSamplingRecord subset
struct BRec {
    Vec wi, wo;
    Float eta = 1.0f;
    bool sampledDelta = false;  // sampledType & EDelta (EDelta includes ENull, bsdf.h:280)
    bool sampledNull = false;   // sampledType == ENull
    // `bumpmap` adapter (bumpmap.cpp:135-219): the intersection's shading frame and the frame perturbed by the displacement
    // texture's gradient; nullptr = no bump map on this surface
    const Frame *bumpSh = nullptr, *bumpPert = nullptr;
};

// A BSDF instance: the ABI record plus what the plugin's configure() precomputes
struct Material : ppg_material {
    Float fdrInt = 0, specularSamplingWeight = 0, invEta2 = 0;  // plastic.cpp:191-204
    const Float *rt = nullptr;  // roughplastic: this material's rough-transmittance slice (ppg_scene.rtrans), rtN samples + 1
    uint32_t rtN = 0;
    Spectrum R() const { return Spectrum(reflectance[0], reflectance[1], reflectance[2]); }
    Spectrum S() const { return Spectrum(specular[0], specular[1], specular[2]); }
    Spectrum Eta() const { return Spectrum(eta[0], eta[1], eta[2]); }
    Spectrum K() const { return Spectrum(k[0], k[1], k[2]); }
    Spectrum Opacity() const { return Spectrum(opacity[0], opacity[1], opacity[2]); }
    bool masked() const { return (flags & PPG_MAT_MASK) != 0; }
    void configure() {
        if (type == PPG_BSDF_TWOSIDED_DIFFUSE) { type = PPG_BSDF_DIFFUSE; flags |= PPG_MAT_TWOSIDED; }
        if (type == PPG_BSDF_MIRROR) { for (int i = 0; i < 3; ++i) { eta[i] = 0.0f; k[i] = 1.0f; } }  // material "none", conductor.cpp:171-173
        if (type == PPG_BSDF_ROUGHCONDUCTOR || type == PPG_BSDF_ROUGHDIELECTRIC || type == PPG_BSDF_ROUGHPLASTIC) alpha = ppg_max(alpha, 1e-4f);  // microfacet.h:135
        if (type == PPG_BSDF_ROUGHPLASTIC) {  // roughplastic.cpp:277-283 (Fdr comes from the slice, :372)
            Float dAvg = luminance(R()), sAvg = luminance(S());
            specularSamplingWeight = sAvg / (dAvg + sAvg);
            invEta2 = 1.0f / (eta[0] * eta[0]);
            fdrInt = rt ? 1 - rt[rtN] : 0.0f;
        }
        if (type == PPG_BSDF_PLASTIC) {
            fdrInt = ppg_fresnel_diffuse_reflectance(1 / eta[0]);
            Float dAvg = luminance(R()), sAvg = luminance(S());
            specularSamplingWeight = sAvg / (dAvg + sAvg);
            invEta2 = 1 / (eta[0] * eta[0]);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Scene: triangles + the callees of Li ("direct callees")
// ------------------------------------------------------------------------------------------------
struct Intersection {
    bool valid = false;
    Float t = std::numeric_limits<Float>::infinity();
    Point p;
    Vec geoN;
    Frame shFrame;
    Vec wi;
    int prim = -1;
    int emitter = -1;
    uint32_t material = 0;
    Point2 uv{0, 0};   // its.uv (skdtree.h:403-410); only filled for surfaces whose BSDF reads a texture
    Vec dpdu, dpdv;    // its.dpdu / dpdv (skdtree.h:374-381), likewise
};

// DiscreteDistribution (pmf.h:30-189): float running sums, normalize(), sample()/sampleReuse()
struct Pmf {
    std::vector<Float> cdf{0.0f};
    Float sum = 0, normalization = 0;
    void append(Float v) { cdf.push_back(cdf.back() + v); }
    Float operator[](size_t i) const { return cdf[i + 1] - cdf[i]; }
    Float normalize() {  // pmf.h:101-114
        sum = cdf.back();
        if (sum > 0) {
            normalization = 1.0f / sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
            cdf.back() = 1.0f;
        } else {
            normalization = 0.0f;
        }
        return sum;
    }
    size_t sample(Float v) const {  // pmf.h:124-136
        ptrdiff_t entry = std::lower_bound(cdf.begin(), cdf.end(), v) - cdf.begin();
        size_t index = std::min(cdf.size() - 2, (size_t)std::max((ptrdiff_t)0, entry - 1));
        while ((*this)[index] == 0 && index < cdf.size() - 1) ++index;
        return index;
    }
    size_t sampleReuse(Float &v, Float &pdf) const {  // pmf.h:183-188
        size_t index = sample(v);
        pdf = (*this)[index];
        v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
};

// DirectSamplingRecord subset (common.h / records.inl:160-178)
struct DRec {
    Point ref, p;
    Vec refN, n, d;
    Float dist = 0, pdf = 0;
    int emitter = -1;
};

inline Vec squareToCosineHemisphere(const Point2 &sample);  // warp.cpp:43-52, defined below

// EnvironmentMap (emitters/envmap.cpp): latitude-longitude map, level-0 bilinear lookups (mipmap.h:503-596, u repeats, v clamps),
// luminance x sin(theta) importance sampling with tent-filtered pixel positions and the matching solid-angle density
struct EnvMap {
    int w = 0, h = 0;
    std::vector<Spectrum> texel;
    Float scale = 1;
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // toWorld rotation; inverse = transpose
    std::vector<float> cdfRows, cdfCols;
    std::vector<Float> rowWeights;
    Float normalization = 0, pixelSizeX = 0, pixelSizeY = 0;

    static int floorToInt(Float v) { return (int)std::floor(v); }
    Vec toWorld(const Vec &v) const { return Vec(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z); }
    Vec toLocal(const Vec &v) const { return Vec(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z); }
    Spectrum evalTexel(int x, int y) const {  // mipmap.h:503-570: ERepeat in u, EClamp in v
        if (x < 0 || x >= w) { x %= w; if (x < 0) x += w; }  // math::modulo, math.h:67-70
        if (y < 0 || y >= h) y = y < 0 ? 0 : h - 1;
        return texel[(size_t)y * w + x];
    }
    void configure() {  // envmap.cpp:255-322
        cdfCols.assign((size_t)(w + 1) * h, 0.0f);
        cdfRows.assign(h + 1, 0.0f);
        rowWeights.assign(h, 0.0f);
        size_t colPos = 0, rowPos = 0;
        Float rowSum = 0.0f;
        cdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            Float colSum = 0;
            cdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) {
                colSum += luminance(texel[(size_t)y * w + x]);
                cdfCols[colPos++] = (float)colSum;
            }
            float normalization = 1.0f / (float)colSum;
            for (int x = 1; x < w; ++x) cdfCols[colPos - x - 1] *= normalization;
            cdfCols[colPos - 1] = 1.0f;
            Float weight, cosUnused;
            ppg_sincos((y + 0.5f) * PPG_PI_F / h, &weight, &cosUnused);  // std::sin in the reference
            rowWeights[y] = weight;
            rowSum += colSum * weight;
            cdfRows[rowPos++] = (float)rowSum;
        }
        float norm = 1.0f / (float)rowSum;
        for (int y = 1; y < h; ++y) cdfRows[rowPos - y - 1] *= norm;
        cdfRows[rowPos - 1] = 1.0f;
        normalization = 1.0f / (rowSum * (2 * PPG_PI_F / w) * (PPG_PI_F / h));
        pixelSizeX = 2 * PPG_PI_F / w; pixelSizeY = PPG_PI_F / h;
        valid = rowSum > 0 && std::isfinite(rowSum);
    }
    bool valid = false;
    // evalEnvironment (envmap.cpp:381-407) for a ray without differentials: `d` is the world direction the ray travels in
    Spectrum eval(const Vec &dWorld) const {
        const Vec v = toLocal(dWorld);
        const Point2 uv{ppg_atan2(v.x, -v.z) * (PPG_INV_PI_F * 0.5f), ppg_acos(v.y) * PPG_INV_PI_F};
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return Spectrum(0.0f);
        const Float u = uv.x * w - 0.5f, vv = uv.y * h - 0.5f;  // mipmap.h:585-595
        const int xPos = floorToInt(u), yPos = floorToInt(vv);
        const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = vv - yPos, dy2 = 1.0f - dy1;
        const Spectrum value = evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos) * dx1 * dy2 +
                               evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
        return value * scale;
    }
    static uint32_t sampleReuse(const float *cdf, uint32_t size, Float &sample) {  // envmap.cpp:659-664
        const float *entry = std::lower_bound(cdf, cdf + size + 1, (float)sample);
        uint32_t index = std::min((uint32_t)std::max((ptrdiff_t)0, entry - cdf - 1), size - 1);
        sample = (sample - (Float)cdf[index]) / (Float)(cdf[index + 1] - cdf[index]);
        return index;
    }
    static Float intervalToTent(Float sample) {  // warp.cpp:143-155
        Float sign;
        if (sample < 0.5f) { sign = 1; sample *= 2; }
        else { sign = -1; sample = 2 * (sample - 0.5f); }
        return sign * (1 - std::sqrt(sample));
    }
    int clampRow(int y) const { return y < 0 ? 0 : (y > h - 1 ? h - 1 : y); }
    // internalSampleDirection (envmap.cpp:557-595): d in the emitter's frame
    void sampleDirection(Point2 sample, Vec &d, Spectrum &value, Float &pdf) const {
        uint32_t row = sampleReuse(cdfRows.data(), (uint32_t)h, sample.y);
        uint32_t col = sampleReuse(cdfCols.data() + (size_t)row * (w + 1), (uint32_t)w, sample.x);
        const Float posX = (Float)col + intervalToTent(sample.x), posY = (Float)row + intervalToTent(sample.y);
        const int xPos = floorToInt(posX), yPos = floorToInt(posY);
        const Float dx1 = posX - xPos, dx2 = 1.0f - dx1, dy1 = posY - yPos, dy2 = 1.0f - dy1;
        const Spectrum value1 = evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos + 1, yPos) * dx1 * dy2;
        const Spectrum value2 = evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
        value = (value1 + value2) * scale;
        pdf = (luminance(value1) * rowWeights[clampRow(yPos)] + luminance(value2) * rowWeights[clampRow(yPos + 1)]) * normalization;
        Float sinPhi, cosPhi, sinTheta, cosTheta;
        ppg_sincos(pixelSizeX * (posX + 0.5f), &sinPhi, &cosPhi);
        ppg_sincos(pixelSizeY * (posY + 0.5f), &sinTheta, &cosTheta);
        d = Vec(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
        pdf /= ppg_max(ppg_abs(sinTheta), PPG_EPSILON);
    }
    // internalPdfDirection (envmap.cpp:598-633): d in the emitter's frame
    Float pdfDirection(const Vec &d) const {
        const Point2 uv{ppg_atan2(d.x, -d.z) * (PPG_INV_PI_F * 0.5f), ppg_acos(d.y) * PPG_INV_PI_F};
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return 0.0f;
        const Float u = uv.x * w - 0.5f, v = uv.y * h - 0.5f;
        const int xPos = floorToInt(u), yPos = floorToInt(v);
        const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        const Spectrum value1 = evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos + 1, yPos) * dx1 * dy2;
        const Spectrum value2 = evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
        const Float sinTheta = std::sqrt(ppg_max(0.0f, 1 - d.y * d.y));
        return (luminance(value1) * rowWeights[clampRow(yPos)] + luminance(value2) * rowWeights[clampRow(yPos + 1)]) * normalization /
               ppg_max(ppg_abs(sinTheta), PPG_EPSILON);
    }
};

// BitmapTexture as this integrator sees it (include/ppg.h ppg_texture): Li never computes UV partials, so every lookup is
// BitmapTexture::eval(uv) (bitmap.cpp:431-452) = MIPMap::evalBilinear(0, uv) / evalBox(0, uv) (mipmap.h:566-596) and bump maps read
// evalGradientBilinear(0, uv) (mipmap.h:601-626); texel addressing with the boundary conditions of evalTexel (mipmap.h:503-563).
struct Texture {
    int w = 0, h = 0;
    std::vector<Spectrum> texel;
    Float su = 1, sv = 1, ou = 0, ov = 0;
    int wrapU = PPG_WRAP_REPEAT, wrapV = PPG_WRAP_REPEAT;
    bool nearest = false;
    static int floorToInt(Float v) { return (int)std::floor(v); }
    static int modulo(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }  // math.h:67-70
    Spectrum evalTexel(int x, int y) const {
        if (x < 0 || x >= w) {
            switch (wrapU) {
                case PPG_WRAP_REPEAT: x = modulo(x, w); break;
                case PPG_WRAP_CLAMP: x = x < 0 ? 0 : w - 1; break;
                case PPG_WRAP_MIRROR: x = modulo(x, 2 * w); if (x >= w) x = 2 * w - x - 1; break;
                case PPG_WRAP_ZERO: return Spectrum(0.0f);
                default: return Spectrum(1.0f);
            }
        }
        if (y < 0 || y >= h) {
            switch (wrapV) {
                case PPG_WRAP_REPEAT: y = modulo(y, h); break;
                case PPG_WRAP_CLAMP: y = y < 0 ? 0 : h - 1; break;
                case PPG_WRAP_MIRROR: y = modulo(y, 2 * h); if (y >= h) y = 2 * h - y - 1; break;
                case PPG_WRAP_ZERO: return Spectrum(0.0f);
                default: return Spectrum(1.0f);
            }
        }
        return texel[(size_t)y * w + x];
    }
    Point2 transform(const Point2 &uv) const { return Point2{uv.x * su + ou, uv.y * sv + ov}; }  // Texture2D::eval, texture.cpp:112-113
    Spectrum eval(const Point2 &itsUv) const {
        const Point2 uv = transform(itsUv);
        if (nearest) return evalTexel(floorToInt(uv.x * w), floorToInt(uv.y * h));  // evalBox(0, uv)
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return Spectrum(0.0f);
        const Float u = uv.x * w - 0.5f, v = uv.y * h - 0.5f;
        const int xPos = floorToInt(u), yPos = floorToInt(v);
        const Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        return evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos) * dx1 * dy2 +
               evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
    }
    // Texture2D::evalGradient(its) (texture.cpp:123-131) over BitmapTexture::evalGradient(uv) (bitmap.cpp:454-477)
    void evalGradient(const Point2 &itsUv, Spectrum *gradient) const {
        const Point2 uv = transform(itsUv);
        if (nearest || !std::isfinite(uv.x) || !std::isfinite(uv.y)) { gradient[0] = gradient[1] = Spectrum(0.0f); return; }
        const Float u = uv.x * w - 0.5f, v = uv.y * h - 0.5f;
        const int xPos = floorToInt(u), yPos = floorToInt(v);
        const Float dx = u - xPos, dy = v - yPos;
        const Spectrum p00 = evalTexel(xPos, yPos), p10 = evalTexel(xPos + 1, yPos), p01 = evalTexel(xPos, yPos + 1), p11 = evalTexel(xPos + 1, yPos + 1);
        const Spectrum tmp = p01 + p10 - p11;
        gradient[0] = (p10 + p00 * (dy - 1) - tmp * dy) * (Float)w;
        gradient[1] = (p01 + p00 * (dx - 1) - tmp * dx) * (Float)h;
        gradient[0] = gradient[0] * su;
        gradient[1] = gradient[1] * sv;
    }
};

struct Scene {
    // one area emitter = the triangles carrying its id, in index order (area.cpp: one emitter per shape)
    struct EmitterMesh { std::vector<uint32_t> tris; Pmf areaDistr; Float surfaceArea = -1, invSurfaceArea = -1; };
    std::vector<EmitterMesh> emMesh;
    Pmf emitterPDF;
    // ConstantBackgroundEmitter (constant.cpp): the last entry of the emitter list; bounding sphere of createShape() (constant.cpp:67-78)
    bool hasEnv = false;       // an environment emitter: constant (envRadiance) or image based (envMap.valid)
    Spectrum envRadiance;
    EnvMap envMap;
    // Scene::evalEnvironment for a ray travelling along d that left the scene
    Spectrum envEval(const Vec &d) const { return envMap.valid ? envMap.eval(d) : envRadiance; }
    Point bsCenter;
    Float bsRadius = 0;
    int envIndex() const { return (int)emMesh.size(); }
    std::vector<Point> P;
    std::vector<Vec> N;
    bool hasNormals = false;
    std::vector<uint32_t> idx, triMat;
    std::vector<int32_t> triEmitter;
    std::vector<Material> materials;
    std::vector<Float> rtrans;  // ppg_scene.rtrans (roughplastic slices)
    std::vector<Texture> textures;     // ppg_scene.textures
    std::vector<Point2> UV;            // ppg_scene.texcoords (per vertex; NaN = none), empty if the scene has none
    std::vector<ppg_sphere> spheres;   // analytic spheres (shapes/sphere.cpp), primitives nTris(), nTris() + 1, ..
    std::vector<int> emitterSphere;    // per emitter: the sphere carrying it, or -1 (a triangle emitter)
    std::vector<ppg_emitter> emitters;
    ppg_camera cam;
    AABB aabb;  // what Scene::getAABB() returns: the kd-tree's enlarged box (gkdtree.h:1213-1220)

    // own BVH for scenes too large for brute force (independent of the product's BVH)
    struct BNode { Point bmin, bmax; int left, right, first, count; };
    std::vector<BNode> bvh;
    std::vector<uint32_t> order;

    size_t nTris() const { return idx.size() / 3; }

    void finalize() {
        Point mn(std::numeric_limits<Float>::infinity()), mx(-std::numeric_limits<Float>::infinity());
        for (size_t t = 0; t < idx.size(); ++t) {
            const Point &p = P[idx[t]];
            for (int a = 0; a < 3; ++a) { mn[a] = ppg_min(mn[a], p[a]); mx[a] = ppg_max(mx[a], p[a]); }
        }
        for (const ppg_sphere &sp : spheres)  // Sphere::getAABB, sphere.cpp:152-157
            for (int a = 0; a < 3; ++a) { mn[a] = ppg_min(mn[a], sp.center[a] - sp.radius); mx[a] = ppg_max(mx[a], sp.center[a] + sp.radius); }
        const Float eps = 1e-3f;  // MTS_KD_AABB_EPSILON
        aabb.min = mn - ((mx - mn) * eps + Vec(eps));
        aabb.max = mx + ((mx - aabb.min) * eps + Vec(eps));
        // Scene::initializeBidirectional (scene.cpp:386-414): expandBy(sensor AABB) = the pinhole position
        // (perspective.cpp:444-446); area emitters lie inside the geometry box already.
        for (int a = 0; a < 3; ++a) {
            Float c = cam.camera_to_world[4 * a + 3];
            aabb.min[a] = ppg_min(aabb.min[a], c);
            aabb.max[a] = ppg_max(aabb.max[a], c);
        }
        buildAccel();
        bvh.clear();
        if (nTris() > 64) buildBVH();
        // TriMesh::prepareSamplingTable (trimesh.cpp:388-403) per emitter; Scene::configure's emitter pmf
        // (scene.cpp:375-380, samplingWeight = 1)
        emMesh.assign(emitters.size(), EmitterMesh());
        emitterSphere.assign(emitters.size(), -1);
        for (size_t k = 0; k < spheres.size(); ++k)
            if (spheres[k].emitter >= 0) emitterSphere[spheres[k].emitter] = (int)k;
        for (uint32_t t = 0; t < nTris(); ++t)
            if (triEmitter[t] >= 0) emMesh[triEmitter[t]].tris.push_back(t);
        emitterPDF = Pmf();
        for (EmitterMesh &m : emMesh) {
            for (uint32_t t : m.tris) {
                const Point &p0 = P[idx[3 * t]], &p1 = P[idx[3 * t + 1]], &p2 = P[idx[3 * t + 2]];
                m.areaDistr.append(0.5f * length(cross(p1 - p0, p2 - p0)));  // Triangle::surfaceArea, triangle.cpp:61-67
            }
            if (!m.tris.empty()) {
                m.surfaceArea = m.areaDistr.normalize();
                m.invSurfaceArea = 1.0f / m.surfaceArea;
            }
            emitterPDF.append(1.0f);
        }
        if (hasEnv) {
            emitterPDF.append(1.0f);
            bsCenter = (aabb.max + aabb.min) * 0.5f;  // AABB::getBSphere, aabb.cpp:44-47
            bsRadius = ppg_max(PPG_EPSILON, length(bsCenter - aabb.max) * 1.5f);
        }
        if (!emMesh.empty() || hasEnv) emitterPDF.normalize();
    }

    // solveQuadratic (util.cpp:447-485)
    static bool solveQuadratic(Float a, Float b, Float c, Float &x0, Float &x1) {
        if (a == 0) {
            if (b != 0) { x0 = x1 = -c / b; return true; }
            return false;
        }
        Float discrim = b * b - 4.0f * a * c;
        if (discrim < 0) return false;
        Float temp, sqrtDiscrim = std::sqrt(discrim);
        if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
        else temp = -0.5f * (b + sqrtDiscrim);
        x0 = temp / a;
        x1 = c / temp;
        if (x0 > x1) std::swap(x0, x1);
        return true;
    }
    // Sphere::rayIntersect (sphere.cpp:164-189) with solveQuadraticDouble (util.cpp:487-525)
    static bool sphereRayIntersect(const ppg_sphere &sp, const Point &ro, const Vec &rd, Float mint, Float maxt, Float &t) {
        const double ox = (double)ro.x - (double)sp.center[0], oy = (double)ro.y - (double)sp.center[1], oz = (double)ro.z - (double)sp.center[2];
        const double dx = rd.x, dy = rd.y, dz = rd.z;
        const double A = dx * dx + dy * dy + dz * dz;
        const double B = 2 * (ox * dx + oy * dy + oz * dz);
        const double C = (ox * ox + oy * oy + oz * oz) - sp.radius * sp.radius;  // m_radius * m_radius is a Float product
        double nearT, farT;
        if (A == 0) {
            if (B != 0) nearT = farT = -C / B;
            else return false;
        } else {
            const double discrim = B * B - 4.0f * A * C;
            if (discrim < 0) return false;
            double temp, sqrtDiscrim = std::sqrt(discrim);
            if (B < 0) temp = -0.5f * (B - sqrtDiscrim);
            else temp = -0.5f * (B + sqrtDiscrim);
            nearT = temp / A;
            farT = C / temp;
            if (nearT > farT) std::swap(nearT, farT);
        }
        if (!(nearT <= maxt && farT >= mint)) return false;
        if (nearT < mint) {
            if (farT > maxt) return false;
            t = (Float)farT;
        } else {
            t = (Float)nearT;
        }
        return true;
    }
    // Sphere::fillIntersectionRecord (sphere.cpp:213-263): position re-projected onto the sphere, normal, dpdu of the (theta, phi)
    // parameterisation (the shading frame's tangent, skdtree.h:427); worldToObject of a vector taken as the transposed rotation
    static void sphereFill(const ppg_sphere &sp, const Point &ro, const Vec &rd, Float t, Point &p, Vec &n, Vec &dpdu) {
        const Point c(sp.center[0], sp.center[1], sp.center[2]);
        p = ro + rd * t;
        p = c + normalize(p - c) * sp.radius;
        const Vec v = p - c;
        const float *R = sp.to_world;
        const Vec local(R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z);
        const Vec du = Vec(-local.y, local.x, 0.0f) * (2 * PPG_PI_F);
        dpdu = Vec(R[0] * du.x + R[1] * du.y + R[2] * du.z, R[3] * du.x + R[4] * du.y + R[5] * du.z, R[6] * du.x + R[7] * du.y + R[8] * du.z);
        n = normalize(p - c);
        if (sp.flip_normals) n = n * -1.0f;
    }
    // Sphere::sampleDirect (sphere.cpp:291-355), solid-angle measure
    static void sphereSampleDirect(const ppg_sphere &sp, DRec &dRec, const Point2 &sample) {
        const Point c(sp.center[0], sp.center[1], sp.center[2]);
        const Float invSurfaceArea = 1 / (4 * PPG_PI_F * sp.radius * sp.radius);
        const Vec refToCenter = c - dRec.ref;
        const Float refDist2 = dot(refToCenter, refToCenter);
        const Float invRefDist = 1.0f / std::sqrt(refDist2);
        const Float sinAlpha = sp.radius * invRefDist;
        if (sinAlpha < 1 - PPG_EPSILON) {  // outside: the cone subtended by the sphere
            Float cosAlpha = std::sqrt(ppg_max(0.0f, 1.0f - sinAlpha * sinAlpha));
            // warp::squareToUniformCone (warp.cpp:54-63) in Frame(refToCenter * invRefDist) (coordinateSystem, util.cpp:592-601)
            Float cosTheta = (1 - sample.x) + sample.x * cosAlpha;
            Float sinTheta = std::sqrt(ppg_max(0.0f, 1.0f - cosTheta * cosTheta));
            Float sinPhi, cosPhi;
            ppg_sincos(2.0f * PPG_PI_F * sample.y, &sinPhi, &cosPhi);
            const Vec lv(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
            const Vec a = refToCenter * invRefDist;
            Vec sF, tF;
            if (ppg_abs(a.x) > ppg_abs(a.y)) {
                Float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
                tF = Vec(a.z * invLen, 0.0f, -a.x * invLen);
            } else {
                Float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
                tF = Vec(0.0f, a.z * invLen, -a.y * invLen);
            }
            sF = cross(tF, a);
            dRec.d = sF * lv.x + tF * lv.y + a * lv.z;
            dRec.pdf = (PPG_INV_PI_F * 0.5f) / (1 - cosAlpha);  // squareToUniformConePdf, warp.h:74-76
            const Float projDist = dot(refToCenter, dRec.d);
            const Float baseT = refDist2 / projDist;
            const Point query = dRec.ref + dRec.d * baseT;
            const Vec queryToCenter = c - query;
            const Float queryDist2 = dot(queryToCenter, queryToCenter);
            const Float queryProjDist = dot(queryToCenter, dRec.d);
            Float A = 1.0f, B = -2 * queryProjDist, C = queryDist2 - sp.radius * sp.radius;
            Float nearT, farT;
            if (!solveQuadratic(A, B, C, nearT, farT)) nearT = queryProjDist;
            dRec.dist = baseT + nearT;
            dRec.n = normalize(dRec.d * nearT - queryToCenter);
            dRec.p = c + dRec.n * sp.radius;
        } else {  // inside: uniform over the sphere
            Float z = 1.0f - 2.0f * sample.y;  // warp::squareToUniformSphere, warp.cpp:25-31
            Float r = std::sqrt(ppg_max(0.0f, 1.0f - z * z));
            Float sinPhi, cosPhi;
            ppg_sincos(2.0f * PPG_PI_F * sample.x, &sinPhi, &cosPhi);
            const Vec d(r * cosPhi, r * sinPhi, z);
            dRec.p = c + d * sp.radius;
            dRec.n = d;
            dRec.d = dRec.p - dRec.ref;
            Float dist2 = dot(dRec.d, dRec.d);
            dRec.dist = std::sqrt(dist2);
            dRec.d = dRec.d / dRec.dist;
            dRec.pdf = invSurfaceArea * dist2 / ppg_abs(dot(dRec.d, dRec.n));
        }
        if (sp.flip_normals) dRec.n = dRec.n * -1.0f;
    }
    // Sphere::pdfDirect (sphere.cpp:357-378), solid-angle measure
    static Float spherePdfDirect(const ppg_sphere &sp, const DRec &dRec) {
        const Point c(sp.center[0], sp.center[1], sp.center[2]);
        const Vec refToCenter = c - dRec.ref;
        const Float invRefDist = 1.0f / length(refToCenter);
        const Float sinAlpha = sp.radius * invRefDist;
        if (sinAlpha < 1 - PPG_EPSILON) {
            Float cosAlpha = std::sqrt(ppg_max(0.0f, 1 - sinAlpha * sinAlpha));
            return (PPG_INV_PI_F * 0.5f) / (1 - cosAlpha);
        }
        const Float invSurfaceArea = 1 / (4 * PPG_PI_F * sp.radius * sp.radius);
        return invSurfaceArea * dRec.dist * dRec.dist / ppg_abs(dot(dRec.d, dRec.n));
    }

    // BSphere::rayIntersect (bsphere.h:88-95) + solveQuadratic (util.cpp:447-485)
    bool bsphereIntersect(const Point &o_, const Vec &d, Float &nearHit, Float &farHit) const {
        Vec o = o_ - bsCenter;
        Float A = dot(d, d), B = 2 * dot(o, d), Cq = dot(o, o) - bsRadius * bsRadius;
        if (A == 0) {
            if (B != 0) { nearHit = farHit = -Cq / B; return true; }
            return false;
        }
        Float discrim = B * B - 4.0f * A * Cq;
        if (discrim < 0) return false;
        Float temp, sqrtDiscrim = std::sqrt(discrim);
        if (B < 0) temp = -0.5f * (B - sqrtDiscrim);
        else temp = -0.5f * (B + sqrtDiscrim);
        nearHit = temp / A;
        farHit = Cq / temp;
        if (nearHit > farHit) std::swap(nearHit, farHit);
        return true;
    }
    // ConstantBackgroundEmitter::fillDirectSamplingRecord (constant.cpp:240-254)
    bool envFillDirectSamplingRecord(DRec &dRec, const Point &o, const Vec &d) const {
        Float nearT, farT;
        if (!bsphereIntersect(o, d, nearT, farT) || nearT > 0 || farT < 0) return false;
        dRec.p = o + d * farT;
        dRec.n = normalize(bsCenter - dRec.p);
        dRec.emitter = envIndex();
        dRec.d = d;
        dRec.dist = farT;
        return true;
    }
    // ConstantBackgroundEmitter::sampleDirect (constant.cpp:176-214)
    Spectrum envSampleDirect(DRec &dRec, const Point2 &sample) const {
        if (envMap.valid) {  // EnvironmentMap::sampleDirect, envmap.cpp:510-538
            Spectrum value; Vec dl; Float pdfM;
            envMap.sampleDirection(sample, dl, value, pdfM);
            const Vec dw = envMap.toWorld(dl);
            Float nearT, farT;
            if (isZero(value) || pdfM == 0 || !bsphereIntersect(dRec.ref, dw, nearT, farT) || nearT >= 0 || farT <= 0) {
                dRec.pdf = 0.0f;
                return Spectrum(0.0f);
            }
            dRec.pdf = pdfM;
            dRec.p = dRec.ref + dw * farT;
            dRec.n = normalize(bsCenter - dRec.p);
            dRec.dist = farT;
            dRec.d = dw;
            return value / pdfM;
        }
        Vec d;
        Float pdf;
        const bool hasRefN = !(dRec.refN.x == 0 && dRec.refN.y == 0 && dRec.refN.z == 0);
        if (hasRefN) {
            d = squareToCosineHemisphere(sample);
            pdf = PPG_INV_PI_F * d.z;
            // Frame(dRec.refN): coordinateSystem (util.cpp:592-601)
            const Vec &a = dRec.refN;
            Vec sF, tF;
            if (ppg_abs(a.x) > ppg_abs(a.y)) {
                Float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
                tF = Vec(a.z * invLen, 0.0f, -a.x * invLen);
            } else {
                Float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
                tF = Vec(0.0f, a.z * invLen, -a.y * invLen);
            }
            sF = cross(tF, a);
            d = sF * d.x + tF * d.y + a * d.z;
        } else {
            Float z = 1.0f - 2.0f * sample.y;  // warp::squareToUniformSphere, warp.cpp:25-31
            Float r = std::sqrt(ppg_max(0.0f, 1.0f - z * z));
            Float sinPhi, cosPhi;
            ppg_sincos(2.0f * PPG_PI_F * sample.x, &sinPhi, &cosPhi);
            d = Vec(r * cosPhi, r * sinPhi, z);
            pdf = PPG_INV_PI_F * 0.25f;  // INV_FOURPI
        }
        Float nearT, farT;
        dRec.pdf = 0.0f;
        if (!bsphereIntersect(dRec.ref, d, nearT, farT)) return Spectrum(0.0f);
        if (!(nearT < 0 && farT > 0)) return Spectrum(0.0f);
        dRec.p = dRec.ref + d * farT;
        dRec.n = normalize(bsCenter - dRec.p);
        dRec.d = d;
        dRec.dist = farT;
        dRec.pdf = pdf;
        if (hasRefN && dot(dRec.d, dRec.refN) <= 0) return Spectrum(0.0f);
        return envRadiance / pdf;
    }

    // Scene::sampleAttenuatedEmitterDirect (scene.cpp:876-897) → AreaLight::sampleDirect (area.cpp:158-173) →
    // Shape::sampleDirect (shape.cpp:102-115) → TriMesh::samplePosition (trimesh.cpp:412-423) →
    // Triangle::sample (triangle.cpp:24-58), then evalTransmittance (scene.cpp:619-679) without media / null BSDFs
    // Scene::evalTransmittance (scene.cpp:619-679), surfaces only: 0 behind an occluder, otherwise the product of the
    // null components (BSDF::eval with typeMask = ENull, EDiscrete, in the GEOMETRIC frame) of the surfaces passed.
    // evalNull(material, cosThetaI) is supplied by the caller (the BSDF dispatch is defined further down).
    template <typename HasNull, typename EvalNull>
    Spectrum evalTransmittance(const Point &p1, const Point &p2, bool p2OnSurface, int &interactions, uint64_t &rays, HasNull hasNull,
                               EvalNull evalNull) const {
        Vec d = p2 - p1;
        Float remaining = length(d);
        d = d / remaining;
        const Float lengthFactor = p2OnSurface ? (1 - PPG_SHADOW_EPSILON) : 1;
        Point o = p1;
        Float mint = PPG_EPSILON, maxt = remaining * lengthFactor;  // p1OnSurface
        Spectrum transmittance(1.0f);
        Intersection its;
        int maxInteractions = interactions;
        interactions = 0;
        while (remaining > 0) {
            ++rays;
            bool surface = rayIntersect(o, d, mint, maxt, its);
            if (surface && (interactions == maxInteractions || !hasNull(materials[its.material]))) return Spectrum(0.0f);
            if (!surface || isZero(transmittance)) break;
            Float cosThetaI = dot(its.geoN, -d);  // wi = -wo in Frame(its.geoFrame.n)
            transmittance = mul(transmittance, evalNull(materials[its.material], cosThetaI));
            if (++interactions > 100) break;
            o = o + d * its.t;
            remaining -= its.t;
            maxt = remaining * lengthFactor;
            mint = PPG_EPSILON;
        }
        return transmittance;
    }

    template <typename HasNull, typename EvalNull>
    Spectrum sampleEmitterDirect(DRec &dRec, Point2 sample, uint64_t &shadowRays, int interactions, HasNull hasNull, EvalNull evalNull) const {
        dRec.pdf = 0;
        if (emMesh.empty() && !hasEnv) return Spectrum(0.0f);
        Float emPdf;
        size_t index = emitterPDF.sampleReuse(sample.x, emPdf);
        Spectrum value(0.0f);
        const bool isEnv = hasEnv && (int)index == envIndex();
        if (isEnv) {
            value = envSampleDirect(dRec, sample);
        } else if (emitterSphere[index] >= 0) {  // AreaLight::sampleDirect (area.cpp:158-173) on a sphere
            sphereSampleDirect(spheres[emitterSphere[index]], dRec, sample);
            if (!(dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0)) {
                dRec.pdf = 0.0f;
                return Spectrum(0.0f);
            }
            const float *r = emitters[index].radiance;
            value = Spectrum(r[0], r[1], r[2]) / dRec.pdf;
        } else {
            const EmitterMesh &m = emMesh[index];
            if (m.tris.empty()) return Spectrum(0.0f);
            Float dummy;
            size_t ti = m.areaDistr.sampleReuse(sample.y, dummy);
            const uint32_t t = m.tris[ti];
            const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
            const Point &p0 = P[i0], &p1 = P[i1], &p2 = P[i2];
            Float a = std::sqrt(ppg_max(0.0f, 1.0f - sample.x));  // warp::squareToUniformTriangle, warp.cpp:76-79
            Point2 bary{1 - a, a * sample.y};
            Vec sideA = p1 - p0, sideB = p2 - p0;
            dRec.p = p0 + sideA * bary.x + sideB * bary.y;
            if (hasNormals) dRec.n = normalize(N[i0] * (1.0f - bary.x - bary.y) + N[i1] * bary.x + N[i2] * bary.y);
            else dRec.n = normalize(cross(sideA, sideB));
            dRec.pdf = m.invSurfaceArea;
            dRec.d = dRec.p - dRec.ref;
            Float distSquared = dot(dRec.d, dRec.d);
            dRec.dist = std::sqrt(distSquared);
            dRec.d = dRec.d / dRec.dist;
            Float dp = ppg_abs(dot(dRec.d, dRec.n));
            dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
            if (!(dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0)) {
                dRec.pdf = 0.0f;
                return Spectrum(0.0f);
            }
            const float *r = emitters[index].radiance;
            value = Spectrum(r[0], r[1], r[2]) / dRec.pdf;
        }
        if (dRec.pdf == 0) return Spectrum(0.0f);  // scene.cpp:885, 896
        // value *= evalTransmittance(its.p, true, dRec.p, emitter->isOnSurface(), ...) / emPdf
        if (!isZero(value)) {
            Spectrum tr = evalTransmittance(dRec.ref, dRec.p, !isEnv, interactions, shadowRays, hasNull, evalNull);
            value = mul(value, tr) / emPdf;
        }
        dRec.emitter = (int)index;
        dRec.pdf *= emPdf;
        return value;
    }

    // Scene::pdfEmitterDirect (scene.cpp:949-952) → AreaLight::pdfDirect (area.cpp:175-183) → Shape::pdfDirect
    // (shape.cpp:117-126), solid-angle measure
    Float pdfEmitterDirect(const DRec &dRec) const {
        if (dRec.emitter < 0) return 0.0f;
        Float pdf = 0.0f;
        if (hasEnv && dRec.emitter == envIndex() && envMap.valid) {  // EnvironmentMap::pdfDirect, envmap.cpp:540-552 (solid angle)
            pdf = envMap.pdfDirection(envMap.toLocal(dRec.d));
        } else if (hasEnv && dRec.emitter == envIndex()) {  // ConstantBackgroundEmitter::pdfDirect, constant.cpp:216-231 (solid angle)
            const bool hasRefN = !(dRec.refN.x == 0 && dRec.refN.y == 0 && dRec.refN.z == 0);
            pdf = hasRefN ? PPG_INV_PI_F * ppg_max(0.0f, dot(dRec.d, dRec.refN)) : PPG_INV_PI_F * 0.25f;
        } else if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0) {
            if (emitterSphere[dRec.emitter] >= 0) pdf = spherePdfDirect(spheres[emitterSphere[dRec.emitter]], dRec);
            else pdf = emMesh[dRec.emitter].invSurfaceArea * (dRec.dist * dRec.dist) / ppg_abs(dot(dRec.d, dRec.n));
        }
        return pdf * (1.0f * emitterPDF.normalization);  // pdfEmitterDiscrete, scene.h:848-850
    }

    void triBounds(uint32_t t, Point &mn, Point &mx) const {
        mn = Point(std::numeric_limits<Float>::infinity()); mx = Point(-std::numeric_limits<Float>::infinity());
        for (int k = 0; k < 3; ++k) {
            const Point &p = P[idx[3 * t + k]];
            for (int a = 0; a < 3; ++a) { mn[a] = ppg_min(mn[a], p[a]); mx[a] = ppg_max(mx[a], p[a]); }
        }
    }

    int buildRec(int first, int count) {
        BNode n;
        n.bmin = Point(std::numeric_limits<Float>::infinity()); n.bmax = Point(-std::numeric_limits<Float>::infinity());
        Point cmin = n.bmin, cmax = n.bmax;
        for (int i = first; i < first + count; ++i) {
            Point a, b; triBounds(order[i], a, b);
            for (int k = 0; k < 3; ++k) {
                n.bmin[k] = ppg_min(n.bmin[k], a[k]); n.bmax[k] = ppg_max(n.bmax[k], b[k]);
                Float c = 0.5f * (a[k] + b[k]);
                cmin[k] = ppg_min(cmin[k], c); cmax[k] = ppg_max(cmax[k], c);
            }
        }
        Vec ext = aabb.max - aabb.min;
        Float pad = 1e-5f * ppg_max(ppg_max(ext.x, ext.y), ext.z);
        n.bmin = n.bmin - Vec(pad); n.bmax = n.bmax + Vec(pad);
        n.left = n.right = -1; n.first = first; n.count = count;
        int me = (int)bvh.size();
        bvh.push_back(n);
        if (count > 4) {
            Vec ce = cmax - cmin;
            int axis = ce.x > ce.y ? (ce.x > ce.z ? 0 : 2) : (ce.y > ce.z ? 1 : 2);
            int mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) {
                                 Point a0, a1, b0, b1; triBounds(a, a0, a1); triBounds(b, b0, b1);
                                 return a0[axis] + a1[axis] < b0[axis] + b1[axis];
                             });
            int l = buildRec(first, mid - first);
            int r = buildRec(mid, first + count - mid);
            bvh[me].left = l; bvh[me].right = r; bvh[me].count = 0;
        }
        return me;
    }
    void buildBVH() {
        order.resize(nTris());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
        buildRec(0, (int)order.size());
    }

    // TriAccel (triaccel.h:36-199): Wald's pre-projected triangle, the test Mitsuba's kd-tree leaves run
    struct TriAccel {
        uint32_t k;
        Float n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv;
        int load(const Point &A, const Point &B, const Point &C) {  // triaccel.h:62-97
            static const int waldModulo[4] = {1, 2, 0, 1};
            Vec b = C - A, c = B - A, N = cross(c, b);
            k = 0;
            for (int j = 0; j < 3; j++)
                if (ppg_abs(N[j]) > ppg_abs(N[k])) k = j;
            uint32_t u = waldModulo[k], v = waldModulo[k + 1];
            const Float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
            n_u = n_v = n_d = a_u = a_v = b_nu = b_nv = c_nu = c_nv = 0;
            if (denom == 0) { k = 3; return 1; }
            n_u = N[u] / n_k;
            n_v = N[v] / n_k;
            n_d = dot(Vec(A), N) / n_k;
            b_nu = b[u] / denom;
            b_nv = -b[v] / denom;
            a_u = A[u];
            a_v = A[v];
            c_nu = c[v] / denom;
            c_nv = -c[u] / denom;
            return 0;
        }
        bool rayIntersect(const Point &o, const Vec &d, Float mint, Float maxt, Float &u, Float &v, Float &t) const {  // triaccel.h:99-195
            Float o_u, o_v, o_k, d_u, d_v, d_k;
            switch (k) {
                case 0: o_u = o[1]; o_v = o[2]; o_k = o[0]; d_u = d[1]; d_v = d[2]; d_k = d[0]; break;
                case 1: o_u = o[2]; o_v = o[0]; o_k = o[1]; d_u = d[2]; d_v = d[0]; d_k = d[1]; break;
                case 2: o_u = o[0]; o_v = o[1]; o_k = o[2]; d_u = d[0]; d_v = d[1]; d_k = d[2]; break;
                default: return false;
            }
            t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
            if (t < mint || t > maxt) return false;
            const Float hu = o_u + t * d_u - a_u;
            const Float hv = o_v + t * d_v - a_v;
            u = hv * b_nu + hu * b_nv;
            v = hu * c_nu + hv * c_nv;
            return u >= 0 && v >= 0 && u + v <= 1.0f;
        }
    };
    std::vector<TriAccel> accel;
    void buildAccel() {
        accel.resize(nTris());
        for (size_t t = 0; t < nTris(); ++t) accel[t].load(P[idx[3 * t]], P[idx[3 * t + 1]], P[idx[3 * t + 2]]);
    }

    bool triHit(uint32_t t, const Point &o, const Vec &d, Float mint, Float maxt, Float &tt, Float &uu, Float &vv) const {
        return accel[t].rayIntersect(o, d, mint, maxt, uu, vv, tt);
    }

    bool closest(const Point &o, const Vec &d, Float mint, Float maxt, Float &bt, Float &bu, Float &bv, int &bp) const {
        bt = std::numeric_limits<Float>::infinity(); bp = -1;
        auto consider = [&](uint32_t t) {
            Float tt, uu, vv;
            if (triHit(t, o, d, mint, ppg_min(maxt, bt), tt, uu, vv)) {
                if (tt < bt || (tt == bt && (int)t < bp)) { bt = tt; bu = uu; bv = vv; bp = (int)t; }
            }
        };
        if (bvh.empty()) {
            for (uint32_t t = 0; t < nTris(); ++t) consider(t);
        } else {
            int stack[128], sp = 0;
            stack[sp++] = 0;
            while (sp) {
                const BNode &n = bvh[stack[--sp]];
                Float t0 = mint, t1 = ppg_min(maxt, bt);
                bool ok = true;
                for (int a = 0; a < 3 && ok; ++a) {
                    if (d[a] == 0.0f) { ok = (o[a] >= n.bmin[a] && o[a] <= n.bmax[a]); continue; }
                    Float inv = 1.0f / d[a];
                    Float ta = (n.bmin[a] - o[a]) * inv, tb = (n.bmax[a] - o[a]) * inv;
                    if (ta > tb) std::swap(ta, tb);
                    t0 = ppg_max(t0, ta * 0.9999f - 1e-6f); t1 = ppg_min(t1, tb * 1.0001f + 1e-6f);
                    ok = t0 <= t1;
                }
                if (!ok) continue;
                if (n.left < 0) {
                    for (int i = n.first; i < n.first + n.count; ++i) consider(order[i]);
                } else {
                    stack[sp++] = n.left; stack[sp++] = n.right;
                }
            }
        }
        return bp >= 0;
    }

    // Scene::rayIntersect + fillIntersectionRecord (skdtree.cpp:112-142, skdtree.h:343-430, util.cpp:603-608)
    bool rayIntersect(const Point &o, const Vec &d, Float rayMint, Float rayMaxt, Intersection &its) const {
        Float rayMinT = rayMint;
        if (rayMinT == PPG_EPSILON)
            rayMinT *= ppg_max(ppg_max(ppg_max(ppg_abs(o.x), ppg_abs(o.y)), ppg_abs(o.z)), PPG_EPSILON);
        Float t, u, v; int prim;
        its.valid = false;
        its.t = std::numeric_limits<Float>::infinity();
        bool found = closest(o, d, rayMinT, rayMaxt, t, u, v, prim);
        int sph = -1;  // spheres: primitives after the triangles; a tie in t keeps the smaller index
        for (size_t k = 0; k < spheres.size(); ++k) {
            Float ts;
            if (sphereRayIntersect(spheres[k], o, d, rayMinT, found ? ppg_min(rayMaxt, t) : rayMaxt, ts) && (!found || ts < t)) { found = true; t = ts; sph = (int)k; }
        }
        if (!found) return false;
        if (sph >= 0) {
            const ppg_sphere &sp = spheres[sph];
            its.valid = true; its.t = t; its.prim = (int)nTris() + sph;
            Vec n, dpdu;
            sphereFill(sp, o, d, t, its.p, n, dpdu);
            its.geoN = n;
            its.shFrame.n = n;
            its.shFrame.s = normalize(dpdu - n * dot(n, dpdu));  // computeShadingFrame, util.cpp:603-608
            its.shFrame.t = cross(n, its.shFrame.s);
            its.wi = its.shFrame.toLocal(-d);
            its.material = sp.material;
            its.emitter = sp.emitter;
            return true;
        }
        its.valid = true; its.t = t; its.prim = prim;
        const uint32_t i0 = idx[3 * prim], i1 = idx[3 * prim + 1], i2 = idx[3 * prim + 2];
        const Point &p0 = P[i0], &p1 = P[i1], &p2 = P[i2];
        const Vec b(1 - u - v, u, v);
        its.p = p0 * b.x + p1 * b.y + p2 * b.z;
        Vec side1(p1 - p0), side2(p2 - p0);
        Vec faceNormal(cross(side1, side2));
        Float len = length(faceNormal);
        if (!(faceNormal.x == 0 && faceNormal.y == 0 && faceNormal.z == 0)) faceNormal = faceNormal / len;
        Vec shN;
        if (hasNormals) {
            shN = normalize(N[i0] * b.x + N[i1] * b.y + N[i2] * b.z);
            if (dot(faceNormal, shN) < 0) faceNormal = -faceNormal;
        } else {
            shN = faceNormal;
        }
        its.geoN = faceNormal;
        its.material = triMat[prim];
        its.emitter = triEmitter[prim];
        Vec dpdu = side1;
        if (materials[its.material].texture != 0) {
            // a BSDF with a bitmap (usesRayDifferentials): texture coordinates (skdtree.h:403-410) and, on meshes that carry them, the
            // tangents of TriMesh::computeUVTangents (trimesh.cpp:683-735) instead of the triangle edges (skdtree.h:374-381)
            its.uv = Point2{b.y, b.z};
            its.dpdu = side1; its.dpdv = side2;
            if (!UV.empty() && !std::isnan(UV[i0].x) && !std::isnan(UV[i1].x) && !std::isnan(UV[i2].x)) {
                const Point2 &t0 = UV[i0], &t1 = UV[i1], &t2 = UV[i2];
                its.uv = Point2{t0.x * b.x + t1.x * b.y + t2.x * b.z, t0.y * b.x + t1.y * b.y + t2.y * b.z};
                const Float dUV1x = t1.x - t0.x, dUV1y = t1.y - t0.y, dUV2x = t2.x - t0.x, dUV2y = t2.y - t0.y;
                const Vec n = cross(side1, side2);
                const Float nlen = length(n);
                if (nlen != 0) {
                    const Float determinant = dUV1x * dUV2y - dUV1y * dUV2x;
                    if (determinant == 0) {
                        const Vec a = n / nlen;  // coordinateSystem(n / length, dpdu, dpdv), util.cpp:592-601
                        if (ppg_abs(a.x) > ppg_abs(a.y)) {
                            Float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
                            its.dpdv = Vec(a.z * invLen, 0.0f, -a.x * invLen);
                        } else {
                            Float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
                            its.dpdv = Vec(0.0f, a.z * invLen, -a.y * invLen);
                        }
                        its.dpdu = cross(its.dpdv, a);
                    } else {
                        const Float invDet = 1.0f / determinant;
                        its.dpdu = (side1 * dUV2y - side2 * dUV1y) * invDet;
                        its.dpdv = (side1 * -dUV2x + side2 * dUV1x) * invDet;
                    }
                }
            }
            dpdu = its.dpdu;
        }
        // computeShadingFrame(n, dpdu)
        its.shFrame.n = shN;
        its.shFrame.s = normalize(dpdu - shN * dot(shN, dpdu));
        its.shFrame.t = cross(shN, its.shFrame.s);
        its.wi = its.shFrame.toLocal(-d);
        return true;
    }

    // The BSDF instance at an intersection: a bitmap on the diffuse reflectance replaces the constant (m_reflectance->eval(bRec.its),
    // diffuse.cpp:112 / plastic.cpp:258 / roughplastic.cpp:365); the sampling weights of configure() keep the texture's average.
    Material materialAt(const Intersection &its) const {
        Material m = materials[its.material];
        const uint32_t t = m.texture & 0xffffu;
        if (t) {
            const Spectrum c = textures[t - 1].eval(its.uv);
            m.reflectance[0] = c.x; m.reflectance[1] = c.y; m.reflectance[2] = c.z;
        }
        return m;
    }
    // BumpMap::getFrame (bumpmap.cpp:135-160): false = no bump map on this surface
    bool bumpFrame(const Intersection &its, Frame &result) const {
        const uint32_t t = materials[its.material].texture >> 16;
        if (!t) return false;
        Spectrum grad[2];
        textures[t - 1].evalGradient(its.uv, grad);
        const Float dDispDu = luminance(grad[0]), dDispDv = luminance(grad[1]);
        const Vec dpdu = its.dpdu + its.shFrame.n * (dDispDu - dot(its.shFrame.n, its.dpdu));
        const Vec dpdv = its.dpdv + its.shFrame.n * (dDispDv - dot(its.shFrame.n, its.dpdv));
        result.n = normalize(cross(dpdu, dpdv));
        result.s = normalize(dpdu - result.n * dot(result.n, dpdu));
        result.t = cross(result.n, result.s);
        if (dot(result.n, its.geoN) < 0) result.n = result.n * -1.0f;
        return true;
    }

    // AreaLight::eval (area.cpp:104-109) through Intersection::Le(d) (records.inl:56-58)
    Spectrum Le(const Intersection &its, const Vec &d) const {
        if (its.emitter < 0) return Spectrum(0.0f);
        if (dot(its.shFrame.n, d) <= 0) return Spectrum(0.0f);
        const float *r = emitters[its.emitter].radiance;
        return Spectrum(r[0], r[1], r[2]);
    }
};

// Transform::operator()(Point) transform.h:108-125, operator()(Vector) :175-183, transformAffine :128-136
inline Point xfPoint(const float *m, const Point &p) {
    Float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    Float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    Float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w == 1.0f) return Point(x, y, z);
    return Point(x, y, z) / w;
}
inline Vec xfVec(const float *m, const Vec &v) {
    return Vec(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// warp.cpp:81-102 / 43-52
inline Point2 squareToUniformDiskConcentric(const Point2 &sample) {
    Float r1 = 2.0f * sample.x - 1.0f;
    Float r2 = 2.0f * sample.y - 1.0f;
    Float phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = (PPG_PI_F / 4.0f) * (r2 / r1);
    } else {
        r = r2;
        phi = (PPG_PI_F / 2.0f) - (r1 / r2) * (PPG_PI_F / 4.0f);
    }
    Float cosPhi, sinPhi;
    ppg_sincos(phi, &sinPhi, &cosPhi);
    return Point2{r * cosPhi, r * sinPhi};
}
inline Vec squareToCosineHemisphere(const Point2 &sample) {
    Point2 p = squareToUniformDiskConcentric(sample);
    Float z = std::sqrt(ppg_max(0.0f, 1.0f - p.x * p.x - p.y * p.y));  // math::safe_sqrt
    if (z == 0) z = 1e-10f;
    return Vec(p.x, p.y, z);
}
inline Float squareToCosineHemispherePdf(const Vec &d) { return PPG_INV_PI_F * d.z; }  // warp.h

// util.cpp:651-681
inline Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    Float scale = (cosThetaI_ > 0) ? 1 / eta : eta, cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    Float cosThetaI = ppg_abs(cosThetaI_);
    Float cosThetaT = std::sqrt(cosThetaTSqr);
    Float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    Float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}
inline Float fresnelDielectricExt(Float cosThetaI, Float eta) { Float t; return fresnelDielectricExt(cosThetaI, t, eta); }

inline Spectrum safeSqrt(const Spectrum &s) {
    return Spectrum(std::sqrt(ppg_max(0.0f, s.x)), std::sqrt(ppg_max(0.0f, s.y)), std::sqrt(ppg_max(0.0f, s.z)));
}
inline Spectrum cdiv(const Spectrum &a, const Spectrum &b) { return Spectrum(a.x / b.x, a.y / b.y, a.z / b.z); }

// util.cpp:739-761
inline Spectrum fresnelConductorExact(Float cosThetaI, const Spectrum &eta, const Spectrum &k) {
    Float cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
    Spectrum temp1 = mul(eta, eta) - mul(k, k) - Spectrum(sinThetaI2),
             a2pb2 = safeSqrt(mul(temp1, temp1) + mul(mul(mul(k, k), eta), eta) * 4.0f),
             a = safeSqrt((a2pb2 + temp1) * 0.5f);
    Spectrum term1 = a2pb2 + Spectrum(cosThetaI2), term2 = a * (2 * cosThetaI);
    Spectrum Rs2 = cdiv(term1 - term2, term1 + term2);
    Spectrum term3 = a2pb2 * cosThetaI2 + Spectrum(sinThetaI4), term4 = term2 * sinThetaI2;
    Spectrum Rp2 = cdiv(mul(Rs2, term3 - term4), term3 + term4);
    return (Rp2 + Rs2) * 0.5f;
}

// SmoothDiffuse diffuse.cpp:110-150
struct Diffuse {
    static Spectrum eval(const Material &m, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return Spectrum(0.0f);
        return m.R() * (PPG_INV_PI_F * b.wo.z);
    }
    static Float pdf(const Material &, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return 0.0f;
        return squareToCosineHemispherePdf(b.wo);
    }
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &sample) {
        if (b.wi.z <= 0) { pdf = 0.0f; return Spectrum(0.0f); }
        b.wo = squareToCosineHemisphere(sample);
        b.eta = 1.0f;
        b.sampledDelta = false;
        pdf = squareToCosineHemispherePdf(b.wo);
        return m.R();
    }
};

// SmoothConductor conductor.cpp:220-290 (solid-angle measure: eval = pdf = 0)
struct Conductor {
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &) {
        if (b.wi.z <= 0) { pdf = 0.0f; return Spectrum(0.0f); }
        b.wo = Vec(-b.wi.x, -b.wi.y, b.wi.z);  // reflect(wi)
        b.eta = 1.0f;
        b.sampledDelta = true;
        pdf = 1;
        return mul(m.R(), fresnelConductorExact(b.wi.z, m.Eta(), m.K()));
    }
};

// math.cpp:25-72
inline Float mtsErfinv(Float x) {
    Float w = -ppg_log(((Float)1 - x) * ((Float)1 + x));
    Float p;
    if (w < (Float)5) {
        w = w - (Float)2.5;
        p = (Float)2.81022636e-08;
        p = (Float)3.43273939e-07 + p * w;
        p = (Float)-3.5233877e-06 + p * w;
        p = (Float)-4.39150654e-06 + p * w;
        p = (Float)0.00021858087 + p * w;
        p = (Float)-0.00125372503 + p * w;
        p = (Float)-0.00417768164 + p * w;
        p = (Float)0.246640727 + p * w;
        p = (Float)1.50140941 + p * w;
    } else {
        w = std::sqrt(w) - (Float)3;
        p = (Float)-0.000200214257;
        p = (Float)0.000100950558 + p * w;
        p = (Float)0.00134934322 + p * w;
        p = (Float)-0.00367342844 + p * w;
        p = (Float)0.00573950773 + p * w;
        p = (Float)-0.0076224613 + p * w;
        p = (Float)0.00943887047 + p * w;
        p = (Float)1.00167406 + p * w;
        p = (Float)2.83297682 + p * w;
    }
    return p * x;
}
inline Float mtsErf(Float x) {
    Float a1 = (Float)0.254829592, a2 = (Float)-0.284496736, a3 = (Float)1.421413741, a4 = (Float)-1.453152027, a5 = (Float)1.061405429;
    Float p = (Float)0.3275911;
    Float sign = (ppg_f2u(x) >> 31) ? -1.0f : 1.0f;  // math::signum: the FP sign, never zero
    x = ppg_abs(x);
    Float t = (Float)1.0 / ((Float)1.0 + p * x);
    Float y = (Float)1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * ppg_exp(-x * x);
    return sign * y;
}

// MicrofacetDistribution, isotropic GGX or Beckmann with visible-normal sampling (microfacet.h)
struct GGX {
    Float alpha;
    bool beckmann = false;
    Float eval(const Vec &m) const {  // microfacet.h:191-237
        if (m.z <= 0) return 0.0f;
        Float cosTheta2 = m.z * m.z;
        Float beckmannExponent = ((m.x * m.x) / (alpha * alpha) + (m.y * m.y) / (alpha * alpha)) / cosTheta2;
        Float result;
        if (beckmann) {
            result = ppg_exp(-beckmannExponent) / (PPG_PI_F * alpha * alpha * cosTheta2 * cosTheta2);
        } else {
            Float root = ((Float)1 + beckmannExponent) * cosTheta2;
            result = (Float)1 / (PPG_PI_F * alpha * alpha * root * root);
        }
        if (result * m.z < 1e-20f) result = 0;
        return result;
    }
    static Float hypot2(Float a, Float b) {  // math.cpp:74-86
        Float r;
        if (ppg_abs(a) > ppg_abs(b)) { r = b / a; r = ppg_abs(a) * std::sqrt(1.0f + r * r); }
        else if (b != 0.0f) { r = a / b; r = ppg_abs(b) * std::sqrt(1.0f + r * r); }
        else r = 0.0f;
        return r;
    }
    Float smithG1(const Vec &v, const Vec &m) const {  // microfacet.h:476-516
        if (dot(v, m) * v.z <= 0) return 0.0f;
        Float temp = 1 - v.z * v.z;
        Float tanTheta = temp <= 0.0f ? 0.0f : ppg_abs(std::sqrt(temp) / v.z);  // |Frame::tanTheta(v)|, frame.h:122-127
        if (tanTheta == 0.0f) return 1.0f;
        if (beckmann) {
            Float a = 1.0f / (alpha * tanTheta);
            if (a >= 1.6f) return 1.0f;
            Float aSqr = a * a;
            return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
        }
        Float root = alpha * tanTheta;  // projectRoughness: isotropic ⇒ alpha
        return 2.0f / (1.0f + hypot2(1.0f, root));
    }
    Float G(const Vec &wi, const Vec &wo, const Vec &m) const { return smithG1(wi, m) * smithG1(wo, m); }
    Float pdfVisible(const Vec &wi, const Vec &m) const {  // microfacet.h:462-466
        if (wi.z == 0) return 0.0f;
        return smithG1(wi, m) * ppg_abs(dot(wi, m)) * eval(m) / ppg_abs(wi.z);
    }
    static void sampleVisible11Beckmann(Float thetaI, Point2 sample, Float &sx, Float &sy) {  // microfacet.h:565-642
        const Float SQRT_PI_INV = 1 / std::sqrt(PPG_PI_F);
        if (thetaI < 1e-4f) {
            Float sinPhi, cosPhi;
            Float r = std::sqrt(-ppg_log(1.0f - sample.x));
            ppg_sincos(2 * PPG_PI_F * sample.y, &sinPhi, &cosPhi);
            sx = r * cosPhi; sy = r * sinPhi;
            return;
        }
        Float tanThetaI = ppg_tan(thetaI);
        Float cotThetaI = 1 / tanThetaI;
        Float a = -1, c = mtsErf(cotThetaI);
        Float sample_x = ppg_max(sample.x, (Float)1e-6f);
        Float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        Float b = c - (1 + c) * ppg_pow(1 - sample_x, fit);
        Float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * ppg_exp(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            Float invErf = mtsErfinv(b);
            Float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * ppg_exp(-invErf * invErf)) - sample_x;
            Float derivative = normalization * (1 - invErf * tanThetaI);
            if (ppg_abs(value) < 1e-5f) break;
            if (value > 0) c = b;
            else a = b;
            b -= value / derivative;
        }
        sx = mtsErfinv(b);
        sy = mtsErfinv(2.0f * ppg_max(sample.y, (Float)1e-6f) - 1.0f);
    }
    static void sampleVisible11(Float thetaI, Point2 sample, Float &sx, Float &sy) {  // microfacet.h:645-690
        if (thetaI < 1e-4f) {
            Float sinPhi, cosPhi;
            Float r = std::sqrt(ppg_max(0.0f, sample.x / (1 - sample.x)));
            ppg_sincos(2 * PPG_PI_F * sample.y, &sinPhi, &cosPhi);
            sx = r * cosPhi; sy = r * sinPhi;
            return;
        }
        Float tanThetaI = ppg_tan(thetaI);
        Float a = 1 / tanThetaI;
        Float G1 = 2.0f / (1.0f + std::sqrt(ppg_max(0.0f, 1.0f + 1.0f / (a * a))));
        Float A = 2.0f * sample.x / G1 - 1.0f;
        if (ppg_abs(A) == 1) A -= (A < 0 ? -1.0f : 1.0f) * PPG_EPSILON;  // math::signum never returns 0 (math.h:269-277)
        Float tmp = 1.0f / (A * A - 1.0f);
        Float B = tanThetaI;
        Float D = std::sqrt(ppg_max(0.0f, B * B * tmp * tmp - (A * A - B * B) * tmp));
        Float slope_x_1 = B * tmp - D;
        Float slope_x_2 = B * tmp + D;
        sx = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
        Float S;
        if (sample.y > 0.5f) { S = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
        else { S = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
        Float z = (sample.y * (sample.y * (sample.y * (-(Float)0.365728915865723) + (Float)0.790235037209296) - (Float)0.424965825137544) +
                   (Float)0.000152998850436920) /
                  (sample.y * (sample.y * (sample.y * (sample.y * (Float)0.169507819808272 - (Float)0.397203533833404) - (Float)0.232500544458471) +
                               (Float)1) - (Float)0.539825872510702);
        sy = S * z * std::sqrt(1.0f + sx * sx);
    }
    Vec sampleVisible(const Vec &_wi, const Point2 &sample) const {  // microfacet.h:425-460
        Vec wi = normalize(Vec(alpha * _wi.x, alpha * _wi.y, _wi.z));
        Float theta = 0, phi = 0;
        if (wi.z < (Float)0.99999) {
            theta = ppg_acos(wi.z);
            phi = ppg_atan2(wi.y, wi.x);
        }
        Float sinPhi, cosPhi;
        ppg_sincos(phi, &sinPhi, &cosPhi);
        Float sx, sy;
        if (beckmann) sampleVisible11Beckmann(theta, sample, sx, sy);
        else sampleVisible11(theta, sample, sx, sy);
        Float rx = cosPhi * sx - sinPhi * sy, ry = sinPhi * sx + cosPhi * sy;
        rx *= alpha; ry *= alpha;
        Float normalization = (Float)1 / std::sqrt(rx * rx + ry * ry + (Float)1.0);
        return Vec(-rx * normalization, -ry * normalization, normalization);
    }
};

// RoughConductor roughconductor.cpp:247-415 (sampleVisible = true)
struct RoughConductor {
    static Vec reflect(const Vec &wi, const Vec &m) { return m * (2 * dot(wi, m)) - wi; }  // roughconductor.cpp:243-245
    static Spectrum eval(const Material &mt, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return Spectrum(0.0f);
        Vec H = normalize(b.wo + b.wi);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        const Float D = distr.eval(H);
        if (D == 0) return Spectrum(0.0f);
        const Spectrum F = mul(fresnelConductorExact(dot(b.wi, H), mt.Eta(), mt.K()), mt.R());
        const Float G = distr.G(b.wi, b.wo, H);
        Float model = D * G / (4.0f * b.wi.z);
        return F * model;
    }
    static Float pdf(const Material &mt, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return 0.0f;
        Vec H = normalize(b.wo + b.wi);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        return distr.eval(H) * distr.smithG1(b.wi, H) / (4.0f * b.wi.z);
    }
    static Spectrum sample(const Material &mt, BRec &b, Float &pdf, const Point2 &sample) {
        pdf = 0;
        if (b.wi.z < 0) return Spectrum(0.0f);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        Vec m = distr.sampleVisible(b.wi, sample);
        pdf = distr.pdfVisible(b.wi, m);
        if (pdf == 0) return Spectrum(0.0f);
        b.wo = reflect(b.wi, m);
        b.eta = 1.0f;
        b.sampledDelta = false;
        if (b.wo.z <= 0) return Spectrum(0.0f);
        Spectrum F = mul(fresnelConductorExact(dot(b.wi, m), mt.Eta(), mt.K()), mt.R());
        Float weight = distr.smithG1(b.wo, m);
        pdf /= 4.0f * dot(b.wo, m);
        return F * weight;
    }
};

// SmoothPlastic plastic.cpp:247-455
struct Plastic {
    static Spectrum diffTerm(const Material &m) {
        Spectrum diff = m.R();
        if (m.flags & PPG_MAT_NONLINEAR) return cdiv(diff, Spectrum(1.0f) - diff * m.fdrInt);
        return diff / (1 - m.fdrInt);
    }
    static Float probSpecular(const Material &m, Float Fi) {
        return (Fi * m.specularSamplingWeight) / (Fi * m.specularSamplingWeight + (1 - Fi) * (1 - m.specularSamplingWeight));
    }
    static Spectrum eval(const Material &m, const BRec &b) {  // solid-angle measure: the diffuse component
        if (b.wo.z <= 0 || b.wi.z <= 0) return Spectrum(0.0f);
        Float Fi = fresnelDielectricExt(b.wi.z, m.eta[0]);
        Float Fo = fresnelDielectricExt(b.wo.z, m.eta[0]);
        return diffTerm(m) * (squareToCosineHemispherePdf(b.wo) * m.invEta2 * (1 - Fi) * (1 - Fo));
    }
    static Float pdf(const Material &m, const BRec &b) {
        if (b.wo.z <= 0 || b.wi.z <= 0) return 0.0f;
        Float Fi = fresnelDielectricExt(b.wi.z, m.eta[0]);
        return squareToCosineHemispherePdf(b.wo) * (1 - probSpecular(m, Fi));
    }
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &sample) {
        pdf = 0;
        if (b.wi.z <= 0) return Spectrum(0.0f);
        Float Fi = fresnelDielectricExt(b.wi.z, m.eta[0]);
        b.eta = 1.0f;
        Float pS = probSpecular(m, Fi);
        if (sample.x < pS) {
            b.sampledDelta = true;
            b.wo = Vec(-b.wi.x, -b.wi.y, b.wi.z);
            pdf = pS;
            return m.S() * Fi / pS;
        }
        b.sampledDelta = false;
        b.wo = squareToCosineHemisphere(Point2{(sample.x - pS) / (1 - pS), sample.y});
        Float Fo = fresnelDielectricExt(b.wo.z, m.eta[0]);
        pdf = (1 - pS) * squareToCosineHemispherePdf(b.wo);
        return diffTerm(m) * (m.invEta2 * (1 - Fi) * (1 - Fo) / (1 - pS));
    }
};

// SmoothDielectric dielectric.cpp:229-400 (ERadiance mode)
struct Dielectric {
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &sample) {
        Float cosThetaT;
        const Float eta = m.eta[0], invEta = 1 / eta;
        Float F = fresnelDielectricExt(b.wi.z, cosThetaT, eta);
        b.sampledDelta = true;
        if (sample.x <= F) {
            b.wo = Vec(-b.wi.x, -b.wi.y, b.wi.z);
            b.eta = 1.0f;
            pdf = F;
            return m.R();
        }
        Float scale = -(cosThetaT < 0 ? invEta : eta);  // refract(), dielectric.cpp:222-225
        b.wo = Vec(scale * b.wi.x, scale * b.wi.y, cosThetaT);
        b.eta = cosThetaT < 0 ? eta : invEta;
        pdf = 1 - F;
        Float factor = cosThetaT < 0 ? invEta : eta;
        return m.S() * (factor * factor);
    }
};

// RoughDielectric roughdielectric.cpp:268-606 (sampleVisible = true, ERadiance)
struct RoughDielectric {
    static Float signum(Float v) { return (ppg_f2u(v) >> 31) ? -1.0f : 1.0f; }  // math::signum: never zero
    static Spectrum eval(const Material &mt, const BRec &b) {
        if (b.wi.z == 0) return Spectrum(0.0f);
        const Float m_eta = mt.eta[0], m_invEta = 1 / m_eta;
        bool reflect = b.wi.z * b.wo.z > 0;
        Vec H;
        if (reflect) H = normalize(b.wo + b.wi);
        else {
            Float eta = b.wi.z > 0 ? m_eta : m_invEta;
            H = normalize(b.wi + b.wo * eta);
        }
        H = H * signum(H.z);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        const Float D = distr.eval(H);
        if (D == 0) return Spectrum(0.0f);
        const Float F = fresnelDielectricExt(dot(b.wi, H), m_eta);
        const Float G = distr.G(b.wi, b.wo, H);
        if (reflect) {
            Float value = F * D * G / (4.0f * ppg_abs(b.wi.z));
            return mt.R() * value;
        }
        Float eta = b.wi.z > 0.0f ? m_eta : m_invEta;
        Float sqrtDenom = dot(b.wi, H) + eta * dot(b.wo, H);
        Float value = ((1 - F) * D * G * eta * eta * dot(b.wi, H) * dot(b.wo, H)) / (b.wi.z * sqrtDenom * sqrtDenom);
        Float factor = b.wi.z > 0 ? m_invEta : m_eta;
        return mt.S() * ppg_abs(value * factor * factor);
    }
    static Float pdf(const Material &mt, const BRec &b) {
        const Float m_eta = mt.eta[0], m_invEta = 1 / m_eta;
        bool reflect = b.wi.z * b.wo.z > 0;
        Vec H;
        Float dwh_dwo;
        if (reflect) {
            H = normalize(b.wo + b.wi);
            dwh_dwo = 1.0f / (4.0f * dot(b.wo, H));
        } else {
            Float eta = b.wi.z > 0 ? m_eta : m_invEta;
            H = normalize(b.wi + b.wo * eta);
            Float sqrtDenom = dot(b.wi, H) + eta * dot(b.wo, H);
            dwh_dwo = (eta * eta * dot(b.wo, H)) / (sqrtDenom * sqrtDenom);
        }
        H = H * signum(H.z);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        Float prob = distr.pdfVisible(b.wi * signum(b.wi.z), H);
        Float F = fresnelDielectricExt(dot(b.wi, H), m_eta);
        prob *= reflect ? F : (1 - F);
        return ppg_abs(prob * dwh_dwo);
    }
    static Spectrum sample(const Material &mt, BRec &b, Float &pdf, const Point2 &sample, Sampler *sampler) {
        const Float m_eta = mt.eta[0], m_invEta = 1 / m_eta;
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        pdf = 0;
        b.sampledDelta = false;
        const Vec wis = b.wi * signum(b.wi.z);
        const Vec m = distr.sampleVisible(wis, sample);
        Float microfacetPDF = distr.pdfVisible(wis, m);
        if (microfacetPDF == 0) return Spectrum(0.0f);
        pdf = microfacetPDF;
        Float cosThetaT;
        Float F = fresnelDielectricExt(dot(b.wi, m), cosThetaT, m_eta);
        Spectrum weight(1.0f);
        bool sampleReflection = true;
        if (sampler->next1D() > F) { sampleReflection = false; pdf *= 1 - F; }
        else pdf *= F;
        Float dwh_dwo;
        if (sampleReflection) {
            b.wo = m * (2 * dot(b.wi, m)) - b.wi;  // reflect(wi, m), util.cpp:763-765
            b.eta = 1.0f;
            if (b.wi.z * b.wo.z <= 0) return Spectrum(0.0f);
            weight = mul(weight, mt.R());
            dwh_dwo = 1.0f / (4.0f * dot(b.wo, m));
        } else {
            if (cosThetaT == 0) return Spectrum(0.0f);
            Float eta = m_eta;  // refract(wi, m, eta, cosThetaT), util.cpp:767-772
            if (cosThetaT < 0) eta = 1 / eta;
            b.wo = m * (dot(b.wi, m) * eta + cosThetaT) - b.wi * eta;
            b.eta = cosThetaT < 0 ? m_eta : m_invEta;
            if (b.wi.z * b.wo.z >= 0) return Spectrum(0.0f);
            Float factor = cosThetaT < 0 ? m_invEta : m_eta;
            weight = mul(weight, mt.S() * (factor * factor));
            Float sqrtDenom = dot(b.wi, m) + b.eta * dot(b.wo, m);
            dwh_dwo = (b.eta * b.eta * dot(b.wo, m)) / (sqrtDenom * sqrtDenom);
        }
        weight = weight * distr.smithG1(b.wo, m);
        pdf *= ppg_abs(dwh_dwo);
        return weight;
    }
};

// evalCubicInterp1D (spline.cpp:23-60) on [min, max] = [0, 1], extrapolate = false
inline Float evalCubicInterp1D(Float x, const Float *values, size_t size) {
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    Float t = ((x - 0.0f) * (Float)(size - 1)) / (1.0f - 0.0f);
    size_t k = std::max((size_t)0, std::min((size_t)t, size - 2));
    Float f0 = values[k], f1 = values[k + 1], d0, d1;
    if (k > 0) d0 = 0.5f * (values[k + 1] - values[k - 1]);
    else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = 0.5f * (values[k + 2] - values[k]);
    else d1 = values[k + 1] - values[k];
    t = t - (Float)k;
    Float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

// RoughPlastic roughplastic.cpp:330-501 (constant alpha, sampleVisible = true, both components requested)
struct RoughPlastic {
    // m_externalRoughTransmittance->eval(cosTheta, alpha) with eta and alpha fixed (rtrans.h:185-196, 233)
    static Float T(const Material &m, Float cosTheta) {
        Float warpedCosTheta = ppg_pow(ppg_abs(cosTheta), 0.25f);
        if (!(cosTheta >= 0)) return 0.0f;
        Float result = evalCubicInterp1D(warpedCosTheta, m.rt, m.rtN);
        return ppg_min(1.0f, ppg_max(0.0f, result));
    }
    static Float probSpecular(const Material &m, Float cosThetaI) {  // roughplastic.cpp:406-412
        Float probSpecular = 1 - T(m, cosThetaI);
        return (probSpecular * m.specularSamplingWeight) / (probSpecular * m.specularSamplingWeight + (1 - probSpecular) * (1 - m.specularSamplingWeight));
    }
    static Spectrum eval(const Material &mt, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return Spectrum(0.0f);
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        const Vec H = normalize(b.wo + b.wi);
        const Float D = distr.eval(H);
        const Float F = fresnelDielectricExt(dot(b.wi, H), mt.eta[0]);
        const Float G = distr.G(b.wi, b.wo, H);
        Float value = F * D * G / (4.0f * b.wi.z);
        Spectrum result = mt.S() * value;
        Spectrum diff = mt.R();
        Float T12 = T(mt, b.wi.z), T21 = T(mt, b.wo.z);
        Float Fdr = mt.fdrInt;  // 1 - m_internalRoughTransmittance->evalDiffuse(alpha)
        if (mt.flags & PPG_MAT_NONLINEAR) diff = cdiv(diff, Spectrum(1.0f) - diff * Fdr);
        else diff = diff / (1 - Fdr);
        return result + diff * (PPG_INV_PI_F * b.wo.z * T12 * T21 * mt.invEta2);
    }
    static Float pdf(const Material &mt, const BRec &b) {
        if (b.wi.z <= 0 || b.wo.z <= 0) return 0.0f;
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        const Vec H = normalize(b.wo + b.wi);
        Float pS = probSpecular(mt, b.wi.z), pD = 1 - pS;
        const Float dwh_dwo = 1.0f / (4.0f * dot(b.wo, H));
        const Float prob = distr.pdfVisible(b.wi, H);
        Float result = prob * dwh_dwo * pS;
        result += pD * squareToCosineHemispherePdf(b.wo);
        return result;
    }
    static Spectrum sample(const Material &mt, BRec &b, Float &pdf_, const Point2 &sample_) {
        pdf_ = 0;
        if (b.wi.z <= 0) return Spectrum(0.0f);
        Point2 sample = sample_;
        GGX distr{mt.alpha, (mt.flags & PPG_MAT_BECKMANN) != 0};
        bool choseSpecular = true;
        Float pS = probSpecular(mt, b.wi.z);
        if (sample.y < pS) sample.y /= pS;
        else { sample.y = (sample.y - pS) / (1 - pS); choseSpecular = false; }
        b.sampledDelta = false;
        if (choseSpecular) {
            Vec m = distr.sampleVisible(b.wi, sample);
            b.wo = RoughConductor::reflect(b.wi, m);
            if (b.wo.z <= 0) return Spectrum(0.0f);
        } else {
            b.wo = squareToCosineHemisphere(sample);
        }
        b.eta = 1.0f;
        pdf_ = pdf(mt, b);
        if (pdf_ == 0) return Spectrum(0.0f);
        return eval(mt, b) / pdf_;
    }
};

// ThinDielectric thindielectric.cpp:152-252
struct ThinDielectric {
    static Float R(const Material &m, Float cosThetaI) {  // incl. internal reflections: R' = R + TRT + TR^3T + ..
        Float R = fresnelDielectricExt(ppg_abs(cosThetaI), m.eta[0]), T = 1 - R;
        if (R < 1) R += T * T * R / (1 - R * R);
        return R;
    }
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &sample) {
        Float R = ThinDielectric::R(m, b.wi.z);
        b.eta = 1.0f;
        b.sampledDelta = true;
        if (sample.x <= R) {
            b.sampledNull = false;
            b.wo = Vec(-b.wi.x, -b.wi.y, b.wi.z);
            pdf = R;
            return m.R();
        }
        b.sampledNull = true;
        b.wo = -b.wi;  // transmit()
        pdf = 1 - R;
        return m.S();
    }
    // eval(BSDFSamplingRecord(its, -wo, wo), EDiscrete) with typeMask = ENull: what a ray passing straight through keeps
    static Spectrum evalNull(const Material &m, Float cosThetaI) { return m.S() * (1 - R(m, cosThetaI)); }
};

// BSDF dispatch: getType / eval / pdf / sample (solid-angle measure) of the supported plugins, incl. the TwoSided adapter
struct BSDF {
    static bool isSmooth(const Material &m) {  // getType() & ESmooth (diffuse or glossy components)
        return m.type == PPG_BSDF_DIFFUSE || m.type == PPG_BSDF_ROUGHCONDUCTOR || m.type == PPG_BSDF_PLASTIC || m.type == PPG_BSDF_ROUGHDIELECTRIC ||
               m.type == PPG_BSDF_ROUGHPLASTIC;
    }
    static bool allDelta(const Material &m) { return !isSmooth(m); }  // (type & EDelta) == (type & EAll)
    // getType() & (ETransmission | EBackSide): twosided sets EBackSide (twosided.cpp:97-101), the dielectric both
    static bool hasBackSideOrTransmission(const Material &m) {
        return (m.flags & (PPG_MAT_TWOSIDED | PPG_MAT_MASK)) || m.type == PPG_BSDF_DIELECTRIC || m.type == PPG_BSDF_THINDIELECTRIC ||
               m.type == PPG_BSDF_ROUGHDIELECTRIC;  // mask: mask.cpp:103
    }
    static bool hasNull(const Material &m) { return m.masked() || m.type == PPG_BSDF_THINDIELECTRIC; }  // getType() & ENull
    static Spectrum evalNull(const Material &m, Float cosThetaI) {  // eval(bRec(its, -wo, wo), EDiscrete), typeMask = ENull
        if (m.masked()) return Spectrum(1.0f) - m.Opacity();  // mask.cpp:115-116
        return ThinDielectric::evalNull(m, cosThetaI);
    }

    static Spectrum evalOne(const Material &m, const BRec &b) {
        switch (m.type) {
            case PPG_BSDF_DIFFUSE: return Diffuse::eval(m, b);
            case PPG_BSDF_ROUGHCONDUCTOR: return RoughConductor::eval(m, b);
            case PPG_BSDF_PLASTIC: return Plastic::eval(m, b);
            case PPG_BSDF_ROUGHPLASTIC: return RoughPlastic::eval(m, b);
            case PPG_BSDF_ROUGHDIELECTRIC: return RoughDielectric::eval(m, b);
            default: return Spectrum(0.0f);  // delta components are zero for the solid-angle measure (conductor.cpp:222-237)
        }
    }
    static Float pdfOne(const Material &m, const BRec &b) {
        switch (m.type) {
            case PPG_BSDF_DIFFUSE: return Diffuse::pdf(m, b);
            case PPG_BSDF_ROUGHCONDUCTOR: return RoughConductor::pdf(m, b);
            case PPG_BSDF_PLASTIC: return Plastic::pdf(m, b);
            case PPG_BSDF_ROUGHPLASTIC: return RoughPlastic::pdf(m, b);
            case PPG_BSDF_ROUGHDIELECTRIC: return RoughDielectric::pdf(m, b);
            default: return 0.0f;
        }
    }
    static Spectrum sampleOne(const Material &m, BRec &b, Float &pdf, const Point2 &sample, Sampler *sampler) {
        switch (m.type) {
            case PPG_BSDF_DIFFUSE: return Diffuse::sample(m, b, pdf, sample);
            case PPG_BSDF_MIRROR:
            case PPG_BSDF_CONDUCTOR: return Conductor::sample(m, b, pdf, sample);
            case PPG_BSDF_ROUGHCONDUCTOR: return RoughConductor::sample(m, b, pdf, sample);
            case PPG_BSDF_PLASTIC: return Plastic::sample(m, b, pdf, sample);
            case PPG_BSDF_ROUGHPLASTIC: return RoughPlastic::sample(m, b, pdf, sample);
            case PPG_BSDF_DIELECTRIC: return Dielectric::sample(m, b, pdf, sample);
            case PPG_BSDF_THINDIELECTRIC: return ThinDielectric::sample(m, b, pdf, sample);
            case PPG_BSDF_ROUGHDIELECTRIC: return RoughDielectric::sample(m, b, pdf, sample, sampler);
            default: pdf = 0; return Spectrum(0.0f);
        }
    }
    static bool twoSided(const Material &m) {
        return (m.flags & PPG_MAT_TWOSIDED) && m.type != PPG_BSDF_DIELECTRIC && m.type != PPG_BSDF_THINDIELECTRIC && m.type != PPG_BSDF_ROUGHDIELECTRIC;
    }

    static Spectrum evalTS(const Material &m, const BRec &b) {
        if (!twoSided(m) || b.wi.z > 0) return evalOne(m, b);
        BRec c = b;  // twosided.cpp:120-135: flip both directions onto the front side
        c.wi.z *= -1; c.wo.z *= -1;
        return evalOne(m, c);
    }
    static Float pdfTS(const Material &m, const BRec &b) {
        if (!twoSided(m) || b.wi.z > 0) return pdfOne(m, b);
        BRec c = b;
        c.wi.z *= -1; c.wo.z *= -1;
        return pdfOne(m, c);
    }
    // BumpMap adapter (bumpmap.cpp:162-219): the outermost one; queries are re-expressed in the perturbed frame
    static BRec perturbed(const BRec &b) {
        BRec q = b;
        q.bumpSh = q.bumpPert = nullptr;
        q.wi = b.bumpPert->toLocal(b.bumpSh->toWorld(b.wi));
        q.wo = b.bumpPert->toLocal(b.bumpSh->toWorld(b.wo));
        return q;
    }
    static Spectrum eval(const Material &m, const BRec &b) {
        if (!b.bumpPert) return evalMask(m, b);
        const BRec q = perturbed(b);
        if (b.wo.z * q.wo.z <= 0) return Spectrum(0.0f);
        return evalMask(m, q);
    }
    static Float pdf(const Material &m, const BRec &b) {
        if (!b.bumpPert) return pdfMask(m, b);
        const BRec q = perturbed(b);
        if (b.wo.z * q.wo.z <= 0) return 0;
        return pdfMask(m, q);
    }
    static Spectrum sample(const Material &m, BRec &b, Float &pdf, const Point2 &sample, Sampler *sampler) {
        if (!b.bumpPert) return sampleMask(m, b, pdf, sample, sampler);
        BRec q = b;
        q.bumpSh = q.bumpPert = nullptr;
        q.wi = b.bumpPert->toLocal(b.bumpSh->toWorld(b.wi));
        Spectrum result = sampleMask(m, q, pdf, sample, sampler);
        if (!isZero(result)) {
            b.sampledDelta = q.sampledDelta; b.sampledNull = q.sampledNull;
            b.wo = b.bumpSh->toLocal(b.bumpPert->toWorld(q.wo));
            b.eta = q.eta;
            if (b.wo.z * q.wo.z <= 0) return Spectrum(0.0f);
        }
        return result;
    }
    // Mask adapter (mask.cpp:108-214), solid-angle measure for eval / pdf
    static Spectrum evalMask(const Material &m, const BRec &b) {
        Spectrum r = evalTS(m, b);
        return m.masked() ? mul(r, m.Opacity()) : r;
    }
    static Float pdfMask(const Material &m, const BRec &b) {
        Float r = pdfTS(m, b);
        return m.masked() ? r * luminance(m.Opacity()) : r;
    }
    static Spectrum sampleMask(const Material &m, BRec &b, Float &pdf, const Point2 &_sample, Sampler *sampler) {
        if (!m.masked()) return sampleTS(m, b, pdf, _sample, sampler);
        Point2 sample(_sample);
        Spectrum opacity = m.Opacity();
        Float prob = luminance(opacity);
        if (sample.x < prob) {
            sample.x /= prob;
            Spectrum result = mul(sampleTS(m, b, pdf, sample, sampler), opacity) / prob;
            pdf *= prob;
            return result;
        }
        b.wo = -b.wi;
        b.eta = 1.0f;
        b.sampledDelta = true;
        b.sampledNull = true;
        pdf = 1 - prob;
        return (Spectrum(1.0f) - opacity) / pdf;
    }
    static Spectrum sampleTS(const Material &m, BRec &b, Float &pdf, const Point2 &sample, Sampler *sampler) {
        if (!twoSided(m)) return sampleOne(m, b, pdf, sample, sampler);
        bool flipped = false;  // twosided.cpp:160-180
        if (b.wi.z < 0) { b.wi.z *= -1; flipped = true; }
        Spectrum result = sampleOne(m, b, pdf, sample, sampler);
        if (flipped) {
            b.wi.z *= -1;
            if (!isZero(result) && pdf != 0) b.wo.z *= -1;
        }
        return result;
    }
};

// ------------------------------------------------------------------------------------------------
// GuidedPathTracer GP:1012-2419
// ------------------------------------------------------------------------------------------------
struct PathCounters {
    uint64_t rays = 0, pathLen = 0, committed = 0;  // rays includes shadow rays
};

class GuidedPathTracer {
public:
    // properties, GP:1014-1085
    ENee m_nee = ENever;
    ESampleCombination m_sampleCombination = EDiscardWithAutomaticBudget;
    ESpatialFilter m_spatialFilter = ESNearest;
    EDirectionalFilter m_directionalFilter = EDNearest;
    ELoss m_bsdfSamplingFractionLoss = ENone;
    int m_sdTreeMaxMemory = -1, m_sTreeThreshold = 12000;
    Float m_dTreeThreshold = 0.01f, m_bsdfSamplingFraction = 0.5f;
    int m_sppPerPass = 4;
    EBudget m_budgetType = ESeconds;
    Float m_budget = 300.0f;
    bool m_dumpSDTree = false;
    int m_rrDepth = 5, m_maxDepth = -1;
    bool m_strictNormals = false, m_hideEmitters = false;
    uint64_t m_seed = 0;
    std::string m_dumpPrefix;

    Modes modes;
    int threads = 1;
    Scene scene;
    bool haveScene = false;
    int shardRank = 0, shardWorld = 1, tileSize = 32;

    std::unique_ptr<STree> m_sdTree;
    bool m_doNee = false, m_isBuilt = false, m_isFinalIter = false;
    int m_iter = 0, m_passesRendered = 0, m_passesRenderedThisIter = 0;
    std::vector<Float> m_image, m_squaredImage;   // RGB sums of the current performRenderPasses
    std::vector<Float> m_imageW;                  // weights (sample counts)
    std::vector<Float> m_film, m_filmW;           // film accumulators (cleared per iteration)
    std::vector<Float> m_varianceBuffer;
    std::vector<std::vector<Float>> m_images;     // weight-normalised copies, inverse-variance mode
    std::vector<Float> m_variances;
    std::chrono::steady_clock::time_point m_startTime, m_passStart;
    int m_passesLocal = 0;
    PathCounters m_counters;
    uint64_t m_work[CNT_N] = {};
    uint64_t m_lenHist[PPGO_LEN_HIST] = {};  // paths by final rRec.depth (last bin: that or longer) — sizes the GPU's tail phase
    volatile bool cancelled = false;
    bool m_cancelSeen = false;  // the render under way has acted on the flag: beginRender does not apply it to the next one (the product's rule)
    bool seesCancel() { if (!cancelled) return false; m_cancelSeen = true; return true; }
    ppg_pass_hook passHook = nullptr;
    void *passHookUser = nullptr;
    ppg_stop_hook stopHook = nullptr;
    void *stopHookUser = nullptr;
    std::string error;

    int W() const { return scene.cam.width; }
    int H() const { return scene.cam.height; }

    // final iteration in groups of passes (include/ppg.h "Final iteration: groups of passes"): where renderOnePass accumulates, whether it
    // renders the whole film (a rank's own groups of a sharded render), the slots, the state of their exchange
    Float *m_accImage = nullptr, *m_accSq = nullptr, *m_accW = nullptr;
    bool m_allPixels = false;
    std::vector<Float> m_partials;   // [film 3 n][film weights n][groups x (image 3 n, squared image 3 n, weights n)]
    bool m_partialsPending = false, m_partialsExported = false;
    unsigned int m_pendingGroups = 0;
    uint64_t m_samplesLocal = 0;
    static int finalGroupPasses(int n) { n = std::max(1, n); return 16 * ((n + 1023) / 1024); }

    bool ownsPixel(int x, int y) const {
        if (shardWorld <= 1 || m_allPixels) return true;
        int tilesX = (W() + tileSize - 1) / tileSize;
        int t = (y / tileSize) * tilesX + (x / tileSize);
        return t % shardWorld == shardRank;
    }

    void resetSDTree() {  // GP:1108-1113
        // std::pow(2, m_iter) is a double power; the whole expression is double, then truncated to size_t
        double thr = std::sqrt(std::ldexp(1.0, m_iter) * m_sppPerPass / 4) * m_sTreeThreshold;
        m_sdTree->refine((size_t)thr, m_sdTreeMaxMemory);
        m_sdTree->forEachDTreeWrapper([this](DTreeWrapper *dTree) { dTree->reset(20, m_dTreeThreshold); });
    }

    void buildSDTree(ppg_tree_stats *st) {  // GP:1115-1189
        m_sdTree->forEachDTreeWrapper([this](DTreeWrapper *dTree) { dTree->build(modes); });
        int maxDepth = 0, minDepth = std::numeric_limits<int>::max();
        Float avgDepth = 0, maxAvgRadiance = 0, minAvgRadiance = std::numeric_limits<Float>::max(), avgAvgRadiance = 0;
        size_t maxNodes = 0, minNodes = std::numeric_limits<size_t>::max();
        Float avgNodes = 0, maxStatisticalWeight = 0, minStatisticalWeight = std::numeric_limits<Float>::max(),
              avgStatisticalWeight = 0;
        int nPoints = 0, nPointsNodes = 0;
        uint64_t totalNodes = 0;
        m_sdTree->forEachDTreeWrapper([&](const DTreeWrapper *dTree) {
            const int depth = dTree->depth();
            maxDepth = std::max(maxDepth, depth);
            minDepth = std::min(minDepth, depth);
            avgDepth += depth;
            const Float avgRadiance = dTree->meanRadiance();
            maxAvgRadiance = ppg_max(maxAvgRadiance, avgRadiance);
            minAvgRadiance = ppg_min(minAvgRadiance, avgRadiance);
            avgAvgRadiance += avgRadiance;
            if (dTree->numNodes() > 1) {
                const size_t nodes = dTree->numNodes();
                maxNodes = std::max(maxNodes, nodes);
                minNodes = std::min(minNodes, nodes);
                avgNodes += nodes;
                ++nPointsNodes;
            }
            totalNodes += dTree->numNodes();
            const Float statisticalWeight = dTree->statisticalWeight();
            maxStatisticalWeight = ppg_max(maxStatisticalWeight, statisticalWeight);
            minStatisticalWeight = ppg_min(minStatisticalWeight, statisticalWeight);
            avgStatisticalWeight += statisticalWeight;
            ++nPoints;
        });
        if (nPoints > 0) {
            avgDepth /= nPoints;
            avgAvgRadiance /= nPoints;
            if (nPointsNodes > 0) avgNodes /= nPointsNodes;
            avgStatisticalWeight /= nPoints;
        }
        if (st) {
            st->min_depth = minDepth; st->max_depth = maxDepth; st->avg_depth = avgDepth;
            st->min_mean_radiance = minAvgRadiance; st->avg_mean_radiance = avgAvgRadiance; st->max_mean_radiance = maxAvgRadiance;
            st->min_nodes = minNodes; st->max_nodes = maxNodes; st->avg_nodes = avgNodes;
            st->min_stat_weight = minStatisticalWeight; st->avg_stat_weight = avgStatisticalWeight;
            st->max_stat_weight = maxStatisticalWeight;
            st->n_leaves = (uint32_t)nPoints; st->n_stree_nodes = (uint32_t)m_sdTree->nodes().size();
            st->n_dtree_nodes = totalNodes;
        }
        m_isBuilt = true;
    }

    void dumpSDTree(const char *path) {  // GP:1191-1208 (+ DTreeWrapper::dump)
        FILE *f = fopen(path, "wb");
        if (!f) return;
        fwrite(scene.cam.camera_to_world, 4, 16, f);
        m_sdTree->dump(f);
        fclose(f);
    }

    // ---- performRenderPasses GP:1210-1329, split so that a sharded driver can reduce in between ----
    void renderPassesNoStat(int numPasses) {
        std::fill(m_image.begin(), m_image.end(), 0.0f);          // GP:1217-1218
        std::fill(m_squaredImage.begin(), m_squaredImage.end(), 0.0f);
        std::fill(m_imageW.begin(), m_imageW.end(), 0.0f);
        m_passStart = std::chrono::steady_clock::now();
        m_passesLocal = 0;
        m_counters = PathCounters();
        m_samplesLocal = 0;
        m_partialsPending = false; m_partialsExported = false;
        if (m_isFinalIter && m_budgetType != ESeconds && numPasses > 0) { renderFinalGroups(numPasses); return; }
        // ROUND mode: the passes are rendered in rounds of ppg_adam_round_passes() passes; the sampling fractions are frozen during
        // a round and its records are applied afterwards (applyAdamRound).  The time budget is checked once per round then
        // (the reference checks after every finished pass of a batch of up to 128 scheduled ones, GP:1235-1266).
        const bool rounds = m_bsdfSamplingFractionLoss != ENone && modes.adam != PPGO_ADAM_SEQUENTIAL && m_isBuilt && !m_isFinalIter;
        // PPGO_ADAM_HALF (measurement only): in iteration 1 — two one-pass rounds — the optimiser is also applied after each HALF
        // pass (pixels by parity of x + y), i.e. four rounds instead of two
        const bool halves = rounds && modes.adam == PPGO_ADAM_HALF && numPasses == 2;
        // PPGO_ADAM_REGIONS + R (measurement only): iterations of up to 16 passes render every pass in R groups of blocks in spiral order, the
        // optimiser applied after each
        const int regions = (rounds && modes.adam > PPGO_ADAM_REGIONS + 1 && numPasses <= PPG_ADAM_REGION_MAX_PASSES)
                                ? std::min(modes.adam - PPGO_ADAM_REGIONS, ((W() + 31) / 32) * ((H() + 31) / 32)) : 0;
        const int roundPasses = regions ? 1 : (rounds ? adamRoundPasses(numPasses) : 1);
        // include/ppg.h "STRAGGLERS": the rounds of a render of unbounded paths whose record positions are known in advance
        m_deferRound = rounds && m_maxDepth < 0 && m_spatialFilter != ESBox && !(m_doNee && m_nee == EKickstart);
        m_deferredRecords.clear();  // (nothing is carried from one call to the next)
        // (cancelled with a round hook installed: the remaining rounds are entered empty, so that this rank stays in step with the hooks of
        // the others until they have all seen its status — the product's rule, ppg_hip.hip renderPassesNoStat)
        bool drain = false;
        m_hookFailed = false;
        for (int i = 0; i < numPasses;) {
            if (seesCancel()) {
                if (rounds && passHook) drain = true;
                else {  // (a sharded time budget: meet the other ranks in the stop hook they are about to ask — the product's rule)
                    if (m_budgetType == ESeconds && stopHook) (void)stopHook(stopHookUser, PPG_STOP_CANCELLED);
                    break;
                }
            }
            const int n = std::min(roundPasses, numPasses - i);
            m_roundStartPass = m_passesRendered;
            for (int k = 0; k < n; ++k) {
                if (regions) {  // (a cancelled rank kept in step enters every group's round too, empty: the hook calls must match across ranks)
                    m_regions = regions;
                    for (m_region = 0; m_region < regions; ++m_region) { if (!drain) renderOnePass(); if (m_region + 1 < regions) applyAdamRound(); }
                    m_regions = 0; m_region = -1;
                    if (drain) continue;
                    ++m_passesRendered; ++m_passesRenderedThisIter; ++m_passesLocal;
                    m_samplesLocal += ownedPixels() * (uint64_t)m_sppPerPass;
                    continue;
                }
                if (drain) continue;
                if (halves) { m_parity = 0; renderOnePass(); applyAdamRound(); m_parity = 1; renderOnePass(); m_parity = -1; }
                else
                renderOnePass();
                ++m_passesRendered; ++m_passesRenderedThisIter; ++m_passesLocal;
                m_samplesLocal += ownedPixels() * (uint64_t)m_sppPerPass;
            }
            i += n;
            if (rounds) applyAdamRound();
            if (m_hookFailed) break;
            if (m_budgetType == ESeconds) {  // GP:1259-1262: whole seconds; sharded: rank 0's decision for all (include/ppg.h ppg_set_stop_hook)
                int stop = (int)computeElapsedSeconds(m_startTime) > m_budget ? 1 : 0;
                if (stopHook) stop = stopHook(stopHookUser, stop);
                if (stop) break;
            }
        }
        // the records of the last round's stragglers: a round of their own (sharded: the round hook is called for it on every rank)
        if (m_deferRound && !m_hookFailed && (passHook || !m_deferredRecords.empty())) applyAdamRound();
        m_deferRound = false;
    }
    bool m_deferRound = false;
    int m_deferDepth = PPG_ADAM_DEFER_DEPTH;  // (ppgo_debug_set_defer_depth: the tests' switch, include/ppg_testhooks.h)

    uint64_t ownedPixels() const {
        uint64_t c = 0;
        for (int y = 0; y < H(); ++y) for (int x = 0; x < W(); ++x) if (ownsPixel(x, y)) ++c;
        return c;
    }
    // image += slot, squared image += slot, weights += slot, film += slot's image, film weights += slot's weights
    void addGroup(const Float *slot) {
        const size_t n = (size_t)W() * H();
        for (size_t i = 0; i < 3 * n; ++i) { m_image[i] += slot[i]; m_squaredImage[i] += slot[3 * n + i]; m_film[i] += slot[i]; }
        for (size_t i = 0; i < n; ++i) { m_imageW[i] += slot[6 * n + i]; m_filmW[i] += slot[6 * n + i]; }
    }
    // The passes of a final iteration: groups of finalGroupPasses(numPasses) passes, each summed from zero, added in group order; rank r of a
    // sharded render renders groups r, r + world, ... over the whole film — or, when there are fewer than two groups per rank, every group on
    // its own tiles (the product's rule, ppg_hip.hip finalGroupsByRank) — and leaves its slots for the exchange.
    void renderFinalGroups(int numPasses) {
        const int G = finalGroupPasses(numPasses), nGroups = (numPasses + G - 1) / G;
        const size_t n = (size_t)W() * H();
        m_partials.assign(4 * n + (size_t)nGroups * 7 * n, 0.0f);
        const int firstPass = m_passesRendered;
        const bool byRank = shardWorld > 1 && nGroups >= 2 * shardWorld;
        m_allPixels = byRank;
        const uint64_t pixelsMine = byRank || shardWorld <= 1 ? (uint64_t)n : ownedPixels();
        for (int g = byRank ? shardRank : 0; g < nGroups && !seesCancel(); g += byRank ? shardWorld : 1) {
            Float *slot = m_partials.data() + 4 * n + (size_t)g * 7 * n;
            m_accImage = slot; m_accSq = slot + 3 * n; m_accW = slot + 6 * n;
            const int cnt = std::min(G, numPasses - g * G);
            for (int k = 0; k < cnt; ++k) {
                m_passesRendered = firstPass + g * G + k;
                renderOnePass();
                m_samplesLocal += pixelsMine * (uint64_t)m_sppPerPass;
            }
            if (shardWorld <= 1) addGroup(slot);
        }
        m_accImage = m_accSq = m_accW = nullptr;
        m_allPixels = false;
        m_passesRendered = firstPass + numPasses; m_passesRenderedThisIter += numPasses; m_passesLocal += numPasses;
        if (shardWorld > 1) { m_partialsPending = true; m_pendingGroups = (unsigned int)nGroups; }
    }

    // passes per round (include/ppg.h, ppg_adam_round_passes): doubled while it stays within 16 passes, half the call's passes and
    // 2^24 paths of the whole image
    int adamRoundPasses(int numPasses) const {
        const uint64_t perPass = (uint64_t)m_sppPerPass * (uint64_t)W() * (uint64_t)H();
        int r = 1;
        for (;;) {
            const int next = 2 * r;
            if (next > PPG_ADAM_ROUND_MAX_PASSES || 2 * next > numPasses || (uint64_t)next * perPass > PPG_ADAM_ROUND_MAX_PATHS) break;
            r = next;
        }
        return r;
    }

    // ---- the deferred Adam steps of one round (adam_mode = ROUND; include/ppg.h "Learning the BSDF sampling fraction") ----
    int m_roundStartPass = 0;
    std::vector<std::vector<AdamRecord>> m_blockSinks;  // one per image block: filled by whichever thread renders the block
    struct PackedAdamRecord { uint64_t key; float product, woPdf, bsdfPdf, dTreePdf, weight, pad; };  // = ppg_adam_record
    std::vector<PackedAdamRecord> m_adamRecords, m_deferredRecords;
    void applyAdamRound() {
        auto &nodes = m_sdTree->nodes();
        m_adamRecords.clear();
        std::vector<PackedAdamRecord> late;  // this round's stragglers' records (include/ppg.h "STRAGGLERS"): applied with the next round
        for (auto &sink : m_blockSinks) {
            for (const AdamRecord &r : sink) {
                const uint64_t leaf = (uint64_t)(((const char *)r.dTree - (const char *)&nodes[0].dTree) / sizeof(STreeNode));
                const PackedAdamRecord pr{(leaf << PPG_ADAM_LEAF_SHIFT) | ((uint64_t)r.path << PPG_ADAM_CODE_BITS) | r.code,
                                          r.product, r.woPdf, r.bsdfPdf, r.dTreePdf, r.weight, 0.0f};
                if (r.path & PPG_ADAM_DEFER_PATH_BIT) late.push_back(pr); else m_adamRecords.push_back(pr);
            }
            sink.clear();
        }
        // ... and the previous round's, now: the bit in their keys puts them behind this round's own records of the same D-tree
        m_adamRecords.insert(m_adamRecords.end(), m_deferredRecords.begin(), m_deferredRecords.end());
        m_deferredRecords.swap(late);
        // sharded rendering: the driver replaces the records by the union over all ranks — or, with one owner per D-tree
        // (ppgo_adam_records_by_owner), by the records of the D-trees this rank owns
        hookPhase = 0; ownerMode = false;
        if (passHook && passHook(passHookUser) != 0) { m_hookFailed = true; m_adamRecords.clear(); return; }
        std::sort(m_adamRecords.begin(), m_adamRecords.end(), [](const PackedAdamRecord &a, const PackedAdamRecord &b) { return a.key < b.key; });
        const Float ratioPower = m_bsdfSamplingFractionLoss == EKL ? 1.0f : 2.0f;
        for (const PackedAdamRecord &r : m_adamRecords)
            nodes[(size_t)(r.key >> PPG_ADAM_LEAF_SHIFT)].dTree.optimizeBsdfSamplingFraction(r.product, r.woPdf, r.bsdfPdf, r.dTreePdf, r.weight, ratioPower);
        m_adamRecords.clear();
        if (passHook && ownerMode) { hookPhase = 1; if (passHook(passHookUser) != 0) m_hookFailed = true; hookPhase = 0; }  // the owners publish the state they computed
    }
    bool m_hookFailed = false;
    int m_parity = -1;
    int m_region = -1, m_regions = 0;
    std::vector<int> m_spiralRank;  // [block] position of the block in the reference scheduler's spiral (ppg_spiral_block_ranks)
    void buildSpiral(int bx, int by) {
        m_spiralRank.assign((size_t)bx * by, -1);
        ppg_spiral_block_ranks(bx, by, m_spiralRank.data());
    }
    int hookPhase = 0;
    bool ownerMode = false;
    std::vector<uint32_t> m_adamState;  // [world * segment][6]

    void finishPasses(ppg_pass_stats *st) {  // GP:1288-1328
        Float variance = 0;
        const int N = m_passesLocal * m_sppPerPass;
        const int w = W(), h = H();
        if (m_sampleCombination == EInverseVariance) {
            std::vector<Float> img(m_image.size());
            for (int i = 0; i < w * h; ++i) {
                Float iw = m_imageW[i] != 0 ? 1.0f / m_imageW[i] : 0.0f;   // fmtconv.cpp:1036-1044
                for (int c = 0; c < 3; ++c) img[3 * i + c] = m_image[3 * i + c] * iw;
            }
            m_images.push_back(std::move(img));
        }
        for (int x = 0; x < w; ++x)
            for (int y = 0; y < h; ++y) {
                int i = y * w + x;
                Float iw = m_imageW[i] != 0 ? 1.0f / m_imageW[i] : 0.0f;
                Spectrum pixel(m_image[3 * i] * iw, m_image[3 * i + 1] * iw, m_image[3 * i + 2] * iw);
                Spectrum sq(m_squaredImage[3 * i] * iw, m_squaredImage[3 * i + 1] * iw, m_squaredImage[3 * i + 2] * iw);
                Spectrum localVar = sq - mul(pixel, pixel) / (Float)N;  // GP:1307
                for (int c = 0; c < 3; ++c) m_varianceBuffer[3 * i + c] = localVar[c];
                variance += ppg_min(luminance(localVar), 10000.0f);
            }
        variance /= (Float)w * h * (N - 1);  // GP:1313
        if (m_sampleCombination == EInverseVariance) m_variances.push_back(variance);
        if (st) {
            st->seconds = computeElapsedSeconds(m_passStart);
            st->passes_rendered_total = m_passesRendered;
            st->passes_rendered_local = m_passesLocal;
            st->variance = variance;
            st->samples = m_samplesLocal;
            st->rays = m_counters.rays; st->path_length_sum = m_counters.pathLen; st->vertices_committed = m_counters.committed;
        }
        m_lastVariance = variance;
    }
    Float m_lastVariance = 0;

    static Float computeElapsedSeconds(std::chrono::steady_clock::time_point start) {  // GP:1428-1432
        auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start);
        return (Float)ms.count() / 1000;
    }

    // one BlockedRenderProcess: every pixel × sppPerPass through Li (renderBlock GP:1587-1641)
    void renderOnePass() {
        const int w = W(), h = H();
        const int bs = 32;  // scene->getBlockSize()
        const int bx = (w + bs - 1) / bs, by = (h + bs - 1) / bs;
        const uint32_t passIndex = (uint32_t)m_passesRendered;
        const uint32_t sampleInRound0 = (uint32_t)(m_passesRendered - m_roundStartPass) * (uint32_t)m_sppPerPass;
        if (m_blockSinks.size() != (size_t)bx * by) m_blockSinks.assign((size_t)bx * by, std::vector<AdamRecord>());
        uint64_t rays = 0, plen = 0, comm = 0;
        if (m_regions > 0 && m_spiralRank.size() != (size_t)bx * by) buildSpiral(bx, by);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : rays, plen, comm)
#endif
        for (int b = 0; b < bx * by; ++b) {
            if (m_regions > 0 && (int)((long long)m_spiralRank[b] * m_regions / (bx * by)) != m_region) continue;
            PathCounters pc;
            Modes tm = modes;
            tm.sink = &m_blockSinks[b];
            uint64_t cnt0[CNT_N];
            memcpy(cnt0, t_cnt, sizeof cnt0);
            int x0 = (b % bx) * bs, y0 = (b / bx) * bs;
            for (int y = y0; y < std::min(y0 + bs, h); ++y)
                for (int x = x0; x < std::min(x0 + bs, w); ++x) {
                    if (!ownsPixel(x, y)) continue;
                    if (m_parity >= 0 && ((x + y) & 1) != m_parity) continue;
                    const uint32_t pixel = (uint32_t)(y * w + x);
                    for (int j = 0; j < m_sppPerPass; ++j) {
                        Sampler sampler{ppg_path_key(m_seed, pixel, passIndex * (uint32_t)m_sppPerPass + (uint32_t)j), 0};
                        Point2 s2 = sampler.next2D();
                        Point2 samplePos{(Float)x + s2.x, (Float)y + s2.y};  // GP:1620
                        Point o; Vec d; Float mint, maxt;
                        sampleRay(samplePos, o, d, mint, maxt);
                        tm.path = (sampleInRound0 + (uint32_t)j) * (uint32_t)(w * h) + pixel;
                        Spectrum spec = Li(o, d, mint, maxt, sampler, pc, tm);  // GP:1632 (sensor weight is 1)
                        // block->put / squaredBlock->put with the box filter: own pixel, weight 1 (SURVEY App. A "Film")
                        if (m_accImage) {  // a group of a final iteration: its own partial; image and film receive it whole (addGroup)
                            for (int c = 0; c < 3; ++c) { m_accImage[3 * pixel + c] += spec[c]; m_accSq[3 * pixel + c] += spec[c] * spec[c]; }
                            m_accW[pixel] += 1.0f;
                            continue;
                        }
                        for (int c = 0; c < 3; ++c) {
                            m_image[3 * pixel + c] += spec[c];
                            m_squaredImage[3 * pixel + c] += spec[c] * spec[c];
                            m_film[3 * pixel + c] += spec[c];  // BlockedRenderProcess::processResult → film->put
                        }
                        m_imageW[pixel] += 1.0f;
                        m_filmW[pixel] += 1.0f;
                    }
                }
            rays += pc.rays; plen += pc.pathLen; comm += pc.committed;
            for (int k = 0; k < CNT_N; ++k) __atomic_fetch_add(&m_work[k], t_cnt[k] - cnt0[k], __ATOMIC_RELAXED);
        }
        m_counters.rays += rays; m_counters.pathLen += plen; m_counters.committed += comm;
    }

    // PerspectiveCamera::sampleRayDifferential perspective.cpp:271-298
    void sampleRay(const Point2 &pixelSample, Point &o, Vec &d, Float &mint, Float &maxt) const {
        const ppg_camera &c = scene.cam;
        Float invResX = 1.0f / (Float)c.width, invResY = 1.0f / (Float)c.height;
        Point nearP = xfPoint(c.sample_to_camera, Point(pixelSample.x * invResX, pixelSample.y * invResY, 0.0f));
        Vec dl = normalize(Vec(nearP));
        Float invZ = 1.0f / dl.z;
        mint = c.near_clip * invZ;
        maxt = c.far_clip * invZ;
        o = Point(c.camera_to_world[3], c.camera_to_world[7], c.camera_to_world[11]);  // transformAffine(Point(0))
        d = xfVec(c.camera_to_world, dl);
    }

    // sampleMat / pdfMat GP:1650-1710 (diffuse BSDF: smooth, no delta component)
    Spectrum sampleMat(const Material &bsdf, BRec &bRec, const Frame &shFrame, Float &woPdf, Float &bsdfPdf, Float &dTreePdf,
                       Float bsdfSamplingFraction, Sampler &sampler, const DTreeWrapper *dTree) const {
        Point2 sample = sampler.next2D();
        if (!m_isBuilt || !dTree || BSDF::allDelta(bsdf)) {
            Spectrum result = BSDF::sample(bsdf, bRec, bsdfPdf, sample, &sampler);
            woPdf = bsdfPdf;
            dTreePdf = 0;
            return result;
        }
        Spectrum result;
        if (sample.x < bsdfSamplingFraction) {
            sample.x /= bsdfSamplingFraction;
            result = BSDF::sample(bsdf, bRec, bsdfPdf, sample, &sampler);
            if (isZero(result)) {
                woPdf = bsdfPdf = dTreePdf = 0;
                return Spectrum(0.0f);
            }
            if (bRec.sampledDelta) {  // GP:1672-1676 (unreachable with the supported BSDFs: a delta lobe implies all-delta here)
                dTreePdf = 0;
                woPdf = bsdfPdf * bsdfSamplingFraction;
                return result / bsdfSamplingFraction;
            }
            result = result * bsdfPdf;
        } else {
            sample.x = (sample.x - bsdfSamplingFraction) / (1 - bsdfSamplingFraction);
            bRec.wo = shFrame.toLocal(dTree->sample(&sampler));
            bRec.sampledDelta = false;
            bRec.eta = 1.0f;
            result = BSDF::eval(bsdf, bRec);
        }
        pdfMat(woPdf, bsdfPdf, dTreePdf, bsdfSamplingFraction, bsdf, bRec, shFrame, dTree);
        if (woPdf == 0) return Spectrum(0.0f);
        return result / woPdf;
    }

    void pdfMat(Float &woPdf, Float &bsdfPdf, Float &dTreePdf, Float bsdfSamplingFraction, const Material &bsdf, const BRec &bRec,
                const Frame &shFrame, const DTreeWrapper *dTree) const {
        dTreePdf = 0;
        if (!m_isBuilt || !dTree || BSDF::allDelta(bsdf)) {
            woPdf = bsdfPdf = BSDF::pdf(bsdf, bRec);
            return;
        }
        bsdfPdf = BSDF::pdf(bsdf, bRec);
        if (!ppg_isfinite(bsdfPdf)) {
            woPdf = 0;
            return;
        }
        dTreePdf = dTree->pdf(shFrame.toWorld(bRec.wo));
        woPdf = bsdfSamplingFraction * bsdfPdf + (1 - bsdfSamplingFraction) * dTreePdf;
    }

    static Float miWeight(Float pdfA, Float pdfB) {  // GP:2247-2250
        pdfA *= pdfA;
        pdfB *= pdfB;
        return pdfA / (pdfA + pdfB);
    }

    struct Vertex {  // GP:1713-1769
        DTreeWrapper *dTree;
        Vec dTreeVoxelSize;
        Point rayO;
        Vec rayD;
        Spectrum throughput, bsdfVal, radiance;
        Float woPdf, bsdfPdf, dTreePdf;
        bool isDelta;

        void record(const Spectrum &r) { radiance = radiance + r; }

        bool commit(STree &sdTree, Float statisticalWeight, ESpatialFilter spatialFilter, EDirectionalFilter directionalFilter,
                    ELoss loss, Sampler *sampler, Modes &modes, uint32_t code) {
            if (!(woPdf > 0) || !isValid(radiance) || !isValid(bsdfVal)) return false;
            Spectrum localRadiance(0.0f);
            if (throughput[0] * woPdf > PPG_EPSILON) localRadiance[0] = radiance[0] / throughput[0];
            if (throughput[1] * woPdf > PPG_EPSILON) localRadiance[1] = radiance[1] / throughput[1];
            if (throughput[2] * woPdf > PPG_EPSILON) localRadiance[2] = radiance[2] / throughput[2];
            Spectrum product = mul(localRadiance, bsdfVal);
            DTreeRecord rec{rayD, average(localRadiance), average(product), woPdf, bsdfPdf, dTreePdf, statisticalWeight, isDelta};
            modes.code = code;  // place of this vertex's records in the round's canonical order
            switch (spatialFilter) {
                case ESNearest:
                    dTree->record(rec, directionalFilter, loss, modes);
                    break;
                case EStochasticBox: {
                    Vec offset = dTreeVoxelSize;
                    offset.x *= sampler->next1D() - 0.5f;
                    offset.y *= sampler->next1D() - 0.5f;
                    offset.z *= sampler->next1D() - 0.5f;
                    Point origin = sdTree.aabb().clip(rayO + offset);
                    DTreeWrapper *splatDTree = sdTree.dTreeWrapper(origin);
                    if (splatDTree) splatDTree->record(rec, directionalFilter, loss, modes);
                    break;
                }
                case ESBox:
                    sdTree.record(rayO, dTreeVoxelSize, rec, directionalFilter, loss, modes);
                    break;
            }
            return true;
        }
    };

    // Li GP:1712-2157, surface branch
    Spectrum Li(Point o, Vec d, Float rayMint, Float rayMaxt, Sampler &sampler, PathCounters &pc, Modes &modes) {
        static const int MAX_NUM_VERTICES = 32;
        std::array<Vertex, MAX_NUM_VERTICES> vertices;
        Intersection its;
        Spectrum Li(0.0f);
        Float eta = 1.0f;
        int depth = 1;                 // rRec.newQuery: depth = 1
        bool emittedAllowed = true;    // rRec.type & EEmittedRadiance (ERadiance → ERadianceNoEmission)

        scene.rayIntersect(o, d, rayMint, rayMaxt, its);  // GP:1784
        pc.rays++;

        Spectrum throughput(1.0f);
        bool scattered = false;
        int nVertices = 0;

        auto recordRadiance = [&](Spectrum radiance) {  // GP:1791-1796
            Li = Li + radiance;
            for (int i = 0; i < nVertices; ++i) vertices[i].record(radiance);
        };

        while (depth <= m_maxDepth || m_maxDepth < 0) {
            if (!its.valid) {  // GP:1902-1914: radiance from a background luminaire
                if (emittedAllowed && (!m_hideEmitters || scattered) && scene.hasEnv) recordRadiance(mul(throughput, scene.envEval(d)));
                break;
            }
            if (its.emitter >= 0 && emittedAllowed && (!m_hideEmitters || scattered))
                recordRadiance(mul(throughput, scene.Le(its, -d)));  // GP:1917-1919

            if (depth >= m_maxDepth && m_maxDepth != -1) break;  // GP:1925

            Float wiDotGeoN = -dot(its.geoN, d), wiDotShN = its.wi.z;  // GP:1929-1932
            if (wiDotGeoN * wiDotShN < 0 && m_strictNormals) break;

            const Material bsdf = scene.materialAt(its);  // its.getBSDF() with the textures evaluated at its.uv
            Frame bumpFrame;
            const bool bumped = scene.bumpFrame(its, bumpFrame);
            Vec dTreeVoxelSize;
            DTreeWrapper *dTree = nullptr;
            if (BSDF::isSmooth(bsdf)) dTree = m_sdTree->dTreeWrapper(its.p, dTreeVoxelSize);  // GP:1942-1944

            Float bsdfSamplingFraction = m_bsdfSamplingFraction;  // GP:1946-1949
            if (dTree && m_bsdfSamplingFractionLoss != ENone) bsdfSamplingFraction = dTree->bsdfSamplingFraction();

            BRec bRec;
            bRec.wi = its.wi;
            if (bumped) { bRec.bumpSh = &its.shFrame; bRec.bumpPert = &bumpFrame; }
            Float woPdf, bsdfPdf, dTreePdf;
            Spectrum bsdfWeight = sampleMat(bsdf, bRec, its.shFrame, woPdf, bsdfPdf, dTreePdf, bsdfSamplingFraction, sampler, dTree);

            // Luminaire sampling, GP:1962-2021
            DRec dRec;  // DirectSamplingRecord dRec(its), records.inl:160-164
            dRec.ref = its.p;
            dRec.refN = BSDF::hasBackSideOrTransmission(bsdf) ? Vec(0.0f) : its.shFrame.n;
            if (m_doNee && BSDF::isSmooth(bsdf)) {
                int interactions = m_maxDepth - depth - 1;
                Spectrum value = scene.sampleEmitterDirect(dRec, sampler.next2D(), pc.rays, interactions, BSDF::hasNull, BSDF::evalNull);
                if (!isZero(value)) {
                    BRec bRecE;
                    bRecE.wi = its.wi;
                    bRecE.wo = its.shFrame.toLocal(dRec.d);
                    if (bumped) { bRecE.bumpSh = &its.shFrame; bRecE.bumpPert = &bumpFrame; }
                    Float woDotGeoNE = dot(its.geoN, dRec.d);
                    if (!m_strictNormals || woDotGeoNE * bRecE.wo.z > 0) {
                        const Spectrum bsdfVal = BSDF::eval(bsdf, bRecE);
                        Float woPdfE = 0, bsdfPdfE = 0, dTreePdfE = 0;
                        pdfMat(woPdfE, bsdfPdfE, dTreePdfE, bsdfSamplingFraction, bsdf, bRecE, its.shFrame, dTree);
                        const Float weight = miWeight(dRec.pdf, woPdfE);
                        value = mul(value, bsdfVal);
                        Spectrum L = mul(throughput, value) * weight;
                        if (!m_isFinalIter && m_nee != EAlways) {
                            if (dTree) {
                                Vertex v{dTree, dTreeVoxelSize, its.p, dRec.d, mul(throughput, bsdfVal) / dRec.pdf, bsdfVal, L,
                                         dRec.pdf, bsdfPdfE, dTreePdfE, false};
                                Sampler cs{sampler.key, PPG_DIM_NEE_COMMIT + 3u * (uint32_t)depth};  // sampler contract, ppg_rng.h
                                if (v.commit(*m_sdTree, 0.5f, m_spatialFilter, m_directionalFilter,
                                             m_isBuilt ? m_bsdfSamplingFractionLoss : ENone, &cs, modes, (uint32_t)std::min(depth, PPG_ADAM_CODE_VERTEX - 1)))
                                    pc.committed++;
                            }
                        }
                        recordRadiance(L);
                    }
                }
            }

            if (isZero(bsdfWeight)) break;  // GP:2024-2025

            const Vec wo = its.shFrame.toWorld(bRec.wo);  // GP:2028-2032
            Float woDotGeoN = dot(its.geoN, wo);
            if (woDotGeoN * bRec.wo.z <= 0 && m_strictNormals) break;

            o = its.p; d = wo;  // ray = Ray(its.p, wo, ray.time): mint = Epsilon, maxt = inf
            throughput = mul(throughput, bsdfWeight);
            eta *= bRec.eta;

            if (bRec.sampledNull) {  // index-matched pass-through, GP:2045-2075
                if (m_bsdfSamplingFractionLoss != ENone && dTree && nVertices < MAX_NUM_VERTICES && !m_isFinalIter) {
                    if (1 / woPdf > 0) {
                        vertices[nVertices] = Vertex{dTree, dTreeVoxelSize, o, d, throughput, bsdfWeight * woPdf, Spectrum(0.0f), woPdf, bsdfPdf, dTreePdf, true};
                        ++nVertices;
                    }
                }
                emittedAllowed = !scattered;  // rRec.type = scattered ? ERadianceNoEmission : ERadiance
                scene.rayIntersect(o, d, PPG_EPSILON, std::numeric_limits<Float>::infinity(), its);
                pc.rays++;
                depth++;
                continue;
            }

            // rayIntersectAndLookForEmitter GP:2184-2245 (no media): `its` stays the FIRST surface hit;
            // the search for an emitter continues through surfaces that have a null component
            Spectrum value(0.0f);
            {
                Intersection its2, *cur = &its;
                Spectrum transmittance(1.0f);
                Point ro = o;
                Float mint = PPG_EPSILON;
                const int maxInteractions = m_maxDepth - depth - 1;
                int interactions = 0;
                bool surface = false, abandoned = false;
                while (true) {
                    surface = scene.rayIntersect(ro, d, mint, std::numeric_limits<Float>::infinity(), *cur);
                    pc.rays++;
                    if (surface && (interactions == maxInteractions || !BSDF::hasNull(scene.materials[cur->material]) || cur->emitter >= 0)) break;
                    if (!surface) break;
                    if (isZero(transmittance)) { abandoned = true; break; }
                    Float cosThetaI = -cur->shFrame.toLocal(d).z;  // bRec(its, -wo, wo) in the shading frame
                    transmittance = mul(transmittance, BSDF::evalNull(scene.materials[cur->material], cosThetaI));
                    ro = ro + d * cur->t;
                    mint = PPG_EPSILON;
                    cur = &its2;
                    if (++interactions > 100) { abandoned = true; break; }
                }
                if (!abandoned && surface && cur->emitter >= 0) {
                    // dRec.setQuery(ray, *its), records.inl:170-178 (dist is measured from the LAST ray origin, as in the reference)
                    dRec.p = cur->p; dRec.n = cur->shFrame.n; dRec.d = d; dRec.dist = cur->t; dRec.emitter = cur->emitter;
                    value = mul(transmittance, scene.Le(*cur, -d));
                } else if (!abandoned && !surface && scene.hasEnv && scene.envFillDirectSamplingRecord(dRec, ro, d)) {  // GP:2236-2243
                    value = mul(transmittance, scene.envEval(d));
                    dRec.dist = std::numeric_limits<Float>::infinity();
                }
            }

            {  // GP:2083-2111
                bool isDelta = bRec.sampledDelta;
                const Float emitterPdf = (m_doNee && !isDelta && !isZero(value)) ? scene.pdfEmitterDirect(dRec) : 0;
                const Float weight = miWeight(woPdf, emitterPdf);
                Spectrum L = mul(throughput, value) * weight;
                if (!isZero(L)) recordRadiance(L);

                if ((!isDelta || m_bsdfSamplingFractionLoss != ENone) && dTree && nVertices < MAX_NUM_VERTICES && !m_isFinalIter) {
                    if (1 / woPdf > 0) {
                        vertices[nVertices] = Vertex{dTree, dTreeVoxelSize, o, d, throughput, bsdfWeight * woPdf,
                                                     (m_nee == EAlways) ? Spectrum(0.0f) : L, woPdf, bsdfPdf, dTreePdf, isDelta};
                        ++nVertices;
                    }
                }
            }

            emittedAllowed = false;  // rRec.type = ERadianceNoEmission, GP:2121

            if (depth++ >= m_rrDepth) {  // GP:2124-2142
                Float successProb = 1.0f;
                if (dTree && !bRec.sampledDelta) {
                    if (!m_isBuilt) successProb = specMax(throughput) * eta * eta;
                    successProb = ppg_max(0.1f, ppg_min(successProb, 0.99f));
                }
                if (sampler.next1D() >= successProb) break;
                throughput = throughput / successProb;
            }
            scattered = true;
        }
        pc.pathLen += (uint64_t)depth;  // avgPathLength += rRec.depth, GP:2147-2148
        __atomic_fetch_add(&m_lenHist[std::min(depth, (int)PPGO_LEN_HIST - 1)], (uint64_t)1, __ATOMIC_RELAXED);

        // include/ppg.h "STRAGGLERS": in a round whose stragglers' records are deferred, a path whose final rRec.depth exceeds
        // PPG_ADAM_DEFER_DEPTH leaves its optimiser records marked — applyAdamRound() applies them with the NEXT round
        if (m_deferRound && depth > m_deferDepth) modes.path |= PPG_ADAM_DEFER_PATH_BIT;
        if (nVertices > 0 && !m_isFinalIter) {  // GP:2150-2154
            const uint32_t dimEnd = sampler.dim;
            for (int i = 0; i < nVertices; ++i) {
                sampler.dim = dimEnd + 3u * (uint32_t)i;  // sampler contract: the commit draws of vertex i (ppg_rng.h)
                bool ok = vertices[i].commit(*m_sdTree, m_nee == EKickstart && m_doNee ? 0.5f : 1.0f, m_spatialFilter,
                                             m_directionalFilter, m_isBuilt ? m_bsdfSamplingFractionLoss : ENone, &sampler, modes, (uint32_t)(PPG_ADAM_CODE_VERTEX + i));
                if (ok) pc.committed++;
            }
        }
        return Li;
    }

    // ---- render() and its drivers ----
    // false: a cancel that arrived before the render began is consumed and cancels it (the product's sticky ppg_cancel)
    bool beginRender() {  // GP:1519-1550
        m_sdTree.reset(new STree(scene.aabb));
        m_iter = 0;
        m_isFinalIter = false;
        m_isBuilt = false;
        size_t n = (size_t)W() * H();
        m_image.assign(3 * n, 0); m_squaredImage.assign(3 * n, 0); m_imageW.assign(n, 0);
        m_film.assign(3 * n, 0); m_filmW.assign(n, 0); m_varianceBuffer.assign(3 * n, 0);
        m_images.clear(); m_variances.clear();
        m_startTime = std::chrono::steady_clock::now();
        m_passesRendered = 0; m_passesRenderedThisIter = 0;
        {
            const bool pending = cancelled, spent = m_cancelSeen;
            cancelled = false; m_cancelSeen = false;
            if (pending && !spent) return false;
        }
        return true;
    }
    void beginIteration(bool isFinal) {  // GP:1378-1381
        m_isFinalIter = isFinal;
        std::fill(m_film.begin(), m_film.end(), 0.0f);
        std::fill(m_filmW.begin(), m_filmW.end(), 0.0f);
        resetSDTree();
    }
    void endIteration() {  // GP:1417-1422
        if (m_dumpSDTree && !m_isFinalIter && !m_dumpPrefix.empty()) {
            char buf[1024];
            snprintf(buf, sizeof buf, "%s-%02d.sdt", m_dumpPrefix.c_str(), m_iter);
            dumpSDTree(buf);
        }
        ++m_iter;
        m_passesRenderedThisIter = 0;
    }
    void endRender() {  // GP:1567-1582
        if (m_sampleCombination == EInverseVariance && !m_images.empty()) {
            std::fill(m_film.begin(), m_film.end(), 0.0f);
            std::fill(m_filmW.begin(), m_filmW.end(), 1.0f);
            size_t begin = m_images.size() - std::min(m_images.size(), (size_t)4);
            Float totalWeight = 0;
            for (size_t i = begin; i < m_variances.size(); ++i) totalWeight += 1.0f / m_variances[i];
            for (size_t i = begin; i < m_images.size(); ++i) {
                Float mult = 1.0f / m_variances[i] / totalWeight;
                for (size_t k = 0; k < m_film.size(); ++k) m_film[k] += m_images[i][k] * mult;
            }
        }
    }

    bool doNeeWithSpp(int spp) const {  // GP:1331-1340
        switch (m_nee) {
            case ENever: return false;
            case EKickstart: return spp < 128;
            default: return true;
        }
    }

    bool renderSPP() {  // GP:1342-1426
        size_t sampleCount = (size_t)m_budget;
        int nPasses = (int)std::ceil(sampleCount / (Float)m_sppPerPass);
        bool result = true;
        Float currentVarAtEnd = std::numeric_limits<Float>::infinity();
        while (result && m_passesRendered < nPasses) {
            const int sppRendered = m_passesRendered * m_sppPerPass;
            m_doNee = doNeeWithSpp(sppRendered);
            int remainingPasses = nPasses - m_passesRendered;
            int passesThisIteration = std::min(remainingPasses, 1 << m_iter);
            if (remainingPasses - passesThisIteration < 2 * passesThisIteration) passesThisIteration = remainingPasses;
            beginIteration(passesThisIteration >= remainingPasses);
            ppg_pass_stats st;
            renderPassesNoStat(passesThisIteration); finishPasses(&st);
            if (seesCancel()) { result = false; break; }
            Float variance = st.variance;
            const Float lastVarAtEnd = currentVarAtEnd;
            currentVarAtEnd = passesThisIteration * variance / remainingPasses;
            remainingPasses -= passesThisIteration;
            if (m_sampleCombination == EDiscardWithAutomaticBudget && remainingPasses > 0 &&
                (remainingPasses < passesThisIteration || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
                m_isFinalIter = true;
                renderPassesNoStat(remainingPasses); finishPasses(&st);
                if (seesCancel()) { result = false; break; }
            }
            buildSDTree(nullptr);
            endIteration();
        }
        return result;
    }

    bool renderTime() {  // GP:1434-1514
        Float nSeconds = m_budget;
        bool result = true;
        Float currentVarAtEnd = std::numeric_limits<Float>::infinity();
        Float elapsedSeconds = 0;
        while (result && elapsedSeconds < nSeconds) {
            const int sppRendered = m_passesRendered * m_sppPerPass;
            m_doNee = doNeeWithSpp(sppRendered);
            Float remainingTime = nSeconds - elapsedSeconds;
            const int passesThisIteration = 1 << m_iter;
            const auto startIter = std::chrono::steady_clock::now();
            beginIteration(false);
            ppg_pass_stats st;
            renderPassesNoStat(passesThisIteration); finishPasses(&st);
            if (seesCancel()) { result = false; break; }
            Float variance = st.variance;
            const Float secondsIter = computeElapsedSeconds(startIter);
            const Float lastVarAtEnd = currentVarAtEnd;
            currentVarAtEnd = secondsIter * variance / remainingTime;
            remainingTime -= secondsIter;
            if (m_sampleCombination == EDiscardWithAutomaticBudget && remainingTime > 0 &&
                (remainingTime < secondsIter || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
                m_isFinalIter = true;
                do {
                    renderPassesNoStat(passesThisIteration); finishPasses(&st);
                    if (seesCancel()) { result = false; break; }
                    elapsedSeconds = computeElapsedSeconds(m_startTime);
                } while (elapsedSeconds < nSeconds);
            }
            buildSDTree(nullptr);
            endIteration();
            elapsedSeconds = computeElapsedSeconds(m_startTime);
        }
        return result;
    }

    bool render() {  // GP:1516-1585
        if (!beginRender()) return false;
        bool result = m_budgetType == ESpp ? renderSPP() : renderTime();
        endRender();
        return result;
    }
};

int parseEnum(const char *s, const char *dflt, std::initializer_list<const char *> names) {
    std::string v = s ? s : dflt;
    int i = 0;
    for (const char *n : names) {
        if (v == n) return i;
        ++i;
    }
    return -1;
}

thread_local std::string g_createError;

}  // namespace

// ================================================================================================
// C interface
// ================================================================================================
struct ppgo_ctx {
    GuidedPathTracer gpt;
};

extern "C" {

int ppgo_create(const ppg_config *cfg, ppgo_ctx **out) {
    if (!cfg || !out) { g_createError = "null argument"; return PPG_ERR_INVALID; }
    std::unique_ptr<ppgo_ctx> c(new ppgo_ctx());
    GuidedPathTracer &g = c->gpt;
    int v;
#define PARSE(field, dflt, target, type, ...)                                                   \
    v = parseEnum(cfg->field, dflt, {__VA_ARGS__});                                             \
    if (v < 0) { g_createError = std::string("invalid value for '" #field "': ") + cfg->field; return PPG_ERR_INVALID; } \
    target = (type)v;
    PARSE(nee, "never", g.m_nee, ENee, "never", "kickstart", "always")
    PARSE(sampleCombination, "automatic", g.m_sampleCombination, ESampleCombination, "discard", "automatic", "inversevar")
    PARSE(spatialFilter, "nearest", g.m_spatialFilter, ESpatialFilter, "nearest", "stochastic", "box")
    PARSE(directionalFilter, "nearest", g.m_directionalFilter, EDirectionalFilter, "nearest", "box")
    PARSE(bsdfSamplingFractionLoss, "none", g.m_bsdfSamplingFractionLoss, ELoss, "none", "kl", "var")
    PARSE(budgetType, "seconds", g.m_budgetType, EBudget, "spp", "seconds")
#undef PARSE
    g.m_sdTreeMaxMemory = cfg->sdTreeMaxMemory; g.m_sTreeThreshold = cfg->sTreeThreshold;
    g.m_dTreeThreshold = cfg->dTreeThreshold; g.m_bsdfSamplingFraction = cfg->bsdfSamplingFraction;
    g.m_sppPerPass = cfg->sppPerPass; g.m_budget = cfg->budget; g.m_dumpSDTree = cfg->dumpSDTree != 0;
    g.m_rrDepth = cfg->rrDepth; g.m_maxDepth = cfg->maxDepth; g.m_strictNormals = cfg->strictNormals != 0;
    g.m_hideEmitters = cfg->hideEmitters != 0; g.m_seed = cfg->seed;
    if (cfg->dumpPrefix) g.m_dumpPrefix = cfg->dumpPrefix;
    if (g.m_sppPerPass <= 0) { g_createError = "sppPerPass must be > 0"; return PPG_ERR_INVALID; }
    *out = c.release();
    return PPG_OK;
}

void ppgo_destroy(ppgo_ctx *ctx) { delete ctx; }
const char *ppgo_last_error(const ppgo_ctx *ctx) { return ctx ? ctx->gpt.error.c_str() : g_createError.c_str(); }

int ppgo_set_modes(ppgo_ctx *ctx, int32_t acc_mode, int32_t adam_mode, int32_t threads) {
    ctx->gpt.modes.acc = acc_mode;
    ctx->gpt.modes.adam = adam_mode;
    ctx->gpt.threads = (acc_mode == PPGO_ACC_FLOAT || adam_mode == PPGO_ADAM_SEQUENTIAL) ? 1 : std::max(1, threads);
    return PPG_OK;
}

int ppgo_set_scene(ppgo_ctx *ctx, const ppg_scene *s) {
    Scene &sc = ctx->gpt.scene;
    sc = Scene();
    if (!s || !s->materials || (s->n_triangles == 0 && s->n_spheres == 0) ||
        (s->n_triangles > 0 && (!s->positions || !s->indices || !s->tri_material || !s->tri_emitter))) {
        ctx->gpt.error = "incomplete scene";
        return PPG_ERR_INVALID;
    }
    for (uint32_t i = 0; i < s->n_vertices; ++i) sc.P.push_back(Point(s->positions[3 * i], s->positions[3 * i + 1], s->positions[3 * i + 2]));
    sc.hasNormals = s->normals != nullptr;
    if (sc.hasNormals)
        for (uint32_t i = 0; i < s->n_vertices; ++i) sc.N.push_back(Vec(s->normals[3 * i], s->normals[3 * i + 1], s->normals[3 * i + 2]));
    if (s->n_triangles) {
        sc.idx.assign(s->indices, s->indices + 3 * (size_t)s->n_triangles);
        sc.triMat.assign(s->tri_material, s->tri_material + s->n_triangles);
        sc.triEmitter.assign(s->tri_emitter, s->tri_emitter + s->n_triangles);
    }
    sc.hasEnv = s->environment != nullptr;
    if (sc.hasEnv) sc.envRadiance = Spectrum(s->environment[0], s->environment[1], s->environment[2]);
    if (s->envmap) {
        const ppg_envmap &em = *s->envmap;
        if (sc.hasEnv) { ctx->gpt.error = "envmap: a scene has one environment emitter (`environment` is set as well)"; return PPG_ERR_INVALID; }
        if (!em.rgb || em.width == 0 || em.height == 0 || em.width > 0xFFFF || em.height > 0xFFFF) { ctx->gpt.error = "envmap: needs pixels and 0 < width, height < 65536"; return PPG_ERR_INVALID; }
        EnvMap &m = sc.envMap;
        m.w = (int)em.width; m.h = (int)em.height; m.scale = em.scale;
        memcpy(m.R, em.to_world, sizeof m.R);
        m.texel.resize((size_t)m.w * m.h);
        for (size_t k = 0; k < m.texel.size(); ++k) m.texel[k] = Spectrum(em.rgb[3 * k], em.rgb[3 * k + 1], em.rgb[3 * k + 2]);
        m.configure();
        if (!m.valid) { ctx->gpt.error = "envmap: the environment map is completely black or holds nan / inf (envmap.cpp:308-312)"; return PPG_ERR_INVALID; }
        sc.hasEnv = true;
    }
    sc.materials.clear();
    if (s->n_rtrans) {
        if (!s->rtrans || s->rtrans_samples < 2) { ctx->gpt.error = "rtrans: need the slices and rtrans_samples >= 2"; return PPG_ERR_INVALID; }
        sc.rtrans.assign(s->rtrans, s->rtrans + (size_t)s->n_rtrans * (s->rtrans_samples + 1));
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        Material m;
        static_cast<ppg_material &>(m) = s->materials[i];
        if (m.type < 0 || m.type > PPG_BSDF_LAST) { ctx->gpt.error = "unsupported BSDF type"; return PPG_ERR_INVALID; }
        if (m.type == PPG_BSDF_ROUGHPLASTIC) {
            if (m.rtrans < 0 || (uint32_t)m.rtrans >= s->n_rtrans) { ctx->gpt.error = "roughplastic: material.rtrans is not a slice of scene.rtrans"; return PPG_ERR_INVALID; }
            if (!(m.eta[0] > 0)) { ctx->gpt.error = "plastic / dielectric need eta[0] = intIOR / extIOR > 0"; return PPG_ERR_INVALID; }
            m.rt = sc.rtrans.data() + (size_t)m.rtrans * (s->rtrans_samples + 1);
            m.rtN = s->rtrans_samples;
        }
        if ((m.type == PPG_BSDF_PLASTIC || m.type == PPG_BSDF_DIELECTRIC || m.type == PPG_BSDF_THINDIELECTRIC || m.type == PPG_BSDF_ROUGHDIELECTRIC) && !(m.eta[0] > 0)) { ctx->gpt.error = "plastic / dielectric need eta[0] = intIOR / extIOR > 0"; return PPG_ERR_INVALID; }
        m.configure();
        sc.materials.push_back(m);
    }
    if (s->n_textures) {
        if (!s->textures) { ctx->gpt.error = "textures: n_textures > 0 but no array"; return PPG_ERR_INVALID; }
        for (uint32_t i = 0; i < s->n_textures; ++i) {
            const ppg_texture &t = s->textures[i];
            if (!t.rgb || t.width == 0 || t.height == 0 || t.width > 0x7fff || t.height > 0x7fff || t.wrap_u < 0 || t.wrap_u > PPG_WRAP_ONE || t.wrap_v < 0 || t.wrap_v > PPG_WRAP_ONE) {
                ctx->gpt.error = "texture: needs pixels, 0 < width, height < 32768 and valid wrap modes"; return PPG_ERR_INVALID;
            }
            Texture x;
            x.w = (int)t.width; x.h = (int)t.height; x.su = t.uv_scale[0]; x.sv = t.uv_scale[1]; x.ou = t.uv_offset[0]; x.ov = t.uv_offset[1];
            x.wrapU = t.wrap_u; x.wrapV = t.wrap_v; x.nearest = t.nearest != 0;
            x.texel.resize((size_t)x.w * x.h);
            for (size_t k = 0; k < x.texel.size(); ++k) x.texel[k] = Spectrum(t.rgb[3 * k], t.rgb[3 * k + 1], t.rgb[3 * k + 2]);
            sc.textures.push_back(std::move(x));
        }
    }
    for (const Material &m : sc.materials) {
        const uint32_t a = m.texture & 0xffffu, b = m.texture >> 16;
        if (a > s->n_textures || b > s->n_textures) { ctx->gpt.error = "material.texture: index out of range"; return PPG_ERR_INVALID; }
        if (a && m.type != PPG_BSDF_DIFFUSE && m.type != PPG_BSDF_PLASTIC && m.type != PPG_BSDF_ROUGHPLASTIC) { ctx->gpt.error = "material.texture: only the diffuse reflectance of diffuse / plastic / roughplastic can carry a bitmap"; return PPG_ERR_INVALID; }
    }
    if (s->texcoords)
        for (uint32_t i = 0; i < s->n_vertices; ++i) sc.UV.push_back(Point2{s->texcoords[2 * i], s->texcoords[2 * i + 1]});
    if (s->n_emitters) sc.emitters.assign(s->emitters, s->emitters + s->n_emitters);
    if (s->n_spheres) {
        if (!s->spheres) { ctx->gpt.error = "spheres: n_spheres > 0 but no array"; return PPG_ERR_INVALID; }
        sc.spheres.assign(s->spheres, s->spheres + s->n_spheres);
        std::vector<int> users(s->n_emitters, 0);
        for (uint32_t t = 0; t < s->n_triangles; ++t) if (s->tri_emitter[t] >= 0 && s->tri_emitter[t] < (int32_t)s->n_emitters) users[s->tri_emitter[t]] = 1;
        for (const ppg_sphere &sp : sc.spheres) {
            if (!(sp.radius > 0)) { ctx->gpt.error = "sphere: radius must be > 0"; return PPG_ERR_INVALID; }
            if (sp.material >= s->n_materials || sp.emitter >= (int32_t)s->n_emitters) { ctx->gpt.error = "index out of range"; return PPG_ERR_INVALID; }
            if (sp.emitter >= 0 && users[sp.emitter]++) { ctx->gpt.error = "sphere: its emitter is shared with another shape"; return PPG_ERR_INVALID; }
            if (sc.materials[sp.material].texture) { ctx->gpt.error = "sphere: textured BSDFs are only supported on triangle meshes"; return PPG_ERR_INVALID; }
        }
    }
    for (uint32_t t = 0; t < s->n_triangles; ++t) {
        if (sc.triMat[t] >= s->n_materials || sc.triEmitter[t] >= (int32_t)s->n_emitters) { ctx->gpt.error = "index out of range"; return PPG_ERR_INVALID; }
        for (int k = 0; k < 3; ++k) if (sc.idx[3 * t + k] >= s->n_vertices) { ctx->gpt.error = "vertex index out of range"; return PPG_ERR_INVALID; }
    }
    sc.cam = s->camera;
    sc.finalize();
    ctx->gpt.haveScene = true;
    return PPG_OK;
}

int ppgo_set_shard(ppgo_ctx *ctx, int32_t rank, int32_t world, int32_t tile_size) {
    if (world < 1 || rank < 0 || rank >= world || tile_size < 1) return PPG_ERR_INVALID;
    ctx->gpt.shardRank = rank; ctx->gpt.shardWorld = world; ctx->gpt.tileSize = tile_size;
    return PPG_OK;
}

#define NEED_SCENE if (!ctx->gpt.haveScene) { ctx->gpt.error = "no scene"; return PPG_ERR_STATE; }
#define NEED_TREE if (!ctx->gpt.m_sdTree) { ctx->gpt.error = "render not begun"; return PPG_ERR_STATE; }

int ppgo_render(ppgo_ctx *ctx) { NEED_SCENE return ctx->gpt.render() ? PPG_OK : PPG_ERR_CANCELLED; }
int ppgo_begin_render(ppgo_ctx *ctx) { NEED_SCENE return ctx->gpt.beginRender() ? PPG_OK : PPG_ERR_CANCELLED; }
int ppgo_begin_iteration(ppgo_ctx *ctx, int32_t is_final) { NEED_TREE ctx->gpt.beginIteration(is_final != 0); return PPG_OK; }
int ppgo_set_final(ppgo_ctx *ctx, int32_t is_final) { ctx->gpt.m_isFinalIter = is_final != 0; return PPG_OK; }
int ppgo_set_do_nee(ppgo_ctx *ctx, int32_t do_nee) { ctx->gpt.m_doNee = do_nee != 0; return PPG_OK; }
int ppgo_render_passes_nostat(ppgo_ctx *ctx, int32_t n) {
    NEED_TREE
    ctx->gpt.renderPassesNoStat(n);
    if (ctx->gpt.m_hookFailed) { ctx->gpt.error = "round hook failed"; return PPG_ERR_INVALID; }
    return ctx->gpt.seesCancel() ? PPG_ERR_CANCELLED : PPG_OK;
}
int ppgo_finish_passes(ppgo_ctx *ctx, ppg_pass_stats *st) { NEED_TREE ctx->gpt.finishPasses(st); return PPG_OK; }
int ppgo_render_passes(ppgo_ctx *ctx, int32_t n, ppg_pass_stats *st) {
    NEED_TREE
    ctx->gpt.renderPassesNoStat(n);
    ctx->gpt.finishPasses(st);
    return ctx->gpt.seesCancel() ? PPG_ERR_CANCELLED : PPG_OK;
}
int ppgo_build_sdtree(ppgo_ctx *ctx, ppg_tree_stats *st) { NEED_TREE ctx->gpt.buildSDTree(st); return PPG_OK; }
int ppgo_end_iteration(ppgo_ctx *ctx) { NEED_TREE ctx->gpt.endIteration(); return PPG_OK; }
int ppgo_end_render(ppgo_ctx *ctx) { NEED_TREE ctx->gpt.endRender(); return PPG_OK; }
int ppgo_cancel(ppgo_ctx *ctx) { ctx->gpt.cancelled = true; return PPG_OK; }

int ppgo_read_film(ppgo_ctx *ctx, float *rgb) {
    NEED_SCENE
    GuidedPathTracer &g = ctx->gpt;
    size_t n = (size_t)g.W() * g.H();
    if (g.m_film.size() != 3 * n) return PPG_ERR_STATE;
    for (size_t i = 0; i < n; ++i) {
        Float iw = g.m_filmW[i] != 0 ? 1.0f / g.m_filmW[i] : 0.0f;
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = g.m_film[3 * i + c] * iw;
    }
    return PPG_OK;
}
int ppgo_read_variance(ppgo_ctx *ctx, float *rgb) {
    NEED_SCENE
    memcpy(rgb, ctx->gpt.m_varianceBuffer.data(), ctx->gpt.m_varianceBuffer.size() * sizeof(float));
    return PPG_OK;
}
int ppgo_dump_sdtree(ppgo_ctx *ctx, const char *path) { NEED_TREE ctx->gpt.dumpSDTree(path); return PPG_OK; }

int ppgo_sdtree_info_get(ppgo_ctx *ctx, ppg_sdtree_info *info) {
    NEED_TREE
    GuidedPathTracer &g = ctx->gpt;
    auto &nodes = g.m_sdTree->nodes();
    memset(info, 0, sizeof *info);
    info->n_stree_nodes = (uint32_t)nodes.size();
    for (auto &n : nodes)
        if (n.isLeaf) {
            info->n_leaves++;
            info->n_sampling_nodes += n.dTree.sampling.numNodes();
            info->n_building_nodes += n.dTree.building.numNodes();
        }
    for (int a = 0; a < 3; ++a) { info->aabb_min[a] = g.m_sdTree->aabb().min[a]; info->aabb_max[a] = g.m_sdTree->aabb().max[a]; }
    info->iter = g.m_iter;
    info->is_built = g.m_isBuilt;
    return PPG_OK;
}

int ppgo_sdtree_read_stree(ppgo_ctx *ctx, int32_t *axis, uint32_t *children) {
    NEED_TREE
    auto &nodes = ctx->gpt.m_sdTree->nodes();
    for (size_t i = 0; i < nodes.size(); ++i) {
        axis[i] = nodes[i].axis;
        children[2 * i] = nodes[i].isLeaf ? 0 : nodes[i].children[0];
        children[2 * i + 1] = nodes[i].isLeaf ? 0 : nodes[i].children[1];
    }
    return PPG_OK;
}

int ppgo_sdtree_read_dtree_headers(ppgo_ctx *ctx, int32_t which, uint64_t *offset, uint32_t *num_nodes, int32_t *max_depth,
                                   float *sum, double *stat_weight) {
    NEED_TREE
    auto &nodes = ctx->gpt.m_sdTree->nodes();
    uint64_t off = 0;
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (!nodes[i].isLeaf) { offset[i] = 0; num_nodes[i] = 0; max_depth[i] = 0; sum[i] = 0; stat_weight[i] = 0; continue; }
        const DTree &t = which == 0 ? nodes[i].dTree.sampling : nodes[i].dTree.building;
        offset[i] = off; num_nodes[i] = (uint32_t)t.numNodes(); max_depth[i] = t.depth(); sum[i] = t.sumValue();
        if (which == 1 && ctx->gpt.modes.acc == PPGO_ACC_FIXED) stat_weight[i] = (double)t.statAcc() / 16777216.0;
        else stat_weight[i] = t.statisticalWeight();
        off += t.numNodes();
    }
    return PPG_OK;
}

int ppgo_sdtree_read_dtree_nodes(ppgo_ctx *ctx, int32_t which, float *sums, uint16_t *children, uint64_t *fixed_sums) {
    NEED_TREE
    auto &nodes = ctx->gpt.m_sdTree->nodes();
    size_t k = 0;
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (!nodes[i].isLeaf) continue;
        const DTree &t = which == 0 ? nodes[i].dTree.sampling : nodes[i].dTree.building;
        for (size_t n = 0; n < t.numNodes(); ++n, ++k)
            for (int j = 0; j < 4; ++j) {
                const QuadTreeNode &q = t.node(n);
                float s = q.sum(j);
                if (which == 1 && ctx->gpt.modes.acc == PPGO_ACC_FIXED && q.isLeaf(j)) s = ppg_from_fixed(q.acc(j));
                sums[4 * k + j] = s;
                children[4 * k + j] = q.child(j);
                if (fixed_sums) fixed_sums[4 * k + j] = q.acc(j);
            }
    }
    return PPG_OK;
}

int ppgo_sdtree_read_adam(ppgo_ctx *ctx, float *theta) {
    NEED_TREE
    auto &nodes = ctx->gpt.m_sdTree->nodes();
    for (size_t i = 0; i < nodes.size(); ++i) theta[i] = nodes[i].dTree.bsdfSamplingFractionOptimizer.variable();
    return PPG_OK;
}

int ppgo_stat_sizes(ppgo_ctx *ctx, uint64_t *n_sums, uint64_t *n_weights) {
    NEED_TREE
    uint64_t s = 0, w = 0;
    for (auto &n : ctx->gpt.m_sdTree->nodes())
        if (n.isLeaf) { s += 4 * n.dTree.building.numNodes(); w += 1; }
    *n_sums = s; *n_weights = w;
    return PPG_OK;
}
int ppgo_stat_export(ppgo_ctx *ctx, uint64_t *sums, uint64_t n_sums, uint64_t *weights, uint64_t n_weights) {
    NEED_TREE
    std::vector<uint64_t> s, w;
    for (auto &n : ctx->gpt.m_sdTree->nodes())
        if (n.isLeaf) n.dTree.building.exportAcc(s, w);
    if (s.size() != n_sums || w.size() != n_weights) return PPG_ERR_INVALID;
    memcpy(sums, s.data(), 8 * s.size());
    memcpy(weights, w.data(), 8 * w.size());
    return PPG_OK;
}
int ppgo_stat_import(ppgo_ctx *ctx, const uint64_t *sums, uint64_t n_sums, const uint64_t *weights, uint64_t n_weights) {
    NEED_TREE
    uint64_t s, w;
    ppgo_stat_sizes(ctx, &s, &w);
    if (s != n_sums || w != n_weights) return PPG_ERR_INVALID;
    for (auto &n : ctx->gpt.m_sdTree->nodes())
        if (n.isLeaf) n.dTree.building.importAcc(sums, weights);
    return PPG_OK;
}
int ppgo_work_counters(ppgo_ctx *ctx, uint64_t *out8) { memcpy(out8, ctx->gpt.m_work, sizeof ctx->gpt.m_work); return PPG_OK; }
int ppgo_set_adam_regions(ppgo_ctx *ctx, int32_t regions) {  // include/ppg.h "Rounds by image region"
    if (regions < 0 || regions > 4096) { ctx->gpt.error = "ppg_set_adam_regions: 0 .. 4096"; return PPG_ERR_INVALID; }
    if (ctx->gpt.modes.adam == PPGO_ADAM_SEQUENTIAL) return PPG_OK;  // (the literal rule has no rounds)
    ctx->gpt.modes.adam = regions >= 2 ? PPGO_ADAM_REGIONS + regions : PPGO_ADAM_ROUND;
    return PPG_OK;
}
int ppgo_debug_set_defer_depth(ppgo_ctx *ctx, int32_t depth) { if (depth < 1 || depth > 64) return PPG_ERR_INVALID; ctx->gpt.m_deferDepth = depth; return PPG_OK; }
int ppgo_path_length_histogram(ppgo_ctx *ctx, uint64_t *out) { memcpy(out, ctx->gpt.m_lenHist, sizeof ctx->gpt.m_lenHist); return PPG_OK; }
int ppgo_set_stop_hook(ppgo_ctx *ctx, ppg_stop_hook hook, void *user) { ctx->gpt.stopHook = hook; ctx->gpt.stopHookUser = user; return PPG_OK; }
int ppgo_set_pass_hook(ppgo_ctx *ctx, ppg_pass_hook hook, void *user) { ctx->gpt.passHook = hook; ctx->gpt.passHookUser = user; return PPG_OK; }
int ppgo_adam_records(ppgo_ctx *ctx, void **records, uint64_t *n) {
    *records = ctx->gpt.m_adamRecords.data(); *n = ctx->gpt.m_adamRecords.size();
    return PPG_OK;
}
int ppgo_adam_records_replace(ppgo_ctx *ctx, const void *records, uint64_t n) {
    const auto *r = (const GuidedPathTracer::PackedAdamRecord *)records;
    std::vector<GuidedPathTracer::PackedAdamRecord> v(r, r + n);  // `records` may alias the current array
    ctx->gpt.m_adamRecords.swap(v);
    return PPG_OK;
}
int ppgo_hook_phase(ppgo_ctx *ctx, int32_t *phase) { *phase = ctx->gpt.hookPhase; return PPG_OK; }
int ppgo_adam_records_by_owner(ppgo_ctx *ctx, int32_t world, void **records, uint64_t *counts) {
    NEED_TREE
    auto &g = ctx->gpt;
    if (world < 1 || g.hookPhase != 0) { g.error = "ppg_adam_records_by_owner: only valid in phase 0 of the round hook"; return PPG_ERR_STATE; }
    std::sort(g.m_adamRecords.begin(), g.m_adamRecords.end(), [](const GuidedPathTracer::PackedAdamRecord &a, const GuidedPathTracer::PackedAdamRecord &b) { return a.key < b.key; });
    const uint64_t nNodes = g.m_sdTree->nodes().size(), seg = (nNodes + world - 1) / world;
    for (int32_t r = 0; r < world; ++r) counts[r] = 0;
    for (const auto &rec : g.m_adamRecords) counts[(rec.key >> PPG_ADAM_LEAF_SHIFT) / seg]++;
    *records = g.m_adamRecords.data();
    g.ownerMode = true;
    return PPG_OK;
}
int ppgo_adam_state(ppgo_ctx *ctx, int32_t world, void **state, uint64_t *segment) {
    NEED_TREE
    auto &g = ctx->gpt;
    if (world < 1 || g.hookPhase != 1) { g.error = "ppg_adam_state: only valid in phase 1 of the round hook"; return PPG_ERR_STATE; }
    auto &nodes = g.m_sdTree->nodes();
    const uint64_t seg = (nodes.size() + world - 1) / world;
    g.m_adamState.assign((size_t)world * seg * 6, 0u);
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i].dTree.bsdfSamplingFractionOptimizer.exportState(&g.m_adamState[6 * i]);
    *state = g.m_adamState.data(); *segment = seg;
    return PPG_OK;
}
int ppgo_adam_state_commit(ppgo_ctx *ctx) {
    NEED_TREE
    auto &g = ctx->gpt;
    auto &nodes = g.m_sdTree->nodes();
    if (g.hookPhase != 1 || g.m_adamState.size() < 6 * nodes.size()) { g.error = "ppg_adam_state_commit: call ppg_adam_state first"; return PPG_ERR_STATE; }
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i].dTree.bsdfSamplingFractionOptimizer.importState(&g.m_adamState[6 * i]);
    return PPG_OK;
}
int32_t ppgo_final_group_passes(int32_t n_passes) { return GuidedPathTracer::finalGroupPasses(n_passes); }
int ppgo_final_partials(ppgo_ctx *ctx, void **data, uint64_t *n_floats) {
    NEED_TREE
    auto &g = ctx->gpt;
    *data = nullptr; *n_floats = 0;
    if (!g.m_partialsPending) return PPG_OK;
    const size_t n = (size_t)g.W() * g.H();
    if (!g.m_partialsExported) {
        std::copy(g.m_film.begin(), g.m_film.end(), g.m_partials.begin());
        std::copy(g.m_filmW.begin(), g.m_filmW.end(), g.m_partials.begin() + 3 * n);
        g.m_partialsExported = true;
    }
    *data = g.m_partials.data(); *n_floats = g.m_partials.size();
    return PPG_OK;
}
int ppgo_final_partials_commit(ppgo_ctx *ctx) {
    NEED_TREE
    auto &g = ctx->gpt;
    if (!g.m_partialsPending || !g.m_partialsExported) { g.error = "ppg_final_partials_commit: nothing to commit"; return PPG_ERR_STATE; }
    const size_t n = (size_t)g.W() * g.H();
    std::copy(g.m_partials.begin(), g.m_partials.begin() + 3 * n, g.m_film.begin());
    std::copy(g.m_partials.begin() + 3 * n, g.m_partials.begin() + 4 * n, g.m_filmW.begin());
    for (unsigned int k = 0; k < g.m_pendingGroups; ++k) g.addGroup(g.m_partials.data() + 4 * n + (size_t)k * 7 * n);
    g.m_partialsPending = false; g.m_partialsExported = false;
    return PPG_OK;
}
int ppgo_film_ptrs(ppgo_ctx *ctx, float **rgb_sum, float **weight) {
    *rgb_sum = ctx->gpt.m_film.data(); *weight = ctx->gpt.m_filmW.data();
    return PPG_OK;
}
int ppgo_image_weight_ptr(ppgo_ctx *ctx, float **w) { *w = ctx->gpt.m_imageW.data(); return PPG_OK; }
int ppgo_image_ptrs(ppgo_ctx *ctx, float **image, float **sq_image) {
    *image = ctx->gpt.m_image.data(); *sq_image = ctx->gpt.m_squaredImage.data();
    return PPG_OK;
}

int ppgo_query_pdf(ppgo_ctx *ctx, uint32_t n, const float *positions, const float *dirs, float *pdf_out) {
    NEED_TREE
    for (uint32_t i = 0; i < n; ++i) {
        DTreeWrapper *d = ctx->gpt.m_sdTree->dTreeWrapper(Point(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]));
        pdf_out[i] = d->pdf(Vec(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    }
    return PPG_OK;
}
int ppgo_query_sample(ppgo_ctx *ctx, uint32_t n, const float *positions, uint64_t seed, float *dirs_out) {
    NEED_TREE
    for (uint32_t i = 0; i < n; ++i) {
        DTreeWrapper *d = ctx->gpt.m_sdTree->dTreeWrapper(Point(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]));
        Sampler s{ppg_path_key(seed, i, 0), 0};
        Vec v = d->sample(&s);
        dirs_out[3 * i] = v.x; dirs_out[3 * i + 1] = v.y; dirs_out[3 * i + 2] = v.z;
    }
    return PPG_OK;
}

// ---- known-answer hooks ----
int ppgo_ka_fresh_reset(float rho, uint32_t *num_nodes, int32_t *depth, float *pdf) {
    DTreeWrapper w;
    Modes m; m.acc = PPGO_ACC_FLOAT; m.adam = PPGO_ADAM_SEQUENTIAL;
    w.reset(20, rho);
    w.build(m);
    *num_nodes = (uint32_t)w.numNodes();
    *depth = w.depth();
    *pdf = w.pdf(Vec(0, 0, 1));
    return PPG_OK;
}
int ppgo_ka_refine(float Wt, uint64_t thr, uint32_t *n_leaves, uint32_t *n_nodes) {
    AABB box; box.min = Point(0.0f); box.max = Point(1.0f);
    STree t(box);
    t.nodes()[0].dTree.setStatisticalWeightBuilding(Wt);
    t.refine((size_t)thr, -1);
    uint32_t l = 0;
    for (auto &n : t.nodes()) l += n.isLeaf ? 1 : 0;
    *n_leaves = l; *n_nodes = (uint32_t)t.nodes().size();
    return PPG_OK;
}
int ppgo_ka_adam(int32_t n, float product, float wo_pdf, float bsdf_pdf, float dtree_pdf, float weight, int32_t loss, float *fraction) {
    DTreeWrapper w;
    Modes m; m.acc = PPGO_ACC_FLOAT; m.adam = PPGO_ADAM_SEQUENTIAL;
    DTreeRecord rec{Vec(0, 0, 1), 1.0f, product, wo_pdf, bsdf_pdf, dtree_pdf, weight, true /* isDelta: only the Adam half */};
    for (int i = 0; i < n; ++i) w.record(rec, EDNearest, loss == 1 ? EKL : EVariance, m);
    *fraction = w.bsdfSamplingFraction();
    return PPG_OK;
}
void ppgo_canonical_to_dir(float x, float y, float *d) {
    Vec v = DTreeWrapper::canonicalToDir(Point2{x, y});
    d[0] = v.x; d[1] = v.y; d[2] = v.z;
}
void ppgo_dir_to_canonical(const float *d, float *xy) {
    Point2 p = DTreeWrapper::dirToCanonical(Vec(d[0], d[1], d[2]));
    xy[0] = p.x; xy[1] = p.y;
}

// element-wise evaluation of the shared numerical contract (tests/test_detmath.py)
int ppgo_math_eval(int32_t op, uint32_t n, const float *a, const float *b, float *out0, float *out1) {
    for (uint32_t i = 0; i < n; ++i) {
        switch (op) {
            case 0: ppg_sincos(a[i], &out0[i], &out1[i]); break;
            case 1: out0[i] = ppg_atan2(a[i], b[i]); break;
            case 2: out0[i] = ppg_exp(a[i]); break;
            case 3: out0[i] = ppg_from_fixed(ppg_to_fixed(a[i])); break;
            case 4: out0[i] = ppg_rand((uint32_t)i * 2654435761u + 17u, (uint32_t)b[i]); break;
            case 5: out0[i] = ppg_powi(a[i], (int)b[i]); break;
            case 9: out0[i] = ppg_adam_learning_rate(0.01f, 0.9f, 0.999f, (int)a[i]); break;  // AdamOptimizer::step's learning rate at iteration a
            case 6: out0[i] = ppg_log(a[i]); break;
            case 7: out0[i] = ppg_pow(a[i], b[i]); break;
            case 8: {  // draw `dim` of the path (seed, pixel, sample index): a = pixel, b = seed << 16 | sample << 4 | dim, as bit patterns
                const uint32_t pixel = ppg_f2u(a[i]), w = ppg_f2u(b[i]);
                out0[i] = ppg_rand(ppg_path_key((uint64_t)(w >> 16), pixel, (w >> 4) & 0xfffu), w & 15u);
                break;
            }
            default: return PPG_ERR_INVALID;
        }
    }
    return PPG_OK;
}

int ppgo_dtree_exercise(int32_t acc_mode, int32_t directional_filter, float rho, uint32_t n, const float *xy, const float *irradiance,
                        const float *weight, uint32_t m, const float *query_xy, uint64_t seed, float *pdf_out, float *sample_xy_out,
                        uint32_t *num_nodes_out, float *node_sums_out, uint16_t *node_children_out, float *stat_weight_out,
                        float *tree_sum_out) {
    Modes md; md.acc = acc_mode; md.adam = PPGO_ADAM_SEQUENTIAL;
    DTreeWrapper w;
    EDirectionalFilter df = directional_filter ? EDBox : EDNearest;
    for (int round = 0; round < 2; ++round) {
        w.reset(20, rho);
        for (uint32_t i = 0; i < n; ++i) w.building.recordIrradiance(Point2{xy[2 * i], xy[2 * i + 1]}, irradiance[i], weight[i], df, md.acc);
        w.build(md);
    }
    for (uint32_t i = 0; i < m; ++i) {
        pdf_out[i] = w.sampling.pdf(Point2{query_xy[2 * i], query_xy[2 * i + 1]});
        Sampler s{ppg_path_key(seed, i, 0), 0};
        Point2 p = w.sampling.sample(&s);
        sample_xy_out[2 * i] = p.x; sample_xy_out[2 * i + 1] = p.y;
    }
    *num_nodes_out = (uint32_t)w.sampling.numNodes();
    for (size_t k = 0; k < w.sampling.numNodes(); ++k)
        for (int j = 0; j < 4; ++j) {
            node_sums_out[4 * k + j] = w.sampling.node(k).sum(j);
            node_children_out[4 * k + j] = w.sampling.node(k).child(j);
        }
    *stat_weight_out = w.sampling.statisticalWeight();
    *tree_sum_out = w.sampling.sumValue();
    return PPG_OK;
}

// BSDF plug-in interface of the supported materials, element-wise (tests/test_bsdfs.py: the chi-square-style checks the
// reference applies to its BSDFs, M/src/tests/test_chisquare.cpp).  wi / wo are in the local shading frame.
static std::vector<float> g_testRtrans;  // the slice ppgo_bsdf_* give a roughplastic material (tests only)
int ppgo_bsdf_set_rtrans(const float *slice, uint32_t samples) {
    g_testRtrans.assign(slice, slice + samples + 1);
    return PPG_OK;
}
static bool testSlice(Material &m) {
    if (m.type != PPG_BSDF_ROUGHPLASTIC) return true;
    if (g_testRtrans.size() < 3) return false;
    m.rt = g_testRtrans.data(); m.rtN = (uint32_t)g_testRtrans.size() - 1;
    return true;
}
int ppgo_bsdf_eval(const ppg_material *mat, uint32_t n, const float *wi, const float *wo, float *f_out, float *pdf_out) {
    Material m; static_cast<ppg_material &>(m) = *mat;
    if (m.type < 0 || m.type > PPG_BSDF_LAST || !testSlice(m)) return PPG_ERR_INVALID;
    m.configure();
    for (uint32_t i = 0; i < n; ++i) {
        BRec b; b.wi = Vec(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]); b.wo = Vec(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
        Spectrum f = BSDF::eval(m, b);
        f_out[3 * i] = f.x; f_out[3 * i + 1] = f.y; f_out[3 * i + 2] = f.z;
        pdf_out[i] = BSDF::pdf(m, b);
    }
    return PPG_OK;
}
int ppgo_bsdf_sample(const ppg_material *mat, uint32_t n, const float *wi, const float *sample_xy, float *wo_out, float *weight_out,
                     float *pdf_out, float *eta_out, int32_t *delta_out) {
    Material m; static_cast<ppg_material &>(m) = *mat;
    if (m.type < 0 || m.type > PPG_BSDF_LAST || !testSlice(m)) return PPG_ERR_INVALID;
    m.configure();
    for (uint32_t i = 0; i < n; ++i) {
        BRec b; b.wi = Vec(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
        Float pdf = 0;
        Sampler smp{ppg_path_key(77, i, 0), 0};  // the extra draw of roughdielectric::sample (roughdielectric.cpp:553)
        Spectrum w = BSDF::sample(m, b, pdf, Point2{sample_xy[2 * i], sample_xy[2 * i + 1]}, &smp);
        wo_out[3 * i] = b.wo.x; wo_out[3 * i + 1] = b.wo.y; wo_out[3 * i + 2] = b.wo.z;
        weight_out[3 * i] = w.x; weight_out[3 * i + 1] = w.y; weight_out[3 * i + 2] = w.z;
        pdf_out[i] = pdf; eta_out[i] = b.eta; delta_out[i] = b.sampledDelta ? 1 : 0;
    }
    return PPG_OK;
}
// EnvironmentMap, element-wise (tests): radiance and solid-angle density for world directions; sampled directions with value / pdf
static bool loadEnvMap(const ppg_envmap *em, EnvMap &m) {
    if (!em || !em->rgb || em->width == 0 || em->height == 0) return false;
    m.w = (int)em->width; m.h = (int)em->height; m.scale = em->scale;
    memcpy(m.R, em->to_world, sizeof m.R);
    m.texel.resize((size_t)m.w * m.h);
    for (size_t k = 0; k < m.texel.size(); ++k) m.texel[k] = Spectrum(em->rgb[3 * k], em->rgb[3 * k + 1], em->rgb[3 * k + 2]);
    m.configure();
    return m.valid;
}
int ppgo_envmap_eval(const ppg_envmap *em, uint32_t n, const float *dirs, float *rgb_out, float *pdf_out) {
    EnvMap m;
    if (!loadEnvMap(em, m)) return PPG_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        const Vec d(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        const Spectrum v = m.eval(d);
        rgb_out[3 * i] = v.x; rgb_out[3 * i + 1] = v.y; rgb_out[3 * i + 2] = v.z;
        pdf_out[i] = m.pdfDirection(m.toLocal(d));
    }
    return PPG_OK;
}
int ppgo_envmap_sample(const ppg_envmap *em, uint32_t n, const float *sample_xy, float *dir_out, float *weight_out, float *pdf_out) {
    EnvMap m;
    if (!loadEnvMap(em, m)) return PPG_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        Vec dl; Spectrum value; Float pdf;
        m.sampleDirection(Point2{sample_xy[2 * i], sample_xy[2 * i + 1]}, dl, value, pdf);
        const Vec d = m.toWorld(dl);
        const Spectrum wgt = pdf > 0 ? value / pdf : Spectrum(0.0f);
        dir_out[3 * i] = d.x; dir_out[3 * i + 1] = d.y; dir_out[3 * i + 2] = d.z;
        weight_out[3 * i] = wgt.x; weight_out[3 * i + 1] = wgt.y; weight_out[3 * i + 2] = wgt.z;
        pdf_out[i] = pdf;
    }
    return PPG_OK;
}
int ppgo_bsdf_flags(const ppg_material *mat, int32_t *is_smooth, int32_t *all_delta, int32_t *backside_or_transmission) {
    Material m; static_cast<ppg_material &>(m) = *mat;
    testSlice(m);
    m.configure();
    *is_smooth = BSDF::isSmooth(m); *all_delta = BSDF::allDelta(m); *backside_or_transmission = BSDF::hasBackSideOrTransmission(m);
    return PPG_OK;
}

}  // extern "C"
