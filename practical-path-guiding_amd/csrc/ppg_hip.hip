/*
 * ppg_hip.hip — libppg_hip.so: host side of the MI355X guided path tracer behind the C-ABI of include/ppg.h.
 *
 * Host responsibilities (thin): property parsing (GP:1014-1085), BVH build, pool management, the
 * iteration schedule of render()/renderSPP()/renderTime() (GP:1342-1585), the statistics log.  Everything per
 * path, per S-tree / D-tree node or per pixel runs in the kernels of ppg_kernels.h — including the SD-tree rebuild
 * between iterations (S-tree refine, D-tree reset and build); the host keeps a read-back mirror of the S-tree for
 * the log, the dumps and the readers.
 */
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture iterator calls memset from host code
#include <map>
#include <mutex>
#include <set>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ppg.h"
#include "../../include/ppg_testhooks.h"
#include "ppg_kernels.h"
#include "ppg_launch.h"

namespace {

thread_local std::string g_createError;

#define HIP_CHECK(expr)                                                                             \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ctx->error = std::string(#expr) + ": " + hipGetErrorString(_e);                         \
            return PPG_ERR_DEVICE;                                                                  \
        }                                                                                           \
    } while (0)

// Device memory blocks outlive the contexts that used them: a context's buffers grow while it renders (path state at begin_render, the
// optimiser's records with the first large round, sort scratch ...), and hipMalloc / hipFree of hundreds of megabytes are synchronous,
// millisecond-scale calls — rocprofv3 showed a 17 ms hole in a 190 ms KITCHEN render where one round's record buffers were grown.  Released
// blocks therefore go to a process-wide cache (per device, keyed by size) and the next reserve() — of this context or of the next one, e.g.
// the render after a warm-up render — takes a fitting block from it.  Blocks come back with stale contents, like fresh hipMalloc memory.
// Ordering: a block is released only by reserve() / the destructor on the host; everything that used it was enqueued earlier on the
// context's in-order stream (the second stream is joined back before the host continues past a batch), and its next user is enqueued later.
// The stream the calling thread's context works on (set by the C-ABI entry points that render): fresh blocks are touched on it and the
// hand-over of a released block is ordered behind it — not behind the legacy null stream, which would serialise every stream of the
// process.  Outside those entry points (scene set-up, destruction) it is null and the null stream's conservative ordering applies.
thread_local hipStream_t g_ctxStream = nullptr;
struct StreamScope {
    hipStream_t prev;
    explicit StreamScope(hipStream_t s) : prev(g_ctxStream) { g_ctxStream = s; }
    ~StreamScope() { g_ctxStream = prev; }
};

struct BlockCache {
    struct Block { void *p; hipEvent_t ready; };  // `ready`: recorded when the block was released, behind everything its last owner had enqueued
    std::mutex m;
    std::map<int, std::multimap<size_t, Block>> blocks;
    size_t held = 0;
    size_t maxHeld = 0;  // a third of the device's memory (set on first use): the cache must not starve other processes of the GPU
    static constexpr size_t kGranule = (size_t)2 << 20;
    static size_t roundUp(size_t bytes) { return (bytes + kGranule - 1) / kGranule * kGranule; }
    void *take(size_t bytes, size_t &got) {
        int dev = 0; (void)hipGetDevice(&dev);
        Block b{nullptr, nullptr};
        {
            std::lock_guard<std::mutex> g(m);
            auto &mm = blocks[dev];
            auto it = mm.lower_bound(bytes);
            if (it == mm.end() || it->first > std::max(2 * bytes, bytes + ((size_t)64 << 20))) return nullptr;
            b = it->second; got = it->first;
            held -= got; mm.erase(it);
        }
        // the previous owner may be another context on another stream: its work on the block must be over before the new owner touches it
        if (b.ready) { (void)hipEventSynchronize(b.ready); (void)hipEventDestroy(b.ready); }
        return b.p;
    }
    void give(void *p, size_t bytes) {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            if (!maxHeld) { size_t fr = 0, tot = 0; maxHeld = hipMemGetInfo(&fr, &tot) == hipSuccess ? tot / 3 : ((size_t)32 << 30); }
            static const bool debugAlloc = getenv("PPG_DEBUG_ALLOC") != nullptr;
            if (debugAlloc && held + bytes > maxHeld) fprintf(stderr, "[ppg alloc] hipFree %zu MiB (cache full: %zu MiB)\n", bytes >> 20, held >> 20);
            if (held + bytes <= maxHeld) {
                Block b{p, nullptr};
                if (hipEventCreateWithFlags(&b.ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(b.ready, g_ctxStream);
                else b.ready = nullptr;
                if (!b.ready) (void)hipDeviceSynchronize();
                blocks[dev].emplace(bytes, b); held += bytes;
                return;
            }
        }
        (void)hipFree(p);
    }
    // Before a context's streams are destroyed: the `ready` events of cached blocks may have been recorded on them, and the runtime
    // dereferences an event's stream when the event is waited for (seen as a spurious "operation not permitted when stream is
    // capturing" from hipEventSynchronize after the stream was gone).  The streams have been synchronised: the events are complete.
    void settle() {
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(m);
        for (auto &kv : blocks[dev])
            if (kv.second.ready) { (void)hipEventSynchronize(kv.second.ready); (void)hipEventDestroy(kv.second.ready); kv.second.ready = nullptr; }
    }
    // Scene set-up: spare blocks for what a render grows as it goes — the SD-tree's pools, the per-iteration images, the straggler sets' small
    // arrays: dozens of allocations of 2 - 30 MB whose sizes depend on the scene.  Each is a synchronous hipMalloc of 0.1 - 0.2 ms when the cache
    // is empty, i.e. in the FIRST render of a process (5 % of a 20-pass KITCHEN render); with spares in the cache they are not.  Once per device.
    void prewarm() {
        int dev = 0; (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(m);
            if (warmed.count(dev)) return;
            warmed.insert(dev);
        }
        static const struct { size_t mib; int n; } kSpares[] = {{2, 64}, {8, 24}, {32, 8}, {128, 2}};
        for (const auto &sp : kSpares)
            for (int k = 0; k < sp.n; ++k) {
                void *q = nullptr;
                const size_t bytes = sp.mib << 20;
                if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
                (void)hipMemsetAsync(q, 0, bytes, g_ctxStream);
                give(q, bytes);
            }
    }
    std::set<int> warmed;
    // give every cached block of the current device back to the driver (ppg_release_cached_memory; also when hipMalloc fails)
    void trim() {
        int dev = 0; (void)hipGetDevice(&dev);
        std::vector<Block> out;
        {
            std::lock_guard<std::mutex> g(m);
            for (auto &kv : blocks[dev]) { out.push_back(kv.second); held -= kv.first; }
            blocks[dev].clear();
        }
        for (Block &b : out) {
            if (b.ready) { (void)hipEventSynchronize(b.ready); (void)hipEventDestroy(b.ready); }
            (void)hipFree(b.p);
        }
    }
};
static BlockCache g_blockCache;

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    size_t bytes = 0;  // size of the block behind p
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), bytes(o.bytes) { o.p = nullptr; o.cap = 0; o.bytes = 0; }
    ~DevBuf() { if (p) g_blockCache.give(p, bytes); }
    hipError_t reserve(size_t n, bool keep = false) {
        if (n <= cap) return hipSuccess;
        const size_t want = BlockCache::roundUp(std::max(n, cap + cap / 2) * sizeof(T));
        size_t got = 0;
        T *np = (T *)g_blockCache.take(want, got);
        hipError_t e = hipSuccess;
        if (!np) {
            e = hipMalloc(&np, want);
            static const bool debugAlloc = getenv("PPG_DEBUG_ALLOC") != nullptr;  // every block asked of the driver (none in a steady state)
            if (debugAlloc) fprintf(stderr, "[ppg alloc] hipMalloc %zu MiB -> %s (cache holds %zu MiB)\n", want >> 20, e == hipSuccess ? "ok" : "FAILED: cache released", g_blockCache.held >> 20);
            if (e != hipSuccess) {  // memory parked in the cache is memory the driver cannot hand out: release it and try once more
                (void)hipGetLastError();
                g_blockCache.trim();
                e = hipMalloc(&np, want);
                if (e != hipSuccess) return e;
            }
            got = want;
            // touch fresh device memory once, where it is allocated (normally scene set-up), so that whatever the driver does lazily for a new
            // allocation is not paid inside the first render that reaches it: the first process on a freshly booted box sometimes ran a
            // 20-pass render 20-30 % slower than the second.  On the context's own stream when a render is under way.
            (void)hipMemsetAsync(np, 0, want, g_ctxStream);
        }
        if (keep && p && cap) e = hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, g_ctxStream);
        if (keep && p && cap && !g_ctxStream) (void)hipStreamSynchronize(nullptr);
        if (p) g_blockCache.give(p, bytes);
        p = np; bytes = got; cap = got / sizeof(T);
        return e;
    }
};

struct HostSNode {  // host mirror of one S-tree node (STreeNode, GP:740-845)
    int axis = 0;
    uint32_t child[2] = {0, 0};
    bool isLeaf() const { return child[0] == 0; }
};

// The stream of the commits that run beside the tail.  PPG_STREAM2_LOW=1: at the device's least priority, so that the tail's crowd phase is
// served first (experiment).
static hipError_t createSecondStream(hipStream_t *st) {
    const char *e = getenv("PPG_STREAM2_LOW");
    if (!e || !atoi(e)) return hipStreamCreate(st);
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    return hipStreamCreateWithPriority(st, hipStreamDefault, least);
}

int parseEnum(const char *s, const char *dflt, std::initializer_list<const char *> names) {
    std::string v = s ? s : dflt;
    int i = 0;
    for (const char *n : names) {
        if (v == n) return i;
        ++i;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// BVH2 construction (binned SAH), stands in for Scene::initialize → ShapeKDTree::build
// ------------------------------------------------------------------------------------------------
struct BvhBuilder {
    const float *pos; const uint32_t *idx;
    std::vector<uint32_t> order;
    std::vector<float> bmin, bmax, cent;  // per tri
    std::vector<BvhNode> nodes;
    float pad;
    int maxLeaf = 4;  // triangles per leaf (the BVH4 child code holds up to 8)

    struct Box { float lo[3], hi[3]; };
    static Box empty() { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } return b; }
    static void grow(Box &b, const float *lo, const float *hi) { for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], lo[a]); b.hi[a] = std::max(b.hi[a], hi[a]); } }
    static float area(const Box &b) { float d[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]}; if (d[0] < 0) return 0; return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }

    Box boundsOf(int first, int count) const {
        Box b = empty();
        for (int i = first; i < first + count; ++i) grow(b, &bmin[3 * order[i]], &bmax[3 * order[i]]);
        return b;
    }

    // returns (ref, n): n > 0 leaf [ref, ref+n), n == 0 interior node ref
    void build(int first, int count, int &ref, int &n, Box &box) {
        box = boundsOf(first, count);
        if (count <= maxLeaf) { ref = first; n = count; return; }
        Box cb = empty();
        for (int i = first; i < first + count; ++i) grow(cb, &cent[3 * order[i]], &cent[3 * order[i]]);
        int bestAxis = -1, bestSplit = -1; float bestCost = INFINITY;
        const int NB = 16;
        for (int a = 0; a < 3; ++a) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Box bb[NB]; int bc[NB];
            for (int k = 0; k < NB; ++k) { bb[k] = empty(); bc[k] = 0; }
            for (int i = first; i < first + count; ++i) {
                int k = std::min(NB - 1, (int)((cent[3 * order[i] + a] - cb.lo[a]) / ext * NB));
                grow(bb[k], &bmin[3 * order[i]], &bmax[3 * order[i]]); bc[k]++;
            }
            float la[NB]; int lc[NB]; Box acc = empty(); int c = 0;
            for (int k = 0; k < NB; ++k) { if (bc[k]) grow(acc, bb[k].lo, bb[k].hi); c += bc[k]; la[k] = area(acc); lc[k] = c; }
            acc = empty(); c = 0;
            for (int k = NB - 1; k > 0; --k) {
                if (bc[k]) grow(acc, bb[k].lo, bb[k].hi); c += bc[k];
                if (lc[k - 1] == 0 || c == 0) continue;
                float cost = la[k - 1] * lc[k - 1] + area(acc) * c;
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
            }
        }
        int mid;
        if (bestAxis < 0) {
            mid = first + count / 2;
        } else {
            float ext = cb.hi[bestAxis] - cb.lo[bestAxis], lo = cb.lo[bestAxis];
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int k = std::min(NB - 1, (int)((cent[3 * t + bestAxis] - lo) / ext * NB));
                return k < bestSplit;
            });
            mid = (int)(it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        int me = (int)nodes.size();
        nodes.emplace_back();
        int r0, n0, r1, n1; Box b0, b1;
        build(first, mid - first, r0, n0, b0);
        build(mid, first + count - mid, r1, n1, b1);
        BvhNode &nd = nodes[me];
        for (int a = 0; a < 3; ++a) { nd.lo0[a] = b0.lo[a] - pad; nd.hi0[a] = b0.hi[a] + pad; nd.lo1[a] = b1.lo[a] - pad; nd.hi1[a] = b1.hi[a] + pad; }
        nd.c0 = r0; nd.n0 = n0; nd.c1 = r1; nd.n1 = n1;
        ref = me; n = 0;
    }

    // ---- collapse the binary tree to 4-wide nodes (children = the 2..4 descendants obtained by repeatedly opening
    // the interior child with the largest surface area) ----
    std::vector<Bvh4Node> nodes4;
    struct Ref { int ref, n; float lo[3], hi[3]; };
    static float areaRef(const Ref &r) { float d[3] = {r.hi[0] - r.lo[0], r.hi[1] - r.lo[1], r.hi[2] - r.lo[2]}; return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
    void childrenOf(int node2, Ref out[2]) const {
        const BvhNode &nd = nodes[node2];
        out[0].ref = nd.c0; out[0].n = nd.n0; out[1].ref = nd.c1; out[1].n = nd.n1;
        for (int a = 0; a < 3; ++a) { out[0].lo[a] = nd.lo0[a]; out[0].hi[a] = nd.hi0[a]; out[1].lo[a] = nd.lo1[a]; out[1].hi[a] = nd.hi1[a]; }
    }
    int make4(int node2) {
        Ref ch[4]; int m = 2;
        childrenOf(node2, ch);
        if (ch[1].n < 0) m = 1;  // single-leaf root
        while (m < 4) {
            int pick = -1; float best = -1;
            for (int k = 0; k < m; ++k) if (ch[k].n == 0 && areaRef(ch[k]) > best) { best = areaRef(ch[k]); pick = k; }
            if (pick < 0) break;
            Ref two[2]; childrenOf(ch[pick].ref, two);
            ch[pick] = two[0]; ch[m++] = two[1];
        }
        int me = (int)nodes4.size();
        nodes4.emplace_back();
        Bvh4Node nd;
        for (int k = 0; k < 4; ++k) {
            if (k < m) {
                nd.lox[k] = ch[k].lo[0]; nd.loy[k] = ch[k].lo[1]; nd.loz[k] = ch[k].lo[2];
                nd.hix[k] = ch[k].hi[0]; nd.hiy[k] = ch[k].hi[1]; nd.hiz[k] = ch[k].hi[2];
                if (ch[k].n > 0) nd.child[k] = ~((ch[k].ref << 3) | (ch[k].n - 1));
                else nd.child[k] = make4(ch[k].ref);
            } else {
                nd.lox[k] = nd.loy[k] = nd.loz[k] = 0; nd.hix[k] = nd.hiy[k] = nd.hiz[k] = 0; nd.child[k] = PPG_BVH4_EMPTY;
            }
            nd.pad[k] = 0;
        }
        nodes4[me] = nd;
        return me;
    }

    // quantise the child boxes of every BVH4 node (Bvh4QNode, ppg_device.h); conservativeness is verified in the device's arithmetic
    std::vector<Bvh4QNode> nodes4q;
    static float scaleOf(unsigned int e) { const uint32_t bits = e << 23; float f; memcpy(&f, &bits, 4); return f; }
    void quantise() {
        nodes4q.resize(nodes4.size());
        for (size_t i = 0; i < nodes4.size(); ++i) {
            const Bvh4Node &nd = nodes4[i];
            Bvh4QNode q{};
            const float *lo[3] = {nd.lox, nd.loy, nd.loz}, *hi[3] = {nd.hix, nd.hiy, nd.hiz};
            float org[3];
            unsigned int ex[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                float mn = INFINITY, mx = -INFINITY;
                for (int k = 0; k < 4; ++k) if (nd.child[k] != PPG_BVH4_EMPTY) { mn = std::min(mn, lo[a][k]); mx = std::max(mx, hi[a][k]); }
                if (!(mn <= mx)) { mn = 0; mx = 0; }
                org[a] = mn;
                int x = 0;
                (void)std::frexp((double)(mx - mn) / 255.0, &x);  // value = m 2^x, m in [0.5, 1): 2^x >= value
                // exponent range: bvh4_children scales the ray by 2^-e and 1/d (up to 1e30, safe_inv) by 2^e — both stay finite within 64..154
                // (cells of 1e-19 .. 1.3e8 scene units; ppg_set_scene refuses scenes of more than 1e9 units)
                int e = std::max(64, std::min(154, x + 127));
                for (;;) {  // find a cell size for which all four boxes fit into 0..255 conservatively
                    const float s = scaleOf((unsigned int)e);
                    bool ok = true;
                    unsigned int wl = 0, wh = 0;
                    for (int k = 0; k < 4 && ok; ++k) {
                        if (nd.child[k] == PPG_BVH4_EMPTY) { wl |= 255u << (8 * k); continue; }  // inverted box: never entered
                        long ql = (long)std::floor(((double)lo[a][k] - (double)mn) / (double)s), qh = (long)std::ceil(((double)hi[a][k] - (double)mn) / (double)s);
                        ql = std::max(0l, std::min(255l, ql)); qh = std::max(0l, qh);
                        // neither the float decode (trace_closest4_wave) nor the exact plane (bvh4_children) may cut into the box
                        auto below = [&](long qq) { return mn + (float)qq * s <= lo[a][k] && (double)mn + (double)qq * (double)s <= (double)lo[a][k]; };
                        auto above = [&](long qq) { return mn + (float)qq * s >= hi[a][k] && (double)mn + (double)qq * (double)s >= (double)hi[a][k]; };
                        while (ql > 0 && !below(ql)) --ql;
                        while (qh <= 255 && !above(qh)) ++qh;
                        if (qh > 255 || !below(ql)) { ok = false; break; }
                        wl |= (unsigned int)ql << (8 * k); wh |= (unsigned int)qh << (8 * k);
                    }
                    if (ok) { qlo[a] = wl; qhi[a] = wh; break; }
                    if (++e > 154) { e = 154; qlo[a] = 0; qhi[a] = 0xffffffffu; break; }  // (unreachable below 1e9 scene units)
                }
                ex[a] = (unsigned int)e;
            }
            q.ox = org[0]; q.oy = org[1]; q.oz = org[2];
            q.exps = ex[0] | (ex[1] << 8) | (ex[2] << 16);
            q.qlox = qlo[0]; q.qloy = qlo[1]; q.qloz = qlo[2]; q.qhix = qhi[0]; q.qhiy = qhi[1]; q.qhiz = qhi[2];
            for (int k = 0; k < 4; ++k) q.child[k] = nd.child[k];
            nodes4q[i] = q;
        }
    }

    // The TOP CUT of the tree for trace_closest4_wave (a whole wave walking ONE ray): up to 64 subtrees that partition the scene — the root's
    // descendants after opening, level by level and then by surface area, as many interior entries as fit —, each with its box exactly as the
    // traversal would decode it from its parent (float decode of Bvh4QNode) and its child reference.  The wave tests all of them in ONE step,
    // one lane each, instead of descending the first three levels one dependent step at a time.  Two float4 per entry: (lo, ref) (hi, -).
    std::vector<float> topCut;
    void buildTopCut() {
        struct Ent { float lo[3], hi[3]; int ref; };
        auto childrenOfQ = [&](int node, Ent out[4]) {
            const Bvh4QNode &q = nodes4q[(size_t)node];
            const float sx = scaleOf(q.exps & 255u), sy = scaleOf((q.exps >> 8) & 255u), sz = scaleOf((q.exps >> 16) & 255u);
            int m = 0;
            for (int k = 0; k < 4; ++k) {
                if (q.child[k] == PPG_BVH4_EMPTY) continue;
                const int sh = 8 * k;
                Ent e;
                e.lo[0] = q.ox + (float)((q.qlox >> sh) & 255u) * sx; e.lo[1] = q.oy + (float)((q.qloy >> sh) & 255u) * sy; e.lo[2] = q.oz + (float)((q.qloz >> sh) & 255u) * sz;
                e.hi[0] = q.ox + (float)((q.qhix >> sh) & 255u) * sx; e.hi[1] = q.oy + (float)((q.qhiy >> sh) & 255u) * sy; e.hi[2] = q.oz + (float)((q.qhiz >> sh) & 255u) * sz;
                e.ref = q.child[k];
                out[m++] = e;
            }
            return m;
        };
        std::vector<Ent> cut;
        {
            Ent ch[4];
            const int m = nodes4q.empty() ? 0 : childrenOfQ(0, ch);
            for (int k = 0; k < m; ++k) cut.push_back(ch[k]);
        }
        auto areaOf = [](const Ent &e) { const float d[3] = {e.hi[0] - e.lo[0], e.hi[1] - e.lo[1], e.hi[2] - e.lo[2]}; return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]; };
        for (;;) {  // open the interior entry with the largest box while the cut stays within 64 entries
            int pick = -1; float best = -1;
            for (size_t k = 0; k < cut.size(); ++k) if (cut[k].ref >= 0 && areaOf(cut[k]) > best) { best = areaOf(cut[k]); pick = (int)k; }
            if (pick < 0) break;
            Ent ch[4];
            const int m = childrenOfQ(cut[(size_t)pick].ref, ch);
            if (cut.size() - 1 + (size_t)m > 64) break;
            cut.erase(cut.begin() + pick);
            for (int k = 0; k < m; ++k) cut.push_back(ch[k]);
        }
        topCut.assign(8 * cut.size(), 0.0f);
        for (size_t k = 0; k < cut.size(); ++k) {
            float *o = &topCut[8 * k];
            o[0] = cut[k].lo[0]; o[1] = cut[k].lo[1]; o[2] = cut[k].lo[2]; memcpy(o + 3, &cut[k].ref, 4);
            o[4] = cut[k].hi[0]; o[5] = cut[k].hi[1]; o[6] = cut[k].hi[2];
        }
    }

    void run(const float *positions, const uint32_t *indices, uint32_t nTris, float padAbs) {
        runTree(positions, indices, nTris, padAbs);
        buildTopCut();
    }
    void runTree(const float *positions, const uint32_t *indices, uint32_t nTris, float padAbs) {
        pos = positions; idx = indices; pad = padAbs;
        order.resize(nTris); bmin.resize(3 * (size_t)nTris); bmax.resize(3 * (size_t)nTris); cent.resize(3 * (size_t)nTris);
        for (uint32_t t = 0; t < nTris; ++t) {
            order[t] = t;
            for (int a = 0; a < 3; ++a) {
                float v0 = pos[3 * idx[3 * t] + a], v1 = pos[3 * idx[3 * t + 1] + a], v2 = pos[3 * idx[3 * t + 2] + a];
                bmin[3 * t + a] = std::min(v0, std::min(v1, v2)); bmax[3 * t + a] = std::max(v0, std::max(v1, v2));
                cent[3 * t + a] = 0.5f * (bmin[3 * t + a] + bmax[3 * t + a]);
            }
        }
        nodes.clear();
        nodes.reserve(nTris);
        int ref, n; Box b;
        nodes.emplace_back();  // root placeholder, must be an interior node
        if (nTris == 0) {  // spheres only: one node without children
            nodes4.assign(1, Bvh4Node{});
            for (int k = 0; k < 4; ++k) nodes4[0].child[k] = PPG_BVH4_EMPTY;
            quantise();
            return;
        }
        if (nTris <= 4) {
            BvhNode &nd = nodes[0];
            Box bb = boundsOf(0, (int)nTris);
            for (int a = 0; a < 3; ++a) { nd.lo0[a] = bb.lo[a] - pad; nd.hi0[a] = bb.hi[a] + pad; nd.lo1[a] = 0; nd.hi1[a] = 0; }
            nd.c0 = 0; nd.n0 = (int)nTris; nd.c1 = 0; nd.n1 = -1;
            nodes4.clear(); make4(0); quantise();
            return;
        }
        nodes.pop_back();
        build(0, (int)nTris, ref, n, b);  // nTris > 4 ⇒ root is interior and lands at index 0
        nodes4.clear(); nodes4.reserve(nodes.size() / 2 + 1);
        make4(0);
        quantise();
    }
};

struct KernelTimer {
    struct Rec { hipEvent_t a, b; int id; uint64_t units; };
    std::vector<std::string> names;
    std::vector<double> ms;
    std::vector<uint64_t> launches, units;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    bool enabled = false;
    int idOf(const char *n) {
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i;
        names.push_back(n); ms.push_back(0); launches.push_back(0); units.push_back(0);
        return (int)names.size() - 1;
    }
    hipEvent_t ev() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; (void)hipEventCreate(&e); return e; }
    void resolve() {
        for (auto &r : pending) {
            (void)hipEventSynchronize(r.b);
            float t = 0; (void)hipEventElapsedTime(&t, r.a, r.b);
            ms[r.id] += t; launches[r.id]++; units[r.id] += r.units;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        pending.clear();
    }
    void reset() { resolve(); for (size_t i = 0; i < ms.size(); ++i) { ms[i] = 0; launches[i] = 0; units[i] = 0; } }
};

}  // namespace

// A launch of whole groups of a final iteration's passes (include/ppg.h "Final iteration: groups of passes"): `batch` passes = groups of
// groupPasses passes (the last one may be shorter; or a part of ONE group when groupPasses >= batch), the first starting at pass
// firstPass of the render, consecutive groups of the launch stridePasses apart; group k accumulates into slot slot0 + k * slotStride.
// addCount > 0 (one GPU): the launch's slots are added to image and film right after its film kernel (k_add_groups).
struct GroupLaunch { unsigned int firstPass, groupPasses, stridePasses, slot0, slotStride; bool wholeFilm; unsigned int addCount; };

struct ppg_ctx {
    // properties (GP:1014-1085)
    int nee = 0, sampleCombination = 1, spatialFilter = 0, directionalFilter = 0, loss = 0, budgetType = 1;
    int sdTreeMaxMemory = -1, sTreeThreshold = 12000, sppPerPass = 4, rrDepth = 5, maxDepth = -1;
    float dTreeThreshold = 0.01f, bsdfSamplingFraction = 0.5f, budget = 300.0f;
    bool dumpSDTree = false, strictNormals = false, hideEmitters = false;
    uint64_t seed = 0;
    int device = 0;
    std::string dumpPrefix, error;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // k_commit of the finished paths runs here while k_tail finishes the stragglers on `stream`
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // The end of a round of the optimiser — k_splat_sorted on stream2, the order / apply kernels on stream3 — runs BESIDE the beginning of the
    // next batch (k_generate and the first k_trace, which need neither the building tree nor the sampling fractions): k_adam_apply is a set
    // of serial chains that leaves most of the GPU idle.  treePending: that work has been enqueued and not yet been waited for; joinTree()
    // makes the context's stream wait for it — before the first kernel that reads the SD-tree or the fractions, and in every entry point.
    hipStream_t stream3 = nullptr;
    hipEvent_t evTreeFork = nullptr, evSplatDone = nullptr, evAdamDone = nullptr;
    bool treePending = false;
    void joinTree() {
        if (!treePending) return;
        (void)hipStreamWaitEvent(stream, evSplatDone, 0);
        (void)hipStreamWaitEvent(stream, evAdamDone, 0);
        treePending = false;
    }
    // Entry points that re-size or release buffers (ppg_set_scene, ppg_set_shard, ppg_begin_render): a render that was cancelled or failed
    // may have left kernels on the side streams that no join has waited for — wait for all of them on the host, then forget what they owed.
    void quiesce() {
        joinTree();
        if (stream2) (void)hipStreamSynchronize(stream2);
        if (stream3) (void)hipStreamSynchronize(stream3);
        if (stream4) (void)hipStreamSynchronize(stream4);
        if (stream) (void)hipStreamSynchronize(stream);
        for (Stragglers &g : strag) { g.pending = false; g.aside = false; g.film = nullptr; }
    }
    // STRAGGLERS (see "Stragglers" below renderBatch's helpers): the handful of paths of a batch that are still alive at depth `defer_depth`
    // leave k_tail, are finished by a second k_tail on stream4 BESIDE the next batch, and what the batch still owes — its film kernel, the
    // stragglers' commit — is done by the next batch (drainStragglers) or at the end of the call (flushStragglers).
    hipStream_t stream4 = nullptr;
    hipEvent_t evStragFork = nullptr, evStragDone = nullptr;
    struct Stragglers {
        DevBuf<float4> rec, vert;           // [cap][8] path records as k_tail wrote them; [maxVertices][n][4 | 6] vertex slots
        DevBuf<unsigned int> orig, iota, ticket, base, origSorted, perm;
        DevBuf<unsigned long long> count;   // [0] stragglers (k_tail's StragOut), as k_tail's `total` afterwards
        DevBuf<unsigned char> nv8;
        DevBuf<float> neeCos, theta;        // theta: the optimiser's variables of the stragglers' round (DevTree::theta_frozen)
        size_t iotaN = 0;
        PathState P{};                      // the compact state: n paths, path j = path orig[j] of the batch
        unsigned int n = 0;
        bool pending = false;               // a batch is owed its film kernel (and its stragglers' commit)
        bool aside = false;                 // ... and their k_tail was put on stream4 (evStragDone)
        bool commit = false, adam = false;  // the stragglers recorded vertices / those leave optimiser records (a round)
        std::function<void()> film;         // the batch's film kernel(s), reading liKeep
    } strag[2];                             // [stragCur]: the set the current batch fills; the other one: the previous batch's, while it is pending
    int stragCur = 0;
    DevBuf<float4> d_liKeep;                // Li of every path of the batch whose film kernel is owed (consumed before the next copy is taken)
    Stragglers &stragPrev() { return strag[stragCur ^ 1]; }
    int tuneSplitDepth = PPG_ADAM_DEFER_DEPTH;  // PPG_SPLIT_DEPTH: the depth at which a batch WITHOUT optimiser records hands its stragglers over (pure
                                                // scheduling: final iterations, renders without a learned fraction; 0 = never).  Rounds of the
                                                // optimiser use PPG_ADAM_DEFER_DEPTH — there the depth is part of the result (include/ppg.h)
    bool tuneFinalHalves = false;               // PPG_FINAL_HALVES: a final iteration that fits one launch is rendered in two (renderFinalGroups)
    // rounds by image region (include/ppg.h ppg_set_adam_regions): the owned pixels grouped by the region of their 32x32 block
    int adamRegions = 0;
    DevBuf<unsigned int> d_regionPixels;
    std::vector<unsigned int> regionOffset;   // [regions + 1] into d_regionPixels
    int regionsBuilt = 0;                     // the lists are valid for this many regions (0: none; reset when the shard or the film changes)
    int deferDepthAdam = PPG_ADAM_DEFER_DEPTH;  // (tests lower it through ppg_debug_set_defer_depth to meet many stragglers in small scenes)
    int joinStragglers() {  // the side stream's work is over as far as the context's stream is concerned (entry points, error paths)
        int rc = 0;
        for (Stragglers &g : strag) if (g.pending && g.aside) { g.aside = false; rc = (int)hipStreamWaitEvent(stream, evStragDone, 0); }
        return rc;
    }
    DevBuf<unsigned char> d_straggler;  // [path] 1 = still alive when the persistent-thread tail took over
    DevBuf<unsigned char> d_nv8;        // [path] vertex slots k_commit takes of the path (k_commit_prepare)

    // scene
    bool haveScene = false;
    bool fullMaterials = false;
    DevBuf<float4> d_tris, d_accel, d_accelSmall, d_normals, d_materials, d_emitters, d_emTris, d_emNrm;
    DevBuf<float> d_rtrans;  // ppg_scene.rtrans (roughplastic slices)
    DevBuf<float4> d_spheres;  // 4 float4 per analytic sphere (DevScene::spheres)
    DevBuf<float2> d_uvs;  // bitmap textures: per-triangle texture coordinates, the texel arrays, the DevTex table
    std::vector<DevBuf<float4>> d_texTexels;
    DevBuf<DevTex> d_textures;
    DevBuf<float4> d_emTexels;  // image-based environment emitter: texels, cdfs, row weights
    DevBuf<float> d_emCdfRows, d_emCdfCols, d_emRowWeights;
    DevBuf<float> d_emSel, d_emArea, d_neeCos;
    DevBuf<int4> d_emInfo;
    DevBuf<BvhNode> d_bvh;
    DevBuf<Bvh4QNode> d_bvh4;
    DevBuf<float4> d_bvhTop;  // BvhBuilder::topCut
    DevScene scene{};
    float aabbMin[3], aabbMax[3];  // Scene::getAABB()
    int W = 0, H = 0;

    // shard
    int shardRank = 0, shardWorld = 1, tileSize = 32;
    bool pathsReady = false;
    int maxBatch = 1;
    DevBuf<unsigned int> d_pixels;
    unsigned int nPix = 0;
    // final iteration of a sharded render: whole groups of passes over ALL pixels instead of all passes over this rank's tiles
    // (include/ppg.h "Final iteration: groups of passes")
    DevBuf<unsigned int> d_pixelsAll;
    unsigned int nPixAll = 0;
    DevBuf<float> d_partials;         // group slots (7 floats per pixel each); sharded: preceded by a film head (4 floats per pixel)
    unsigned int partialSlots = 0;    // slots the buffer holds
    bool partialsPending = false;     // sharded: this rank's slots wait for the exchange (ppg_final_partials / ppg_final_partials_commit)
    bool partialsExported = false;
    unsigned int pendingGroups = 0;
    uint64_t samplesLocal = 0;        // samples this rank rendered in the current performRenderPasses

    // path state
    DevBuf<float4> d_ray_o, d_ray_d, d_thr, d_li, d_hit, d_vd, d_vthr, d_vbsdf, d_vrad, d_vo, d_vvox;
    DevBuf<uint4> d_misc;
    DevBuf<float4> d_pathRec, d_vertexRec;  // interleaved layout (PathState Field, ppg_kernels.h): 8 float4 per path, 4 / 6 per vertex slot
    DevBuf<uint4> d_miscCompact;
    bool aosPaths = false;
    int pathLayout = 1;  // 1 soa, 2 aos, 3 pack (allocPaths; PPG_PATH_LAYOUT)
    DevBuf<unsigned int> d_queue[2], d_qcount[2], d_qtotal, d_queueSorted, d_qcommon;
    DevBuf<unsigned char> d_sortKeys;
    int maxBatchFinal = 1;  // passes per batch in the final iteration (nothing is recorded: no vertex slots needed)
    DevBuf<BlockStats> d_stats;
    Queues queues{};
    int nBlocks = 4096;  // persistent workgroups of the path kernels (16 per CU; KITCHEN 720p: 2048 → 76.5, 4096 → 81.3, 8192 → 75.0 Msamples/s)
    PathState paths{};
    int maxVertices = 0;

    // film
    DevBuf<float> d_image, d_sq, d_imageW, d_film, d_filmW, d_var, d_lum, d_tmp;
    float *h_lum = nullptr;  // pinned staging of d_lum (variance sum on the host, finishPasses)
    size_t h_lumCap = 0;
    BlockStats *h_stats = nullptr;  // pinned staging of d_stats
    size_t h_statsCap = 0;
    std::vector<DevBuf<float>> images;  // inverse-variance copies (weight-normalised)
    std::vector<float> variances;

    // SD-tree
    bool treeAlive = false;
    std::vector<HostSNode> snodes;
    std::vector<LeafHdr> hdr;  // host mirror (valid after build / refine)
    std::vector<unsigned int> leaves;
    DevBuf<int4> d_stree;
    DevBuf<LeafHdr> d_hdr;
    DevBuf<SNode> d_snodes[2];
    int cur = 0;  // d_snodes[cur] = sampling pool
    size_t nSamplingNodes = 0, nBuildingNodes = 0;
    DevBuf<ushort4> d_bchild;
    DevBuf<unsigned long long> d_bacc, d_bweight, d_total, d_bweightRep;
    // sampling-fraction optimiser: the records of the current round (include/ppg.h "Learning the BSDF sampling fraction")
    DevBuf<unsigned long long> d_adamKeys[2];
    DevBuf<unsigned int> d_adamIdx[2], d_adamNv, d_adamBase, d_adamCount;
    DevBuf<AdamRec> d_adamRecs, d_adamRecsOut;
    DevBuf<float4> d_splat;       // a round's splat records (k_commit_records), at the positions of the optimiser's records
    bool sortedCommit = false;    // this round commits through records + sort + k_splat_sorted instead of k_commit
    unsigned int adamFlagShift = 0;  // key bit of "a splat only" in such a round: just above the leaf bits
    DevBuf<unsigned char> d_sortTemp, d_orderTemp;
    DevBuf<unsigned int> d_adamLeafCount[2], d_adamLeafOrder[2];  // k_adam_apply's order of the D-trees: most records first
    size_t adamIota = 0;          // d_adamIdx[0][0 .. adamIota) holds the identity permutation
    bool adamFast = false;        // record positions known in advance (DevTree::adam_base)
    bool adamActive = false;      // a round of the optimiser is being rendered
    bool inHook = false, hookReplaced = false;
    uint64_t hookCount = 0;
    int hookPhase = 0;            // 0: before the round's records are applied, 1: after (sharded optimiser, include/ppg.h)
    bool ownerMode = false;       // phase 0 asked for the records by owner: phase 1 follows
    DevBuf<unsigned int> d_adamState;             // [world * segment][6] (ppg_adam_state)
    DevBuf<unsigned long long> d_ownerBounds;     // [world + 1] first record of every owner
    // unbounded paths: live paths after each bulk bounce of the last batch → how many bulk bounces the next batch runs before k_tail
    DevBuf<unsigned int> d_bounceCounts, d_ticket;
    unsigned int *h_round = nullptr;  // pinned: [0..63] bounce counts, [64] Adam record count / overflow, [65] Σ nV, [68..69] paths handed to k_tail, [70..71] stragglers
    // the previous batch's survival curve (live paths after each wavefront bounce it ran) sizes the schedule of the next batch
    int prevBounces = 0;
    unsigned int prevPaths = 0, prevLive[64] = {};
    int bounceMargin = 3;             // PPG_BOUNCE_MARGIN: bounces launched beyond the predicted need (the device skips what it does not need)
    // Tuning switches, read ONCE from the environment by ppg_create (DESIGN.md "Tuning switches"); none of them changes a result.
    unsigned int tailThreshold = 0;   // PPG_TAIL_THRESHOLD: live paths below which k_tail takes over (0 = automatic: max(tailMin, paths / tailDiv))
    // PPG_TAIL_MIN, PPG_TAIL_DIV (KITCHEN 720p: 131072 / 16 -> 786432 / 12: +1.5 % at 127 passes, +5 % at 20; with the tail's state in registers
    // a plateau from 1.6 M to 4 M: 786432 -> 2 M another +1.5 % at 20 passes, equal at 127)
    unsigned int tailMin = 2097152, tailDiv = 12;
    size_t tuneBatchPaths = 0;        // PPG_BATCH_PATHS: paths in flight per batch of passes (0 = automatic)
    int tuneBlocks = 0;               // PPG_BLOCKS: persistent workgroups of the path kernels (0 = 4096)
    bool tuneForceBvh = false;        // PPG_FORCE_BVH: trace small scenes through the BVH as well
    bool tuneFuse = false;            // PPG_FUSE: trace small scenes inside k_generate / k_shade
    bool tuneNoSort = false;          // PPG_NO_SORT: do not sort the queue slices by BSDF type before k_shade<FULL>
    bool tuneNoSortFirst = false;     // PPG_NO_SORT_FIRST: the first bounce of a batch unsorted through the complete k_shade<FULL>
    bool tuneNoSplit = false;         // PPG_NO_SPLIT: one k_shade<FULL> over the whole sorted slice instead of k_shade<.., MSET_COMMON> + the rest
    bool tuneNoOverlap = false;       // PPG_NO_OVERLAP: k_commit after k_tail on one stream instead of beside it
    bool tuneNoSortedCommit = false;  // PPG_NO_SORTED_COMMIT: a round of the optimiser commits with k_commit (global atomics) instead of records + sort + k_splat_sorted
    unsigned int tuneSplatLdsNodes = PPG_SPLAT_NODES;  // PPG_SPLAT_LDS_NODES: k_splat_sorted stages D-trees of up to this many nodes in LDS (<= PPG_SPLAT_NODES)
    bool tuneNoAside = false;         // PPG_NO_ASIDE: the optimiser's order / apply kernels stay on the context's stream (k_splat_sorted still runs beside them)
    bool tuneAdamUnordered = false;   // PPG_ADAM_UNORDERED: k_adam_apply takes the D-trees in leaf order instead of busiest first
    int tuneBulkBounces = -1;         // PPG_BULK_BOUNCES: fixed number of wavefront bounces before k_tail takes over (-1 = adaptive)
    bool debugBatch = false;          // PPG_DEBUG_BATCH: one line per batch on stderr (paths, live paths after every bulk bounce, tail time)
    int tuneFinalBatch = 0;           // PPG_FINAL_BATCH: passes per batch of the final iteration (0 = 64)
    int tuneTailBlocks = 0;           // PPG_TAIL_BLOCKS: workgroups of k_tail when k_commit runs beside it (0 = automatic)
    int tunePathLayout = 0;           // PPG_PATH_LAYOUT = soa | aos | pack: layout of the per-path state (0 = automatic, allocPaths)
    int tuneBlocksSmall = 2048;       // PPG_BLOCKS_SMALL: workgroups of the wavefront kernels for batches of at most tuneSmallPaths paths (0 = nBlocks for every batch)
    size_t tuneSmallPaths = 13000000; // PPG_SMALL_PATHS.  KITCHEN 720p (r06_experiments.json s23, s27): 2048 for batches up to 4 M / 16 M / 32 M paths: +0.1 / +1.1 / +1.2 % on the
                                      // driver's command (its final iteration is one batch of 12 M), +0.2 % at 127 passes, -0.3 % at 1023 with 16 M (its 14.7 M-path rounds): 13 M
    bool tuneSortKernel = false;      // PPG_SORT_KERNEL=1: the BSDF-type sort of the queue slices as a launch of its own (k_sort_slices) instead of inside k_trace
    int tuneBvhLeaf = 4;              // PPG_BVH_LEAF: triangles per BVH leaf (1..8).  KITCHEN 720p, driver's command: 4 -> 3 +3.5 % once the node test had become
                                      // cheap (130.9 -> 135.6, A/B on one box; with the world-space decode 2 / 3 / 4 / 6 / 8 gave 125.8 / 125.9 / 124.5 / 119.6 / 115.2)
    float tuneBvhPad = 2e-6f;         // PPG_BVH_PAD: box padding relative to the scene extent
    DevBuf<unsigned int> d_leaves, d_counts, d_offsets, d_grid;
    DevBuf<unsigned int> d_dfs[2], d_refEv, d_refLv, d_refEvOff, d_refLvOff;  // S-tree refine: leaves in the reference's (right-first) visiting order
    int dfsCur = 0;
    unsigned int dfsCount = 0;
    int ldsNodes = 0, ldsTris = 0;  // scene part cached in LDS by k_trace
    float treeMin[3], treeMax[3], treeExt[3];
    bool isBuilt = false, isFinalIter = false, doNee = false;
    int iter = 0, passesRendered = 0, passesRenderedThisIter = 0, passesLocal = 0;
    std::chrono::steady_clock::time_point startTime, passStart;
    std::atomic<bool> cancelled{false};
    bool cancelSeen = false;  // the render under way has acted on the flag (seesCancel): ppg_begin_render then does not apply it to the next one
    bool seesCancel() { if (!cancelled.load()) return false; cancelSeen = true; return true; }
    float lastVariance = 0;
    ppg_pass_stats lastStats{};
    KernelTimer timer;
#define PPG_TAIL_LOG 256
    DevBuf<unsigned int> d_tailLongest;  // [PPG_TAIL_LOG] longest path (bounces) finished by each k_tail launch of the current performRenderPasses
    unsigned int tailLaunches = 0;
    uint64_t tailLongestSum = 0;  // sum over all k_tail launches of the longest path each finished (bounces): the tails' critical path
    uint64_t shadeCommonRays = 0;  // rays through k_shade<.., MSET_COMMON> while kernel timing is on
    uint64_t bvhNodesVisited = 0, bvhTrisTested = 0;  // by k_trace while kernel timing is on (the roofline's node / triangle counts)
    ppg_pass_hook passHook = nullptr;
    void *passHookUser = nullptr;
    ppg_stop_hook stopHook = nullptr;
    void *stopHookUser = nullptr;
#ifdef PPG_PROBE
    DevBuf<unsigned long long> d_probe;  // development builds: cycle sums of the lone path's bounce (ppg_kernels.h PROBE_MARK), printed by endRender
#endif

    DevTree devTree() {
        DevTree T{};
        T.stree = d_stree.p; T.hdr = d_hdr.p; T.snodes = d_snodes[cur].p; T.bchild = d_bchild.p; T.bacc = d_bacc.p;
        T.bweight = d_bweight.p;
        T.bweight_rep = d_bweightRep.p;
        T.adam_keys = adamActive ? d_adamKeys[0].p : nullptr; T.adam_recs = adamActive ? d_adamRecs.p : nullptr;
        T.adam_count = d_adamCount.p; T.adam_base = (adamActive && adamFast) ? d_adamBase.p : nullptr;
        T.adam_cap = adamActive ? (unsigned int)std::min<size_t>(std::min(d_adamRecs.cap, d_adamKeys[0].cap), 0xfffffff0u) : 0u;  // (both are written at the same position)
        for (int a = 0; a < 3; ++a) { T.aabb_min[a] = treeMin[a]; T.aabb_ext[a] = treeExt[a]; T.aabb_max[a] = treeMax[a]; }
        T.is_built = isBuilt ? 1 : 0;
        T.grid = d_grid.p;
        return T;
    }
    RenderParams params() const {
        RenderParams R{};
        R.nee = nee; R.spatial_filter = spatialFilter; R.directional_filter = directionalFilter; R.loss = loss;
        R.bsdf_sampling_fraction = bsdfSamplingFraction; R.rr_depth = rrDepth; R.max_depth = maxDepth;
        R.strict_normals = strictNormals; R.hide_emitters = hideEmitters; R.spp = sppPerPass;
        R.is_final_iter = isFinalIter; R.do_nee = doNee; R.seed = seed; R.pass_index = (unsigned int)passesRendered;
        R.max_vertices = maxVertices; R.img_pixels = (unsigned int)W * (unsigned int)H;
#ifdef PPG_PROBE
        R.probe = d_probe.p;
#endif
        return R;
    }
};

namespace {

}  // namespace
void ppg_launch_shade(int variant, const ShadeLaunch &a) {
    switch (variant >> 1) {
        case 0: ppg_launch_shade_pair0(variant, a); break;
        case 1: ppg_launch_shade_pair1(variant, a); break;
        case 2: ppg_launch_shade_pair2(variant, a); break;
        default: ppg_launch_shade_pair3(variant, a); break;
    }
}
void ppg_launch_tail(int variant, const TailLaunch &a) {
    switch (variant >> 1) {
        case 0: ppg_launch_tail_pair0(variant, a); break;
        case 1: ppg_launch_tail_pair1(variant, a); break;
        case 2: ppg_launch_tail_pair2(variant, a); break;
        default: ppg_launch_tail_pair3(variant, a); break;
    }
}
void ppg_launch_commit(int sf, int df, const CommitLaunch &a) { ppg_launch_commit_all(sf, df, a); }
void ppg_launch_commit_records(int sf, const CommitLaunch &a) { ppg_launch_commit_records_all(sf, a); }
void ppg_launch_splat(int df, const SplatLaunch &a) { ppg_launch_splat_all(df, a); }
namespace {

template <typename F> void timedLaunch(ppg_ctx *ctx, const char *name, uint64_t units, F &&launch) {
    if (!ctx->timer.enabled) { launch(); return; }
    KernelTimer::Rec r; r.a = ctx->timer.ev(); r.b = ctx->timer.ev(); r.id = ctx->timer.idOf(name); r.units = units;
    (void)hipEventRecord(r.a, ctx->stream);
    launch();
    (void)hipEventRecord(r.b, ctx->stream);
    ctx->timer.pending.push_back(r);
}

int gridFor(size_t n, int block = PPG_BLOCK, int maxBlocks = 256 * 8) {
    size_t b = (n + block - 1) / block;
    return (int)std::max<size_t>(1, std::min<size_t>(b, (size_t)maxBlocks));
}

float elapsedSeconds(std::chrono::steady_clock::time_point start) {  // GP:1428-1432
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start);
    return (float)ms.count() / 1000;
}

// nodes: also copy the S-tree nodes and leaf headers host → device (only needed when the host mirror is the newer copy, i.e. for the
// single root node at the start of a render; afterwards the device refines the tree and the host mirror is downloaded from it)
int uploadTree(ppg_ctx *ctx, bool nodes) {
    size_t n = ctx->snodes.size();
    std::vector<int4> st(nodes ? n : 0);
    if (nodes) {
        for (size_t i = 0; i < n; ++i) st[i] = make_int4(ctx->snodes[i].axis, (int)ctx->snodes[i].child[0], (int)ctx->snodes[i].child[1], 0);
        HIP_CHECK(ctx->d_stree.reserve(n));
        HIP_CHECK(ctx->d_hdr.reserve(n));
        HIP_CHECK(hipMemcpyAsync(ctx->d_stree.p, st.data(), n * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(ctx->d_hdr.p, ctx->hdr.data(), n * sizeof(LeafHdr), hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->leaves.clear();
    for (size_t i = 0; i < n; ++i) if (ctx->snodes[i].isLeaf()) ctx->leaves.push_back((unsigned int)i);
    HIP_CHECK(ctx->d_leaves.reserve(ctx->leaves.size()));
    HIP_CHECK(ctx->d_counts.reserve(ctx->leaves.size()));
    HIP_CHECK(ctx->d_offsets.reserve(ctx->leaves.size()));
    HIP_CHECK(hipMemcpyAsync(ctx->d_leaves.p, ctx->leaves.data(), ctx->leaves.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(ctx->d_bweight.reserve(n));
    HIP_CHECK(hipMemsetAsync(ctx->d_bweight.p, 0, n * 8, ctx->stream));
    HIP_CHECK(ctx->d_bweightRep.reserve(n * PPG_REPLICAS));
    HIP_CHECK(hipMemsetAsync(ctx->d_bweightRep.p, 0, n * PPG_REPLICAS * 8, ctx->stream));
    if (n >= (1u << 27)) { ctx->error = "S-tree exceeds 2^27 nodes"; return PPG_ERR_NOMEM; }
    const unsigned int cells = PPG_GRID_DIM * PPG_GRID_DIM * PPG_GRID_DIM;
    HIP_CHECK(ctx->d_grid.reserve(cells));
    hipLaunchKernelGGL(k_build_grid, dim3((cells + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_stree.p, ctx->d_grid.p);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // st is a stack-local staging buffer
    return PPG_OK;
}

// STree::refine (GP:957-998) + subdivide (GP:876-895) on the device (k_refine_count → scans → k_refine_fill, ppg_kernels.h);
// the host mirror (node array, leaf headers) is then downloaded — it feeds the statistics log, the dumps and the readers.
int refineDevice(ppg_ctx *ctx, size_t sTreeThreshold, int maxMB) {
    auto &nodes = ctx->snodes;
    auto &hdr = ctx->hdr;
    if (maxMB >= 0) {
        size_t foot = 0;  // approxMemoryFootprint with the reference's sizeof (QuadTreeNode 24 B, DTree 40 B)
        for (size_t i = 0; i < nodes.size(); ++i) foot += nodes[i].isLeaf() ? ((size_t)hdr[i].b_num * 24 + 40) + ((size_t)hdr[i].s_num * 24 + 40) : 2 * (24 + 40);
        if (foot / 1000000 >= (size_t)maxMB) return PPG_OK;
    }
    const unsigned int nl = ctx->dfsCount, nOld = (unsigned int)nodes.size();
    hipStream_t s = ctx->stream;
    HIP_CHECK(ctx->d_refEv.reserve(nl)); HIP_CHECK(ctx->d_refLv.reserve(nl)); HIP_CHECK(ctx->d_refEvOff.reserve(nl)); HIP_CHECK(ctx->d_refLvOff.reserve(nl));
    HIP_CHECK(ctx->d_total.reserve(2));
    hipLaunchKernelGGL(k_refine_count, dim3((nl + 255) / 256), dim3(256), 0, s, ctx->d_hdr.p, ctx->d_dfs[ctx->dfsCur].p, nl, (float)sTreeThreshold,
                       ctx->d_refEv.p, ctx->d_refLv.p);
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, s, ctx->d_refEv.p, ctx->d_refEvOff.p, nl, ctx->d_total.p);
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, s, ctx->d_refLv.p, ctx->d_refLvOff.p, nl, ctx->d_total.p + 1);
    unsigned long long totals[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(totals, ctx->d_total.p, 16, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (totals[0] == 0) return PPG_OK;  // no leaf exceeded the threshold
    const unsigned long long nNew = (unsigned long long)nOld + 2ull * totals[0];
    if (nNew >= (1ull << 27)) { ctx->error = "S-tree exceeds 2^27 nodes"; return PPG_ERR_NOMEM; }
    HIP_CHECK(ctx->d_stree.reserve((size_t)nNew, true));
    HIP_CHECK(ctx->d_hdr.reserve((size_t)nNew, true));
    HIP_CHECK(ctx->d_dfs[ctx->dfsCur ^ 1].reserve((size_t)totals[1]));
    hipLaunchKernelGGL(k_refine_fill, dim3(nl), dim3(256), 0, s, ctx->d_stree.p, ctx->d_hdr.p, ctx->d_dfs[ctx->dfsCur].p, nl, nOld, ctx->d_refEv.p,
                       ctx->d_refEvOff.p, ctx->d_refLvOff.p, ctx->d_dfs[ctx->dfsCur ^ 1].p);
    ctx->dfsCur ^= 1;
    ctx->dfsCount = (unsigned int)totals[1];
    std::vector<int4> st((size_t)nNew);
    hdr.resize((size_t)nNew);
    HIP_CHECK(hipMemcpyAsync(st.data(), ctx->d_stree.p, st.size() * sizeof(int4), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(hdr.data(), ctx->d_hdr.p, hdr.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    nodes.resize((size_t)nNew);
    for (size_t i = 0; i < nodes.size(); ++i) { nodes[i].axis = st[i].x; nodes[i].child[0] = (uint32_t)st[i].y; nodes[i].child[1] = (uint32_t)st[i].z; }
    return PPG_OK;
}

int resetSDTree(ppg_ctx *ctx) {  // GP:1108-1113
    ctx->joinTree();
    double thr = std::sqrt(std::ldexp(1.0, ctx->iter) * ctx->sppPerPass / 4) * ctx->sTreeThreshold;
    int rc = refineDevice(ctx, (size_t)thr, ctx->sdTreeMaxMemory);
    if (rc) return rc;
    rc = uploadTree(ctx, false);
    if (rc) return rc;
    unsigned int nl = (unsigned int)ctx->leaves.size();
    HIP_CHECK(ctx->d_total.reserve(1));
    DevTree T = ctx->devTree();
    int blocks = (int)((nl + 127) / 128);
    timedLaunch(ctx, "k_dtree_reset<count>", nl, [&] {
        hipLaunchKernelGGL(k_dtree_reset<false>, dim3(blocks), dim3(128), 0, ctx->stream, T, ctx->d_leaves.p, nl, 20, ctx->dTreeThreshold,
                           ctx->d_counts.p, (ushort4 *)nullptr);
    });
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_counts.p, ctx->d_offsets.p, nl, ctx->d_total.p);
    unsigned long long total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, ctx->d_total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (total > 0xfffffff0ull) { ctx->error = "building D-tree pool exceeds 2^32 nodes"; return PPG_ERR_NOMEM; }
    ctx->nBuildingNodes = (size_t)total;
    HIP_CHECK(ctx->d_bchild.reserve(total));
    HIP_CHECK(ctx->d_bacc.reserve(total * 4));
    HIP_CHECK(ctx->d_snodes[ctx->cur ^ 1].reserve(total));
    HIP_CHECK(hipMemsetAsync(ctx->d_bacc.p, 0, total * 4 * 8, ctx->stream));
    hipLaunchKernelGGL(k_assign_blocks, dim3(blocks), dim3(128), 0, ctx->stream, T, ctx->d_leaves.p, nl, ctx->d_counts.p, ctx->d_offsets.p);
    T = ctx->devTree();
    timedLaunch(ctx, "k_dtree_reset<fill>", nl, [&] {
        hipLaunchKernelGGL(k_dtree_reset<true>, dim3(blocks), dim3(128), 0, ctx->stream, T, ctx->d_leaves.p, nl, 20, ctx->dTreeThreshold,
                           ctx->d_counts.p, ctx->d_bchild.p);
    });
    HIP_CHECK(hipGetLastError());
    return PPG_OK;
}

int foldWeights(ppg_ctx *ctx) {
    ctx->joinTree();
    unsigned int nn = (unsigned int)ctx->snodes.size();
    if (!ctx->d_bweightRep.p || nn == 0) return PPG_OK;
    hipLaunchKernelGGL(k_fold_replicas, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_bweight.p, ctx->d_bweightRep.p, nn);
    HIP_CHECK(hipGetLastError());
    return PPG_OK;
}

int buildSDTree(ppg_ctx *ctx, ppg_tree_stats *st) {  // GP:1115-1189
    ctx->joinTree();  // (the last round's splats and optimiser steps may still be running beside the context's stream)
    { int rc = foldWeights(ctx); if (rc) return rc; }
    unsigned int nl = (unsigned int)ctx->leaves.size();
    DevTree T = ctx->devTree();
    int blocks = (int)((nl + 127) / 128);
    timedLaunch(ctx, "k_dtree_build", nl, [&] {
        hipLaunchKernelGGL(k_dtree_build, dim3(blocks), dim3(128), 0, ctx->stream, T, ctx->d_leaves.p, nl, ctx->d_snodes[ctx->cur ^ 1].p);
    });
    HIP_CHECK(hipGetLastError());
    ctx->cur ^= 1;  // sampling = building
    ctx->nSamplingNodes = ctx->nBuildingNodes;
    HIP_CHECK(hipMemcpyAsync(ctx->hdr.data(), ctx->d_hdr.p, ctx->hdr.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // statistics, in S-tree node order like forEachDTreeWrapperConst (GP:1138-1174)
    int maxDepth = 0, minDepth = std::numeric_limits<int>::max();
    float avgDepth = 0, maxAvgRadiance = 0, minAvgRadiance = std::numeric_limits<float>::max(), avgAvgRadiance = 0;
    size_t maxNodes = 0, minNodes = std::numeric_limits<size_t>::max();
    float avgNodes = 0, maxSW = 0, minSW = std::numeric_limits<float>::max(), avgSW = 0;
    int nPoints = 0, nPointsNodes = 0;
    uint64_t totalNodes = 0;
    for (unsigned int leaf : ctx->leaves) {
        const LeafHdr &h = ctx->hdr[leaf];
        const int depth = h.s_depth;
        maxDepth = std::max(maxDepth, depth); minDepth = std::min(minDepth, depth); avgDepth += depth;
        float avgRadiance = 0;
        if (h.s_statw != 0) { const float factor = 1 / (PPG_PI_F * 4 * h.s_statw); avgRadiance = factor * h.s_sum; }
        maxAvgRadiance = ppg_max(maxAvgRadiance, avgRadiance); minAvgRadiance = ppg_min(minAvgRadiance, avgRadiance);
        avgAvgRadiance += avgRadiance;
        if (h.s_num > 1) {
            const size_t nodes = h.s_num;
            maxNodes = std::max(maxNodes, nodes); minNodes = std::min(minNodes, nodes); avgNodes += nodes; ++nPointsNodes;
        }
        totalNodes += h.s_num;
        maxSW = ppg_max(maxSW, h.s_statw); minSW = ppg_min(minSW, h.s_statw); avgSW += h.s_statw;
        ++nPoints;
    }
    if (nPoints > 0) {
        avgDepth /= nPoints; avgAvgRadiance /= nPoints;
        if (nPointsNodes > 0) avgNodes /= nPointsNodes;
        avgSW /= nPoints;
    }
    if (st) {
        st->min_depth = minDepth; st->max_depth = maxDepth; st->avg_depth = avgDepth;
        st->min_mean_radiance = minAvgRadiance; st->avg_mean_radiance = avgAvgRadiance; st->max_mean_radiance = maxAvgRadiance;
        st->min_nodes = minNodes; st->max_nodes = maxNodes; st->avg_nodes = avgNodes;
        st->min_stat_weight = minSW; st->avg_stat_weight = avgSW; st->max_stat_weight = maxSW;
        st->n_leaves = (uint32_t)nPoints; st->n_stree_nodes = (uint32_t)ctx->snodes.size(); st->n_dtree_nodes = totalNodes;
    }
    ctx->isBuilt = true;
    return PPG_OK;
}

int allocPaths(ppg_ctx *ctx) {
    // owned pixel list
    std::vector<unsigned int> pix;
    int tilesX = (ctx->W + ctx->tileSize - 1) / ctx->tileSize;
    for (int y = 0; y < ctx->H; ++y)
        for (int x = 0; x < ctx->W; ++x) {
            int t = (y / ctx->tileSize) * tilesX + (x / ctx->tileSize);
            if (ctx->shardWorld <= 1 || t % ctx->shardWorld == ctx->shardRank) pix.push_back((unsigned int)(y * ctx->W + x));
        }
    ctx->nPix = (unsigned int)pix.size();
    ctx->regionsBuilt = 0;
    HIP_CHECK(ctx->d_pixels.reserve(std::max<size_t>(1, pix.size())));
    if (!pix.empty()) HIP_CHECK(hipMemcpy(ctx->d_pixels.p, pix.data(), pix.size() * 4, hipMemcpyHostToDevice));
    ctx->nPixAll = (unsigned int)((size_t)ctx->W * ctx->H);
    if (ctx->shardWorld > 1) {  // the final iteration's groups are rendered over the whole film
        std::vector<unsigned int> all(ctx->nPixAll);
        for (unsigned int k = 0; k < ctx->nPixAll; ++k) all[k] = k;
        HIP_CHECK(ctx->d_pixelsAll.reserve(std::max<size_t>(1, all.size())));
        if (!all.empty()) HIP_CHECK(hipMemcpy(ctx->d_pixelsAll.p, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    }
    // Pass batching: the passes of an iteration are independent (frozen sampling tree, accumulate-only building
    // tree), so up to maxBatch of them run as ONE set of launches over nPix * spp * batch paths.  Sample indices,
    // per-pixel accumulation order and all integer statistics are unchanged, i.e. results are bit-identical; what
    // changes is that small images / tile shards (multi-GPU strong scaling) still fill the GPU and that the serial
    // tail of unbounded paths is paid once per batch.  When the BSDF sampling fraction is learned, a batch is one
    // ROUND of the optimiser (include/ppg.h: ppg_adam_round_passes — part of the algorithm, not a tuning knob).
    // With a time budget and no rounds the clock is checked after every pass, as in the reference (GP:1259-1262).
    {
        const size_t perPass = std::max<size_t>(1, (size_t)ctx->nPix * ctx->sppPerPass);
        size_t target = 16u << 20;  // ~16 M paths in flight (measured on cbox-720p: 4 M 996, 8 M 1007, 16 M 1034, 32 M 1041 Msamples/s)
        {   // ... but at most ~24 GB of vertex slots (288 GB HBM: ample, this only bounds allocation time)
            int slots = PPG_MAX_VERTICES;
            if (ctx->maxDepth > 0) slots = std::max(1, std::min(PPG_MAX_VERTICES, ctx->maxDepth - 1));
            const size_t perPath = (size_t)slots * (ctx->spatialFilter != SF_NEAREST ? 96 : 64) + 96;
            target = std::min<size_t>(target, (size_t)24e9 / perPath);
        }
        if (ctx->tuneBatchPaths) target = ctx->tuneBatchPaths;
        ctx->maxBatch = ctx->budgetType == 1 ? 1 : (int)std::max<size_t>(1, std::min<size_t>(64, target / perPass));
        if (ctx->loss != LOSS_NONE) {
            if ((uint64_t)ctx->W * ctx->H * ctx->sppPerPass > (1ull << PPG_ADAM_PATH_BITS)) { ctx->error = "width * height * sppPerPass exceeds 2^27 with a bsdfSamplingFractionLoss"; return PPG_ERR_INVALID; }
            ctx->maxBatch = std::max(ctx->maxBatch, (int)ppg_adam_round_passes(ctx->sppPerPass, ctx->W, ctx->H, 1 << 30));
        }
    }
    {   // the final iteration records nothing: its batches need path state only (96 B per path), so they can be larger — the serial tail of
        // unbounded paths is then paid once per 64 passes
        const size_t perPass = std::max<size_t>(1, (size_t)(ctx->shardWorld > 1 ? ctx->nPixAll : ctx->nPix) * ctx->sppPerPass);
        // PPG_FINAL_BATCH=256 (2^28 paths): KITCHEN 700x400 at 2400 spp 5.2 -> 4.9 s, 720p at 511 passes 169 -> 175 Msamples/s — but the 23 GB of
        // path state it sizes cost the FIRST process on a freshly booted GPU box 20 % of a 20-pass render (86 vs 110 Msamples/s; the second
        // process is at par), so the default stays at 64 passes.
        const size_t capPasses = ctx->tuneFinalBatch ? (size_t)ctx->tuneFinalBatch : 64, capPaths = (size_t)1 << (capPasses > 64 ? 28 : 26);
        ctx->maxBatchFinal = ctx->budgetType == 1 ? 1 : (int)std::max<size_t>((size_t)ctx->maxBatch, std::min<size_t>(capPasses, capPaths / perPass));
        if (ctx->tuneBatchPaths) ctx->maxBatchFinal = ctx->maxBatch;
    }
    size_t n = (size_t)ctx->nPix * ctx->sppPerPass * (size_t)ctx->maxBatch;
    const size_t nFinal = (size_t)(ctx->shardWorld > 1 ? ctx->nPixAll : ctx->nPix) * ctx->sppPerPass * (size_t)ctx->maxBatchFinal;
    if (n > 0xfffffff0ull || nFinal > 0xfffffff0ull) { ctx->error = "too many paths per pass"; return PPG_ERR_INVALID; }
    const size_t nTrain = std::max<size_t>(1, n);  // vertex slots: training batches only
    size_t nn = std::max<size_t>(1, std::max(n, nFinal));
    ctx->maxVertices = PPG_MAX_VERTICES;
    if (ctx->maxDepth > 0) ctx->maxVertices = std::max(1, std::min(PPG_MAX_VERTICES, ctx->maxDepth - 1));
    // layout of the per-path state: one array per field (soa), interleaved 128-byte records (aos), or two 64-byte records (pack).  Interleaving
    // helps the late, scattered bounces — after compaction and the sort by material a lane holds an arbitrary path, and five 16-byte accesses
    // to five arrays cost five sectors — and hurts the early, coalesced ones.  Measured on MI355X, round 4 (profiles/r04_experiments.json,
    // Msamples/s soa / pack / aos): KITCHEN 1023 passes 202 / 207 / 209, 127 passes 183 / 185 / 187, 20 passes equal; torus-class 1080p
    // 129 / 138 / 142; SPACESHIP 1080p 1008 / 925 / 874; cbox-720p 1350 / 1170 / 1038.  So: interleaved for paths of unbounded depth over a BVH
    // scene (long random walks: most bounces are late ones), one array per field otherwise; PPG_PATH_LAYOUT = soa | aos | pack overrides.
    if (ctx->tunePathLayout == 0) ctx->pathLayout = (ctx->maxDepth < 0 && ctx->scene.n_tris > 64) ? 2 : 1; else ctx->pathLayout = ctx->tunePathLayout;
    // PPG_PATH_LAYOUT=pack: two 64-byte records per path — (throughput, Li, key / flags) and (ray origin, direction, hit) — so that a scattered
    // path costs k_shade two sectors to read and two to write instead of five and five, and k_trace one; the vertex slots stay one array per field
    const bool packPaths = ctx->pathLayout == 3;
    ctx->aosPaths = ctx->pathLayout == 2 || packPaths;
    if (ctx->aosPaths) { HIP_CHECK(ctx->d_pathRec.reserve(nn * 8)); HIP_CHECK(ctx->d_miscCompact.reserve(nn)); }
    else {
        HIP_CHECK(ctx->d_ray_o.reserve(nn)); HIP_CHECK(ctx->d_ray_d.reserve(nn)); HIP_CHECK(ctx->d_thr.reserve(nn));
        HIP_CHECK(ctx->d_li.reserve(nn)); HIP_CHECK(ctx->d_hit.reserve(nn)); HIP_CHECK(ctx->d_misc.reserve(nn));
    }
    if (ctx->tuneBlocks) ctx->nBlocks = ctx->tuneBlocks;
    const size_t nb = (size_t)ctx->nBlocks;
    const size_t chunks = (nn + PPG_DCHUNK - 1) / PPG_DCHUNK;  // (PPG_DCHUNK is a multiple of PPG_CHUNK: covers the first bounce's dealing too)
    const size_t cap = ((chunks + nb - 1) / nb) * PPG_DCHUNK;
    for (int k = 0; k < 2; ++k) { HIP_CHECK(ctx->d_queue[k].reserve(cap * nb)); HIP_CHECK(ctx->d_qcount[k].reserve(nb)); }
    HIP_CHECK(ctx->d_stats.reserve(nb)); HIP_CHECK(ctx->d_qtotal.reserve(1));
    // k_scan_counts / k_gather_slices write one offset per persistent workgroup and the live count: these two buffers are otherwise sized by
    // the SD-tree code (one entry per S-tree leaf — a single one in the first iteration)
    HIP_CHECK(ctx->d_offsets.reserve(nb)); HIP_CHECK(ctx->d_total.reserve(2));
    HIP_CHECK(ctx->d_bounceCounts.reserve(72)); HIP_CHECK(ctx->d_ticket.reserve(1));
    HIP_CHECK(ctx->d_adamCount.reserve(2));
    if (!ctx->h_round) HIP_CHECK(hipHostMalloc((void **)&ctx->h_round, 72 * sizeof(unsigned int), hipHostMallocDefault));
    ctx->queues.items[0] = ctx->d_queue[0].p; ctx->queues.items[1] = ctx->d_queue[1].p;
    ctx->queues.count[0] = ctx->d_qcount[0].p; ctx->queues.count[1] = ctx->d_qcount[1].p;
    ctx->queues.cap = (unsigned int)cap; ctx->queues.stats = ctx->d_stats.p; ctx->queues.n_blocks = (unsigned int)nb;
    ctx->queues.n_common = nullptr;
    if (ctx->fullMaterials && !ctx->tuneNoSort) {
        HIP_CHECK(ctx->d_queueSorted.reserve(cap * nb)); HIP_CHECK(ctx->d_sortKeys.reserve(cap * nb));
        if (!ctx->tuneNoSplit) { HIP_CHECK(ctx->d_qcommon.reserve(nb)); ctx->queues.n_common = ctx->d_qcommon.p; }
    }
    size_t nv = nTrain * (size_t)ctx->maxVertices;
    const bool filtered = ctx->spatialFilter != SF_NEAREST;
    PathState &P = ctx->paths;
    P.n_paths = (unsigned int)n; P.n_pix = ctx->nPix; P.pixels = ctx->d_pixels.p;
    if (packPaths) {
        float4 *a = ctx->d_pathRec.p, *b = ctx->d_pathRec.p + 4 * nn;
        P.thr = {a, 4}; P.li = {a + 1, 4}; P.misc = {reinterpret_cast<uint4 *>(a + 2), 4};
        P.ray_o = {b, 4}; P.ray_d = {b + 1, 4}; P.hit = {b + 2, 4};
        HIP_CHECK(ctx->d_vd.reserve(nv)); HIP_CHECK(ctx->d_vthr.reserve(nv)); HIP_CHECK(ctx->d_vbsdf.reserve(nv)); HIP_CHECK(ctx->d_vrad.reserve(nv));
        if (filtered) { HIP_CHECK(ctx->d_vo.reserve(nv)); HIP_CHECK(ctx->d_vvox.reserve(nv)); }
        P.v_d = {ctx->d_vd.p, 1}; P.v_thr = {ctx->d_vthr.p, 1}; P.v_bsdf = {ctx->d_vbsdf.p, 1}; P.v_rad = {ctx->d_vrad.p, 1};
        P.v_o = {filtered ? ctx->d_vo.p : nullptr, 1}; P.v_vox = {filtered ? ctx->d_vvox.p : nullptr, 1};
    } else if (ctx->aosPaths) {
        const unsigned int vs = filtered ? 6u : 4u;
        HIP_CHECK(ctx->d_vertexRec.reserve(nv * vs));
        float4 *r = ctx->d_pathRec.p, *v = ctx->d_vertexRec.p;
        P.ray_o = {r, 8}; P.ray_d = {r + 1, 8}; P.thr = {r + 2, 8}; P.li = {r + 3, 8}; P.hit = {r + 4, 8}; P.misc = {reinterpret_cast<uint4 *>(r + 5), 8};
        P.v_d = {v, vs}; P.v_thr = {v + 1, vs}; P.v_bsdf = {v + 2, vs}; P.v_rad = {v + 3, vs};
        P.v_o = {filtered ? v + 4 : nullptr, vs}; P.v_vox = {filtered ? v + 5 : nullptr, vs};
    } else {
        HIP_CHECK(ctx->d_vd.reserve(nv)); HIP_CHECK(ctx->d_vthr.reserve(nv)); HIP_CHECK(ctx->d_vbsdf.reserve(nv)); HIP_CHECK(ctx->d_vrad.reserve(nv));
        if (filtered) { HIP_CHECK(ctx->d_vo.reserve(nv)); HIP_CHECK(ctx->d_vvox.reserve(nv)); }
        P.ray_o = {ctx->d_ray_o.p, 1}; P.ray_d = {ctx->d_ray_d.p, 1}; P.thr = {ctx->d_thr.p, 1}; P.li = {ctx->d_li.p, 1}; P.hit = {ctx->d_hit.p, 1};
        P.misc = {ctx->d_misc.p, 1};
        P.v_d = {ctx->d_vd.p, 1}; P.v_thr = {ctx->d_vthr.p, 1}; P.v_bsdf = {ctx->d_vbsdf.p, 1}; P.v_rad = {ctx->d_vrad.p, 1};
        P.v_o = {filtered ? ctx->d_vo.p : nullptr, 1}; P.v_vox = {filtered ? ctx->d_vvox.p : nullptr, 1};
    }
    P.nee_cos = nullptr;
    if (ctx->scene.env.w != 0 && ctx->nee != NEE_NEVER) { HIP_CHECK(ctx->d_neeCos.reserve(nn)); P.nee_cos = ctx->d_neeCos.p; }
    return PPG_OK;
}

// Buffers that otherwise grow with the first large round of the optimiser — the record arrays and the sort's — sized where the path state is
// sized (scene set-up), from the pass schedule an spp budget implies (renderSPP): a render then finds them in place instead of paying
// hipMalloc + first touch of a few GB in its second and third iteration (the first 20-pass render after a 5-pass warm-up ran 6 % below the
// later ones).  An estimate — 7 records per path plus max_vertices positions for every path k_tail may be handed —, not a limit: reserve()
// still grows what turns out too small.
int presizeRounds(ppg_ctx *ctx) {
    g_blockCache.prewarm();
    if (ctx->loss == LOSS_NONE || ctx->budgetType != 0 || ctx->spatialFilter == SF_BOX) return PPG_OK;
    const int nPasses = (int)std::ceil((size_t)ctx->budget / (float)ctx->sppPerPass);
    int rendered = 0, largest = 0;
    for (int it = 0; rendered < nPasses; ++it) {  // renderSPP, GP:1342-1426
        const int remaining = nPasses - rendered;
        int n = std::min(remaining, 1 << std::min(it, 30));
        if (remaining - n < 2 * n) n = remaining;
        if (n < remaining && it > 0) largest = std::max(largest, n);  // a training iteration with rounds (the first one has none: not built yet)
        rendered += n;
    }
    if (largest == 0) return PPG_OK;
    const size_t roundPaths = (size_t)ppg_adam_round_passes(ctx->sppPerPass, ctx->W, ctx->H, largest) * ctx->nPix * ctx->sppPerPass;
    const size_t handed = ctx->maxDepth < 0 ? std::min(roundPaths, std::max<size_t>(ctx->tailMin, roundPaths / ctx->tailDiv)) : 0;
    const size_t positions = std::min<size_t>(roundPaths * 7 + handed * (size_t)ctx->maxVertices, 0xfffffff0u);
    for (int k = 0; k < 2; ++k) { HIP_CHECK(ctx->d_adamKeys[k].reserve(positions)); HIP_CHECK(ctx->d_adamIdx[k].reserve(positions)); }
    HIP_CHECK(ctx->d_adamRecs.reserve(positions)); HIP_CHECK(ctx->d_splat.reserve(positions));
    HIP_CHECK(ctx->d_adamNv.reserve(roundPaths)); HIP_CHECK(ctx->d_adamBase.reserve(roundPaths));
    HIP_CHECK(ctx->d_nv8.reserve(roundPaths)); HIP_CHECK(ctx->d_straggler.reserve(roundPaths));
    if (handed) for (auto &g : ctx->strag) { HIP_CHECK(g.rec.reserve(handed * 8)); HIP_CHECK(g.orig.reserve(handed)); HIP_CHECK(g.count.reserve(2)); HIP_CHECK(g.ticket.reserve(1)); }
    {
        size_t bytes = 0;
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, ctx->d_adamKeys[0].p, ctx->d_adamKeys[1].p, ctx->d_adamIdx[0].p, ctx->d_adamIdx[1].p, positions, 0u, 64u, ctx->stream));
        HIP_CHECK(ctx->d_sortTemp.reserve(std::max<size_t>(bytes, 16)));
    }
    return PPG_OK;
}

// stable LSD radix sort of the round's (key, record index) pairs on bits [beginBit, endBit); result in d_adamKeys[1] / d_adamIdx[1]
int sortAdamRecords(ppg_ctx *ctx, size_t n, unsigned int beginBit, unsigned int endBit) {
    hipStream_t s = ctx->stream;
    HIP_CHECK(ctx->d_adamKeys[1].reserve(n)); HIP_CHECK(ctx->d_adamIdx[1].reserve(n)); HIP_CHECK(ctx->d_adamIdx[0].reserve(n));
    if (ctx->adamIota < ctx->d_adamIdx[0].cap) {
        const unsigned int m = (unsigned int)ctx->d_adamIdx[0].cap;
        hipLaunchKernelGGL(k_iota, dim3((m + 255) / 256), dim3(256), 0, s, ctx->d_adamIdx[0].p, m);
        ctx->adamIota = m;
    }
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, ctx->d_adamKeys[0].p, ctx->d_adamKeys[1].p, ctx->d_adamIdx[0].p, ctx->d_adamIdx[1].p, n, beginBit, endBit, s));
    HIP_CHECK(ctx->d_sortTemp.reserve(std::max<size_t>(bytes, 16)));
    HIP_CHECK(rocprim::radix_sort_pairs((void *)ctx->d_sortTemp.p, bytes, ctx->d_adamKeys[0].p, ctx->d_adamKeys[1].p, ctx->d_adamIdx[0].p, ctx->d_adamIdx[1].p, n, beginBit, endBit, s));
    return PPG_OK;
}

// The deferred optimizeBsdfSamplingFraction calls of the round just rendered: sort by key, (multi-GPU: exchange through the round
// hook), apply leaf by leaf (k_adam_apply).  nRecords = positions used in d_adamKeys[0] / d_adamRecs (holes carry key ~0).
int applyAdamRound(ppg_ctx *ctx, size_t nRecords) {
    hipStream_t s = ctx->stream;
    const unsigned int nNodes = (unsigned int)ctx->snodes.size();
    if (nNodes >= (1u << 24)) { ctx->error = "S-tree exceeds 2^24 nodes with a bsdfSamplingFractionLoss"; return PPG_ERR_NOMEM; }
    unsigned int leafBits = 1;
    while ((1u << leafBits) <= nNodes) ++leafBits;  // every leaf index is below the all-ones pattern of a hole
    const unsigned int endBit = PPG_ADAM_LEAF_SHIFT + leafBits;
    if (ctx->sortedCommit && ctx->adamFlagShift != endBit) { ctx->error = "internal: flag bit of the round's records"; return PPG_ERR_STATE; }
    size_t n = nRecords;
    // The splats and the optimiser read the same sorted records and write different things (the building tree; the leaf headers' optimiser
    // state): k_splat_sorted runs on the second stream BESIDE the order / apply kernels — and beside the round hook's exchanges of a
    // sharded render —, whose serial chains leave most of the GPU idle.  Joined before the sorted arrays are reused and before returning.
    struct Join {
        ppg_ctx *c; bool forked = false;
        void now() { if (forked) { (void)hipStreamWaitEvent(c->stream, c->evJoin, 0); forked = false; } }
        ~Join() { now(); }
    } join{ctx};
    // (without a round hook the optimiser's kernels leave the context's stream too: ppg_ctx::treePending)
    const bool aside = ctx->sortedCommit && !ctx->passHook && !ctx->timer.enabled && !ctx->tuneNoOverlap && !ctx->tuneNoAside;
    hipStream_t sa = aside ? ctx->stream3 : s;  // where the order / apply kernels go
    struct AsideGuard {  // an early return with kernels already on stream3: the event joinTree() waits for must lie behind them
        ppg_ctx *c; bool on;
        ~AsideGuard() { if (on && c->treePending) (void)hipEventRecord(c->evAdamDone, c->stream3); }
    } asideGuard{ctx, aside};
    if (n > 0) {
        int rc = PPG_OK;
        // (a round committed through records: one more bit, "a splat only", just above the leaf — those records sort behind the optimiser's)
        timedLaunch(ctx, "adam_sort(rocprim)", n, [&] { rc = sortAdamRecords(ctx, n, ctx->adamFast ? PPG_ADAM_LEAF_SHIFT : 0u, endBit + (ctx->sortedCommit ? 1u : 0u)); });
        if (rc) return rc;
        if (aside) {
            HIP_CHECK(hipEventRecord(ctx->evTreeFork, s));
            HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->evTreeFork, 0));
            HIP_CHECK(hipStreamWaitEvent(ctx->stream3, ctx->evTreeFork, 0));
            const unsigned int chunks = (unsigned int)((n + PPG_SPLAT_CHUNK - 1) / PPG_SPLAT_CHUNK);
            SplatLaunch a{(int)std::max(1u, std::min(chunks, 256u * 8u)), ctx->stream2, ctx->devTree(), ctx->d_adamKeys[1].p, ctx->d_adamIdx[1].p, ctx->d_splat.p, (unsigned int)n, leafBits, ctx->tuneSplatLdsNodes};
            ppg_launch_splat(ctx->directionalFilter, a);
            HIP_CHECK(hipEventRecord(ctx->evSplatDone, ctx->stream2));
            // (from here on work is in flight beside the context's stream: whatever happens below — an early return included — joinTree()
            // must wait for both side streams; the optimiser's event is recorded again behind its kernels)
            HIP_CHECK(hipEventRecord(ctx->evAdamDone, ctx->stream3));
            ctx->treePending = true;
            HIP_CHECK(hipGetLastError());
        } else if (ctx->sortedCommit) {  // DTree::recordIrradiance of every record of the round, D-tree by D-tree (ppg_kernels.h "The commit of a ROUND")
            const unsigned int chunks = (unsigned int)((n + PPG_SPLAT_CHUNK - 1) / PPG_SPLAT_CHUNK);
            SplatLaunch a{(int)std::max(1u, std::min(chunks, 256u * 8u)), s, ctx->devTree(), ctx->d_adamKeys[1].p, ctx->d_adamIdx[1].p, ctx->d_splat.p, (unsigned int)n, leafBits, ctx->tuneSplatLdsNodes};
            if (!ctx->timer.enabled && !ctx->tuneNoOverlap) {
                HIP_CHECK(hipEventRecord(ctx->evFork, s));
                HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
                a.stream = ctx->stream2;
                ppg_launch_splat(ctx->directionalFilter, a);
                HIP_CHECK(hipEventRecord(ctx->evJoin, ctx->stream2));
                join.forked = true;
            } else timedLaunch(ctx, "k_splat_sorted", n, [&] { ppg_launch_splat(ctx->directionalFilter, a); });
            HIP_CHECK(hipGetLastError());
        }
    }
    if (ctx->passHook) {
        // hand the valid records over in key order, compact
        unsigned int nValid = 0;
        if (n > 0) {
            hipLaunchKernelGGL(k_count_valid, dim3(1), dim3(1), 0, s, ctx->d_adamKeys[1].p, (unsigned int)n, ctx->d_adamCount.p,
                               ctx->sortedCommit ? (1ull << ctx->adamFlagShift) : ~0ull);  // (the optimiser's records come first)
            HIP_CHECK(hipMemcpyAsync(&nValid, ctx->d_adamCount.p, 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(ctx->d_adamRecsOut.reserve(std::max<size_t>(1, nValid)));
            if (nValid) hipLaunchKernelGGL(k_gather_records, dim3((nValid + 255) / 256), dim3(256), 0, s, ctx->d_adamRecs.p, ctx->d_adamIdx[1].p, ctx->d_adamRecsOut.p, nValid);
        } else HIP_CHECK(ctx->d_adamRecsOut.reserve(1));
        HIP_CHECK(hipStreamSynchronize(s));
        ctx->inHook = true; ctx->hookReplaced = false; ctx->hookCount = nValid; ctx->hookPhase = 0; ctx->ownerMode = false;
        const int hrc = ctx->passHook(ctx->passHookUser);
        ctx->inHook = false;
        if (hrc != 0) { ctx->error = "round hook failed"; return PPG_ERR_INVALID; }
        if (ctx->hookReplaced) {  // the union over all ranks, in d_adamRecs: sort it by the whole key
            join.now();  // (the sort below overwrites the arrays k_splat_sorted reads)
            n = (size_t)ctx->hookCount;
            if (n > 0) {
                HIP_CHECK(ctx->d_adamKeys[0].reserve(n));
                hipLaunchKernelGGL(k_record_keys, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, s, ctx->d_adamRecs.p, ctx->d_adamKeys[0].p, (unsigned int)n);
                int rc = sortAdamRecords(ctx, n, 0u, endBit);
                if (rc) return rc;
            }
        }
    }
    if (n > 0) {
        const unsigned int nl = (unsigned int)ctx->leaves.size();
        // One wave per D-tree walks its records as a serial chain, so the launch lasts as long as its busiest D-tree — if that one starts
        // last, everything else has finished by then: the D-trees are taken in descending order of their record counts.
        const unsigned int *order = nullptr;
        if (!ctx->tuneAdamUnordered && nl > 1) {
            for (int k = 0; k < 2; ++k) { HIP_CHECK(ctx->d_adamLeafCount[k].reserve(nl)); HIP_CHECK(ctx->d_adamLeafOrder[k].reserve(nl)); }
            size_t bytes = 0;
            HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, ctx->d_adamLeafCount[0].p, ctx->d_adamLeafCount[1].p, ctx->d_adamLeafOrder[0].p, ctx->d_adamLeafOrder[1].p, (size_t)nl, 0u, 32u, sa));
            HIP_CHECK(ctx->d_orderTemp.reserve(std::max<size_t>(bytes, 16)));  // (its own scratch: the next batch's scans use d_sortTemp on the context's stream)
            timedLaunch(ctx, "adam_order", nl, [&] {
                hipLaunchKernelGGL(k_adam_counts, dim3((nl + 255u) / 256u), dim3(256), 0, sa, ctx->d_leaves.p, nl, ctx->d_adamKeys[1].p, (unsigned int)n, ctx->d_adamLeafCount[0].p,
                                   ctx->d_adamLeafOrder[0].p);
                (void)rocprim::radix_sort_pairs_desc((void *)ctx->d_orderTemp.p, bytes, ctx->d_adamLeafCount[0].p, ctx->d_adamLeafCount[1].p, ctx->d_adamLeafOrder[0].p,
                                                     ctx->d_adamLeafOrder[1].p, (size_t)nl, 0u, 32u, sa);
            });
            order = ctx->d_adamLeafOrder[1].p;
        }
        timedLaunch(ctx, "k_adam_apply", n, [&] {
            hipLaunchKernelGGL(k_adam_apply, dim3((nl * 64u + 255u) / 256u), dim3(256), 0, sa, ctx->d_hdr.p, ctx->d_leaves.p, order, nl, ctx->d_adamKeys[1].p, ctx->d_adamIdx[1].p,
                               ctx->d_adamRecs.p, (unsigned int)n, ctx->loss);
        });
        HIP_CHECK(hipGetLastError());
        if (aside) {
            HIP_CHECK(hipEventRecord(ctx->evAdamDone, ctx->stream3));
            ctx->treePending = true;
        }
    }
    if (ctx->passHook && ctx->ownerMode) {  // one owner per D-tree: the owners publish the state they computed
        HIP_CHECK(hipStreamSynchronize(s));
        ctx->inHook = true; ctx->hookPhase = 1;
        const int hrc = ctx->passHook(ctx->passHookUser);
        ctx->inHook = false; ctx->hookPhase = 0; ctx->ownerMode = false;
        if (hrc != 0) { ctx->error = "round hook failed"; return PPG_ERR_INVALID; }
    }
    return PPG_OK;
}

// ------------------------------------------------------------------------------------------------
// Stragglers
// ------------------------------------------------------------------------------------------------
// A guided path survives Russian roulette with probability 0.99 (GP:2124-2139): of the millions of paths of a batch a few hundred are still
// alive after 64 bounces and the longest runs for 300-900 — dependent bounces of ~12 us each, during which one wave works and the GPU waits
// (KITCHEN: 8 % of a 1023-pass render on one GPU, and the term that does not shrink when the image is sharded over eight).  So k_tail runs
// the handed-over paths only up to depth `deferDepth`; the STRAGGLERS — alive at that depth — are written to a compact set (path record by
// k_tail itself, vertex slots by k_extract_vertices), and a second k_tail finishes them on stream4 BESIDE the next batch's kernels, one path
// per wave.  What the batch still owes then — its film kernel (which needs the stragglers' radiance: the batch's Li values are kept in
// d_liKeep and the stragglers' scattered into it), the stragglers' commit — is done by the NEXT batch once its own tail has run
// (drainStragglers), or at the end of the ppg_render_passes call (flushStragglers).  The film sums keep their order and the building tree's
// sums are integers: for a batch without optimiser records nothing changes but the schedule.  In a round of the optimiser the stragglers'
// records are applied one round late — part of the result, the rule is in include/ppg.h ("STRAGGLERS") and in the oracle.
int addGroups(ppg_ctx *ctx, unsigned int first, unsigned int count);

// the compact state of n stragglers over the set's buffers (path records interleaved as k_tail wrote them; vertex slots [slot][path])
int stragglerState(ppg_ctx *ctx, ppg_ctx::Stragglers &G, const PathState &batch, unsigned int n, bool withVertices) {
    const bool filtered = ctx->spatialFilter != SF_NEAREST;
    const unsigned int vs = filtered ? 6u : 4u;
    if (withVertices) HIP_CHECK(G.vert.reserve((size_t)n * (size_t)ctx->maxVertices * vs));
    HIP_CHECK(G.nv8.reserve(n)); HIP_CHECK(G.base.reserve(n));
    if (batch.nee_cos) HIP_CHECK(G.neeCos.reserve(n));
    PathState &Ps = G.P;
    Ps = batch;  // n_pix, pixels: the batch's (adam_path_id)
    Ps.n_paths = n;
    float4 *r = G.rec.p, *v = G.vert.p;
    Ps.ray_o = {r, 8}; Ps.ray_d = {r + 1, 8}; Ps.thr = {r + 2, 8}; Ps.li = {r + 3, 8}; Ps.hit = {r + 4, 8}; Ps.misc = {reinterpret_cast<uint4 *>(r + 5), 8};
    Ps.v_d = {v, vs}; Ps.v_thr = {v + 1, vs}; Ps.v_bsdf = {v + 2, vs}; Ps.v_rad = {v + 3, vs};
    Ps.v_o = {filtered ? v + 4 : nullptr, vs}; Ps.v_vox = {filtered ? v + 5 : nullptr, vs};
    Ps.nee_cos = batch.nee_cos ? G.neeCos.p : nullptr;
    Ps.orig = G.orig.p;
    return PPG_OK;
}

// After the first k_tail of a batch: nStrag paths wait in the current set.  Take their vertex slots and the batch's radiance, start their own
// k_tail (on stream4 when side streams are allowed), and leave the set pending.
int launchStragglers(ppg_ctx *ctx, const PathState &P, const DevScene &S, const DevTree &T, const RenderParams &R, unsigned int nStrag, bool commit, bool adam,
                     bool beside, int tailVariant, size_t ldsBytes) {
    hipStream_t s = ctx->stream;
    ppg_ctx::Stragglers &G = ctx->strag[ctx->stragCur];
    { int rc = stragglerState(ctx, G, P, nStrag, commit); if (rc) return rc; }
    G.n = nStrag; G.commit = commit; G.adam = adam && commit;
    const int small = gridFor(nStrag, 64, 1024);
    if (commit) hipLaunchKernelGGL(k_extract_vertices, dim3(small), dim3(256), 0, s, P, G.P, nStrag, (unsigned int)ctx->maxVertices);
    else if (P.nee_cos) hipLaunchKernelGGL(k_extract_nee_cos, dim3(small), dim3(256), 0, s, P, G.P, nStrag);  // (nothing recorded: a final iteration)
    HIP_CHECK(ctx->d_liKeep.reserve(P.n_paths));
    hipLaunchKernelGGL(k_copy_li, dim3(gridFor(P.n_paths)), dim3(256), 0, s, P, ctx->d_liKeep.p);
    if (G.iotaN < nStrag) {
        HIP_CHECK(G.iota.reserve(nStrag));
        const unsigned int m = (unsigned int)G.iota.cap;
        hipLaunchKernelGGL(k_iota, dim3((m + 255) / 256), dim3(256), 0, s, G.iota.p, m);
        G.iotaN = m;
    }
    HIP_CHECK(hipMemsetAsync(G.ticket.p, 0, 4, s));
    // thinly: as many waves as there are stragglers while the GPU has room for them (a lone path's bounce is 12 us, 64 in one wave 120)
    const unsigned int waves = 1024u * (PPG_BLOCK / 64u);
    const unsigned int laneLimit = std::max(1u, std::min(64u, (nStrag + waves - 1u) / waves));
    const int tailGrid = (int)std::max(1u, std::min(1024u, (nStrag + laneLimit * (PPG_BLOCK / 64u) - 1u) / (laneLimit * (PPG_BLOCK / 64u))));
    DevTree Tt = T;
    if (ctx->loss != LOSS_NONE && ctx->isBuilt) {  // the fractions of THIS round, whatever the optimiser does to the headers meanwhile
        const unsigned int nn = (unsigned int)ctx->snodes.size();
        HIP_CHECK(G.theta.reserve(nn));
        hipLaunchKernelGGL(k_copy_theta, dim3(gridFor(nn, 256, 256)), dim3(256), 0, s, ctx->d_hdr.p, nn, G.theta.p);
        Tt.theta_frozen = G.theta.p;
    }
    hipStream_t st = s;
    if (beside) {
        HIP_CHECK(hipEventRecord(ctx->evStragFork, s));
        HIP_CHECK(hipStreamWaitEvent(ctx->stream4, ctx->evStragFork, 0));
        st = ctx->stream4;
    }
    auto launch = [&] {
        RenderParams Rt = R;
        Rt.defer_depth = 0u;
        TailLaunch a{tailGrid, ldsBytes, st, G.P, S, Tt, Rt, G.iota.p, G.count.p, G.ticket.p, ctx->queues.stats, ctx->ldsTris,
                     ctx->d_tailLongest.p + (ctx->tailLaunches++ % PPG_TAIL_LOG), StragOut{nullptr, nullptr, nullptr}, laneLimit};
        ppg_launch_tail(tailVariant, a);
    };
    if (beside) { launch(); HIP_CHECK(hipEventRecord(ctx->evStragDone, ctx->stream4)); }
    else timedLaunch(ctx, "k_tail(stragglers)", nStrag, launch);  // (its own timer entry: in the product it runs beside the next batch, not on the critical path)
    G.pending = true; G.aside = beside;
    ctx->stragCur ^= 1;
    HIP_CHECK(hipGetLastError());
    return PPG_OK;
}

// What the previous batch still owes: its stragglers have ended (or the context's stream waits until they have) — their radiance into the
// kept copy, the batch's film kernel, their vertices' commit.  In a round of the optimiser their records take the positions from
// recordsFirst on of the round's arrays (sized for them by the caller).
int drainStragglers(ppg_ctx *ctx, size_t recordsFirst) {
    ppg_ctx::Stragglers &G = ctx->stragPrev();
    if (!G.pending) return PPG_OK;
    hipStream_t s = ctx->stream;
    if (G.aside) HIP_CHECK(hipStreamWaitEvent(s, ctx->evStragDone, 0));
    G.pending = false; G.aside = false;
    const int small = gridFor(G.n, 256, 1024);
    hipLaunchKernelGGL(k_scatter_li, dim3(small), dim3(256), 0, s, G.P, G.n, ctx->d_liKeep.p);
    if (G.film) { G.film(); G.film = nullptr; }
    if (G.commit) {
        hipLaunchKernelGGL(k_commit_prepare, dim3((G.n + 255) / 256), dim3(256), 0, s, G.P, (uint4 *)nullptr, G.nv8.p, (const unsigned char *)nullptr);
        DevTree T = ctx->devTree();
        const RenderParams R = ctx->params();
        if (G.adam) {
            if (!T.adam_keys) { ctx->error = "internal: stragglers' records outside a round"; return PPG_ERR_STATE; }
            // Record positions are key order within a D-tree (the round's sort is stable and looks at the leaf bits only): the stragglers take
            // theirs in the order of their paths' ids — k_tail handed them over in whatever order its waves got there.
            HIP_CHECK(G.origSorted.reserve(G.n)); HIP_CHECK(G.perm.reserve(G.n));
            size_t bytes = 0;
            HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, G.orig.p, G.origSorted.p, G.iota.p, G.perm.p, (size_t)G.n, 0u, 32u, s));
            HIP_CHECK(ctx->d_sortTemp.reserve(std::max<size_t>(bytes, 16)));
            HIP_CHECK(rocprim::radix_sort_pairs((void *)ctx->d_sortTemp.p, bytes, G.orig.p, G.origSorted.p, G.iota.p, G.perm.p, (size_t)G.n, 0u, 32u, s));
            hipLaunchKernelGGL(k_ranked_base, dim3(small), dim3(256), 0, s, G.base.p, G.perm.p, G.n, (unsigned int)recordsFirst, (unsigned int)ctx->maxVertices);
            T.adam_base = G.base.p;
        }
        CommitLaunch a{gridFor(G.n, 64, ctx->nBlocks), s, G.P, T, R, ctx->queues, G.nv8.p, nullptr, nullptr, ctx->d_splat.p, ctx->adamFlagShift};
        timedLaunch(ctx, (G.adam && ctx->sortedCommit) ? "k_commit_records" : "k_commit", 0, [&] {
            if (G.adam && ctx->sortedCommit) ppg_launch_commit_records(ctx->spatialFilter, a);
            else ppg_launch_commit(ctx->spatialFilter, ctx->directionalFilter, a);
        });
    }
    HIP_CHECK(hipGetLastError());
    return PPG_OK;
}

// End of a ppg_render_passes call: nothing may stay owed (the variance estimate reads the image, ppg_build_sdtree the building tree).  The
// records of the last round's stragglers are applied in a round of their own (include/ppg.h "STRAGGLERS"); `hookRound`: a sharded render's
// round hook is called for it on every rank, with or without records.
int flushStragglers(ppg_ctx *ctx, bool hookRound) {
    ppg_ctx::Stragglers &G = ctx->stragPrev();
    const bool records = G.pending && G.adam;
    if (!G.pending && !hookRound) return PPG_OK;
    hipStream_t s = ctx->stream;
    ctx->joinTree();  // (the last round's splats and optimiser steps read the record arrays this round is about to overwrite)
    size_t n = 0;
    if (records || hookRound) {
        ctx->adamActive = true; ctx->adamFast = true;
        unsigned int leafBits = 1;
        while ((1u << leafBits) <= (unsigned int)ctx->snodes.size()) ++leafBits;
        ctx->adamFlagShift = PPG_ADAM_LEAF_SHIFT + leafBits;
        // (a few thousand vertices spread over all D-trees: k_commit's global atomics — k_splat_sorted would stage a D-tree in LDS for every
        // record or two, 5 ms for the 50 000 record positions of KITCHEN's 2-pass rounds; integer sums, the same bits)
        ctx->sortedCommit = false;
        n = records ? (size_t)G.n * (size_t)ctx->maxVertices : 0;
        HIP_CHECK(ctx->d_adamKeys[0].reserve(std::max<size_t>(1, n))); HIP_CHECK(ctx->d_adamRecs.reserve(std::max<size_t>(1, n)));
        if (n) HIP_CHECK(hipMemsetAsync(ctx->d_adamKeys[0].p, 0xff, n * 8, s));
    }
    int rc = drainStragglers(ctx, 0);
    if (!rc && (records || hookRound)) rc = applyAdamRound(ctx, n);
    ctx->adamActive = false;
    return rc;
}

// `batch` BlockedRenderProcesses (GP:1087-1106 / renderBlock GP:1587-1641) over all owned pixels in one set of launches.
// adamRound: this batch is one round of the sampling-fraction optimiser.
// A launch of whole groups of a final iteration's passes (include/ppg.h "Final iteration: groups of passes"): `batch` passes = groups of
// groupPasses passes (the last one may be shorter; or a part of ONE group when groupPasses >= batch), the first starting at pass
// firstPass of the render, consecutive groups of the launch stridePasses apart; group k accumulates into slot slot0 + k * slotStride.
int renderBatch(ppg_ctx *ctx, int batch, bool adamRound, const GroupLaunch *gl = nullptr, bool last = true, const unsigned int *regionPixels = nullptr,
                unsigned int regionCount = 0) {
    PathState P = ctx->paths;
    if (regionPixels) { P.n_pix = regionCount; P.pixels = regionPixels; }  // a round over ONE group of blocks (include/ppg.h "Rounds by image region")
    if (gl && gl->wholeFilm && ctx->shardWorld > 1) { P.n_pix = ctx->nPixAll; P.pixels = ctx->d_pixelsAll.p; }  // the whole film, not this rank's tiles
    P.n_paths = (unsigned int)((size_t)P.n_pix * ctx->sppPerPass * (size_t)batch);
    if (P.n_paths == 0 && !(adamRound && ctx->passHook)) return PPG_OK;
    hipStream_t s = ctx->stream;
    // Adam records: positions are known in advance unless one vertex can make several records (box spatial filter) or records are made
    // while the paths are still being traced (the direct-light vertex of nee = kickstart)
    ctx->adamActive = adamRound;
    ctx->adamFast = adamRound && ctx->spatialFilter != SF_BOX && !(ctx->doNee && ctx->nee == NEE_KICKSTART);
    if (adamRound && !ctx->adamFast) {
        const size_t cap = std::max<size_t>((size_t)1 << 16, std::min<size_t>((size_t)48 * P.n_paths, 0xfffffff0u));
        HIP_CHECK(ctx->d_adamKeys[0].reserve(cap)); HIP_CHECK(ctx->d_adamRecs.reserve(cap));
        HIP_CHECK(hipMemsetAsync(ctx->d_adamCount.p, 0, 8, s));
    }
    DevScene S = ctx->scene;
    DevTree T = ctx->devTree();
    RenderParams R = ctx->params();
    R.spp = ctx->sppPerPass * batch;  // sample j of the batch has sample index pass_index * sppPerPass + j, j < spp * batch
    R.pass_index_spp = (unsigned int)ctx->passesRendered * (unsigned int)ctx->sppPerPass;
    if (gl) {
        R.pass_index_spp = gl->firstPass * (unsigned int)ctx->sppPerPass;
        R.group_samples = gl->groupPasses * (unsigned int)ctx->sppPerPass;
        R.group_stride = gl->stridePasses * (unsigned int)ctx->sppPerPass;
    }
    Queues Q = ctx->queues;
    // Workgroups of this batch's wavefront kernels.  A batch of a few million paths runs faster on fewer, larger queue slices (ray replacement
    // and the slice sort have more to work with, fewer half-empty workgroups per launch: +1 % on the driver's command with 2048), the 59 M-path
    // launches of a long render on more (4096: +1.6 % at 1023 passes) — so the grid follows the batch.  Slices are sized for the largest batch on
    // nBlocks workgroups: a smaller grid is taken only if this batch's slices still fit.
    int grid = ctx->nBlocks;
    if (ctx->tuneBlocksSmall > 0 && ctx->tuneBlocksSmall < ctx->nBlocks && (size_t)P.n_paths <= ctx->tuneSmallPaths) {
        const size_t chunks = ((size_t)P.n_paths + PPG_DCHUNK - 1) / PPG_DCHUNK;
        if (((chunks + (size_t)ctx->tuneBlocksSmall - 1) / (size_t)ctx->tuneBlocksSmall) * PPG_DCHUNK <= (size_t)ctx->queues.cap) grid = ctx->tuneBlocksSmall;
    }
    const int gridAll = gridFor(P.n_paths);
    const bool smallScene = ctx->scene.n_tris <= 64 && ctx->ldsTris == ctx->scene.n_tris && ctx->scene.n_spheres == 0 && !ctx->tuneForceBvh;
    // Tracing inside k_generate / k_shade (no k_trace launch, no ray/hit round trip) was measured SLOWER on MI355X
    // (cbox-720p, 63 passes: 184 ms vs 172 ms for generate+trace+shade): the fused kernel needs 142 VGPRs (3 waves/SIMD)
    // and only the surviving lanes trace.  Kept selectable for re-measurement on other scenes.
    // (not in a round whose stragglers' records are deferred: which paths those are is decided where k_tail takes them, include/ppg.h "STRAGGLERS")
    const bool fused = smallScene && ctx->tuneFuse && !(adamRound && ctx->adamFast && ctx->maxDepth < 0);
    const bool neeOn = ctx->doNee;  // m_doNee of this iteration (doNeeWithSpp, GP:1331-1340)
    const bool fullMats = ctx->fullMaterials;  // any BSDF beyond diffuse / two-sided diffuse / mirror: the FULL kernel variants
    const size_t triBytes = (size_t)ctx->scene.n_tris * 48;
    const bool unbounded = ctx->maxDepth < 0;
    const size_t ldsBytes = smallScene ? (size_t)ctx->ldsTris * 48 : (size_t)PPG_LDS_STACK * PPG_BLOCK * 4;
    const size_t traceLds = smallScene ? ldsBytes : (size_t)PPG_TRACE_STACK * PPG_BLOCK * 4 + (PPG_TRACE_PAIRS ? (PPG_BLOCK / 64) * sizeof(PairLds) : 0);  // k_trace: + a wave's pair scratch
    // live paths below which the wavefront stops (unbounded paths: k_tail takes over; bounded paths: nothing is left)
    const unsigned int stopBelow = unbounded ? (ctx->tailThreshold ? ctx->tailThreshold : std::max(ctx->tailMin, P.n_paths / ctx->tailDiv)) : 1u;
    unsigned int hostCount = P.n_paths;
    int bouncesRun = 0;
    if (P.n_paths > 0) {
        int qin = QIN_FIRST;
        timedLaunch(ctx, "k_generate", P.n_paths, [&] {
            if (fused) hipLaunchKernelGGL(k_generate<true>, dim3(gridAll), dim3(PPG_BLOCK), triBytes, s, P, S, R, Q);
            else hipLaunchKernelGGL(k_generate<false>, dim3(gridAll), dim3(PPG_BLOCK), 0, s, P, S, R, Q);
        });
        // Bounce 1 works on all paths of the batch; every later bounce on the dense list of live paths that k_scan_counts + k_gather_slices
        // rebuild from k_shade's output slices (ppg_kernels.h "Queues").  Bounded paths (maxDepth > 0) run maxDepth bounces.  Unbounded
        // paths run wavefront bounces until fewer than `stopBelow` paths are alive and hand those to the persistent-thread tail (k_tail).
        // Nothing is read back in between: the host launches as many bounces as the previous batch's survival curve says this batch needs
        // (plus a margin); the device raises a stop flag when the live count has fallen below the threshold, and the kernels of the
        // bounces launched beyond that point return at once.  The host synchronises once per batch.
        int maxBounces = ctx->maxDepth;
        if (unbounded) {
            if (fused) maxBounces = 1 << 20;
            else if (ctx->tuneBulkBounces >= 0) maxBounces = std::min(64, ctx->tuneBulkBounces);
            else {
                // live(b) of this batch ~ (live(b) / paths of the previous batch) * paths of this one; beyond what was observed, the last ratio
                int need = 8;
                if (ctx->prevBounces > 0 && ctx->prevPaths > 0) {
                    const double scale = (double)P.n_paths / (double)ctx->prevPaths;
                    need = -1;
                    for (int b = 0; b < ctx->prevBounces; ++b) if (ctx->prevLive[b] * scale < stopBelow) { need = b + 1; break; }
                    if (need < 0) {
                        double live = ctx->prevLive[ctx->prevBounces - 1] * scale;
                        const double before = ctx->prevBounces > 1 ? ctx->prevLive[ctx->prevBounces - 2] : (double)ctx->prevPaths;
                        const double ratio = std::min(0.97, std::max(0.5, before > 0 ? ctx->prevLive[ctx->prevBounces - 1] / before : 0.9));
                        need = ctx->prevBounces;
                        while (live >= stopBelow && need < 64) { live *= ratio; ++need; }
                    }
                }
                maxBounces = std::max(1, std::min(64, need + ctx->bounceMargin));
                // (a batch smaller than the hand-over threshold stops after its first bounce whatever happens: no margin — every launch beyond the
                // stop returns at once, but three kernels and two small ones per bounce are still 30 us of launches, 0.5 ms in a first batch)
                if (P.n_paths < stopBelow) maxBounces = 1;
            }
            // (a round whose stragglers' records are deferred: a path alive after wavefront bounce b has depth b, and which paths are
            // stragglers is decided where k_tail takes them — never beyond the depth of the rule, include/ppg.h "STRAGGLERS")
            if (adamRound && ctx->adamFast && !fused) maxBounces = std::min(maxBounces, std::max(1, ctx->deferDepthAdam));
        }
        const bool liveCount = ctx->timer.enabled || (unbounded && fused);  // kernel timing wants the units of every launch
        unsigned int *counts = ctx->d_bounceCounts.p;  // [0..63] live paths after bounce b, [64] the stop flag
        HIP_CHECK(hipMemsetAsync(counts, 0, 72 * 4, s));
        Q.stop = counts + 64;
        Q.dense_n = ctx->d_total.p;
        for (int b = 0; b < maxBounces; ++b) {
            // (the first bounce — every path of the batch, camera rays — is sorted only for the sake of the split into material classes)
            const bool sortSlices = fullMats && !fused && (qin == QIN_DENSE || (Q.n_common && !neeOn && !ctx->tuneNoSortFirst)) && ctx->d_queueSorted.p;
            // k_trace sorts its own slices when it has traced them (sort_slice, ppg_kernels.h); PPG_SORT_KERNEL=1: k_sort_slices as a launch of its own
            const bool sortInTrace = sortSlices && !ctx->tuneSortKernel;
            unsigned int *const trSorted = sortInTrace ? ctx->d_queueSorted.p : nullptr;
            unsigned char *const trKeys = sortInTrace ? ctx->d_sortKeys.p : nullptr;
            if (!fused)
                timedLaunch(ctx, "k_trace", hostCount, [&] {
                    if (smallScene) hipLaunchKernelGGL((k_trace<true, false>), dim3(grid), dim3(PPG_BLOCK), traceLds, s, P, S, Q, qin, 0, ctx->ldsTris, trSorted, trKeys);
                    else if (ctx->timer.enabled) hipLaunchKernelGGL((k_trace<false, true>), dim3(grid), dim3(PPG_BLOCK), traceLds, s, P, S, Q, qin, ctx->ldsNodes, ctx->ldsTris, trSorted, trKeys);
                    else hipLaunchKernelGGL((k_trace<false, false>), dim3(grid), dim3(PPG_BLOCK), traceLds, s, P, S, Q, qin, ctx->ldsNodes, ctx->ldsTris, trSorted, trKeys);
                });
            int shadeIn = qin;
            if (sortSlices) {
                if (!sortInTrace)
                    timedLaunch(ctx, "k_sort_slices", hostCount, [&] {
                        hipLaunchKernelGGL(k_sort_slices, dim3(grid), dim3(PPG_BLOCK), 0, s, P, S, Q, qin, ctx->d_queueSorted.p, ctx->d_sortKeys.p);
                    });
                shadeIn = QIN_SORTED;
            }
            ctx->joinTree();  // (k_generate, the first k_trace and k_sort_slices ran beside the previous round's optimiser: k_shade needs its result)
            // FULL scene, sorted slice, no luminaire sampling: the common material classes first, in their own leaner kernel (MSET_COMMON)
            const bool split = shadeIn == QIN_SORTED && Q.n_common && !neeOn;
            if (split) {
                timedLaunch(ctx, "k_shade<common>", hostCount, [&] {
                    ShadeLaunch a{grid, 0, s, P, S, T, R, Q, QIN_SORTED_COMMON, smallScene ? 1 : 0, ctx->d_queueSorted.p};
                    ppg_launch_shade_common(a);
                });
                shadeIn = QIN_SORTED_REST;
            }
            timedLaunch(ctx, fused ? "k_shade<fused>" : (neeOn ? "k_shade<nee>" : (fullMats ? (split ? "k_shade<rest>" : "k_shade<full>") : "k_shade")), hostCount, [&] {
                const int small = smallScene ? 1 : 0;
                // dynamic LDS: the staged triangles (fused, or luminaire sampling on a small scene) or the shadow rays' BVH stack columns
                const size_t neeBytes = smallScene ? triBytes : (size_t)PPG_LDS_STACK * PPG_BLOCK * 4;
                const size_t lds = fused ? triBytes : ((neeOn || ctx->scene.has_null) ? neeBytes : 0);
                const int variant = (fused ? 4 : 0) | (neeOn ? 2 : 0) | (fullMats ? 1 : 0);
                ShadeLaunch a{grid, lds, s, P, S, T, R, Q, shadeIn, small, ctx->d_queueSorted.p};
                ppg_launch_shade(variant, a);
            });
            qin = QIN_DENSE;
            ++bouncesRun;
            hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, Q.count[0], ctx->d_offsets.p, (unsigned int)grid, ctx->d_total.p,
                               counts + std::min(b, 63), counts + 64, stopBelow);
            hipLaunchKernelGGL(k_gather_slices, dim3(grid), dim3(PPG_BLOCK), 0, s, Q.items[0], Q.count[0], ctx->d_offsets.p, Q.cap, Q.items[1]);
            if (liveCount) {
                HIP_CHECK(hipMemcpyAsync(&hostCount, ctx->d_total.p, 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                if (hostCount == 0 || (unbounded && !fused && hostCount < stopBelow)) break;
            }
        }
    }
    // What follows the wavefront bounces of a batch:
    //   unbounded paths   persistent threads finish the survivors (k_tail): ONE launch, every lane carries a path from the dense list until it
    //                     ends.  Its length is set by the batch's LONGEST path (KITCHEN: 300-900 dependent bounces among 1-12 M paths); the
    //                     crowd of short paths dies away around it and the wave that holds it speeds up as it empties;
    //   training batches  k_commit splats every recorded vertex into the building tree.
    // With both, k_commit for the paths that HAVE ended runs on a second stream beside k_tail, and a second, small k_commit takes the
    // stragglers afterwards.  Integer accumulation makes the split invisible in the result.  In a round of the optimiser, the record
    // positions must then be known before the tail has run: a straggler reserves max_vertices positions (unused ones stay holes).
    ctx->joinTree();  // (a batch without a wavefront bounce)
    const bool tail = unbounded && !fused && P.n_paths > 0;
    const bool commit = !ctx->isFinalIter && P.n_paths > 0;
    const bool fastRound = adamRound && ctx->adamFast && P.n_paths > 0;
    // STRAGGLERS.  k_tail hands the paths still alive at depth `deferDepth` over to a second launch that runs beside the NEXT batch (see
    // "Stragglers" above drainStragglers).  In a round of the optimiser whose record positions are known the depth is part of the result
    // (include/ppg.h "STRAGGLERS": their records are applied one round late); elsewhere it is scheduling only, and worth it only when another
    // batch follows at once (`last` = false).
    const bool deferRecords = adamRound && ctx->adamFast && unbounded;
    const unsigned int deferDepth = !tail ? 0u : (deferRecords ? (unsigned int)ctx->deferDepthAdam : ((!last && !adamRound && !ctx->timer.enabled && !ctx->tuneNoOverlap) ? (unsigned int)ctx->tuneSplitDepth : 0u));
    const bool splitTail = deferDepth > 0;
    const bool beside = !ctx->timer.enabled && !ctx->tuneNoOverlap;  // side streams may be used
    // k_commit of the paths that HAVE ended when k_tail takes over runs beside it; with a split tail the two-phase commit is needed anyway
    // (the hand-over must be known before the tail runs)
    const bool overlap = tail && commit && (beside || splitTail);
    // A round whose record positions are known commits in three steps — records, the optimiser's sort, splats D-tree by D-tree
    // (ppg_kernels.h "The commit of a ROUND") — instead of one lane per vertex adding to the pool with global atomics.
    {
        unsigned int leafBits = 1;
        while ((1u << leafBits) <= (unsigned int)ctx->snodes.size()) ++leafBits;
        ctx->adamFlagShift = PPG_ADAM_LEAF_SHIFT + leafBits;
    }
    // (the flag needs a key bit above the leaf: an S-tree of 2^23 nodes or more — never seen — commits with k_commit)
    ctx->sortedCommit = adamRound && ctx->adamFast && !ctx->tuneNoSortedCommit && ctx->adamFlagShift < 63u;
    size_t nRecords = 0;
    unsigned int *dense = Q.items[1];  // the live paths in one list (k_tail's work list)
    ppg_ctx::Stragglers &G = ctx->strag[ctx->stragCur];  // the set this batch's stragglers go to (the other one may be pending: the previous batch's)
    // the records of the PREVIOUS round's stragglers are applied with this round's: they take the positions behind its own
    const size_t nDeferredIn = (adamRound && ctx->adamFast && ctx->stragPrev().pending && ctx->stragPrev().adam) ? (size_t)ctx->stragPrev().n * (size_t)ctx->maxVertices : 0;
    // (handedPaths: how many paths the launch will find in the list, when the host knows — 0 = unknown)
    auto launchTail = [&](unsigned int depth, size_t handedPaths = 0) {
        HIP_CHECK(hipMemsetAsync(ctx->d_ticket.p, 0, 4, s));
        // (the persistent workgroups of k_tail hold their registers until their last path has ended: no more of them than fit the GPU at once)
        const int tailGrid = std::min(grid, ctx->tuneTailBlocks ? ctx->tuneTailBlocks : 1024);
        // Fewer paths than lanes — a rank's share of a small round, a round over one group of blocks —: dealt THINLY, the same number of
        // lanes in every wave, instead of 64 to the first waves and none to the rest: a wave's bounce costs the union of its lanes' branches
        // and its longest traversal, and idle SIMDs cost nothing.
        const size_t tailWaves = (size_t)tailGrid * (PPG_BLOCK / 64);
        const unsigned int laneLimit = handedPaths ? (unsigned int)std::max<size_t>(1, std::min<size_t>(64, (handedPaths + tailWaves - 1) / tailWaves)) : 64u;
        timedLaunch(ctx, "k_tail", hostCount, [&] {
            RenderParams Rt = R;
            Rt.defer_depth = depth;
            // (the launch's longest path is logged for launches that run their paths to the END: the sum is the tails' critical path; a launch that
            // hands its stragglers over writes to a spare slot nobody reads)
            TailLaunch a{tailGrid, ldsBytes, s, P, S, T, Rt, dense, ctx->d_total.p, ctx->d_ticket.p, Q.stats, ctx->ldsTris,
                         ctx->d_tailLongest.p + (depth ? PPG_TAIL_LOG : (ctx->tailLaunches++ % PPG_TAIL_LOG)), StragOut{G.rec.p, G.orig.p, G.count.p}, laneLimit};
            ppg_launch_tail((smallScene ? 4 : 0) | (neeOn ? 2 : 0) | (fullMats ? 1 : 0), a);
        });
        return PPG_OK;
    };
    // interleaved path records: k_commit and k_path_nv sweep the per-path word once per (slot, path) item — from a contiguous copy
    PathState Pc = P;
    bool miscCopied = false;
    auto copyMisc = [&] {  // k_commit_prepare: a byte per path for k_commit's work items (+ the contiguous copy of the path words)
        if (miscCopied || P.n_paths == 0) return PPG_OK;
        HIP_CHECK(ctx->d_nv8.reserve(P.n_paths));
        hipLaunchKernelGGL(k_commit_prepare, dim3((P.n_paths + 255) / 256), dim3(256), 0, s, P, ctx->aosPaths ? ctx->d_miscCompact.p : (uint4 *)nullptr, ctx->d_nv8.p,
                           overlap ? (const unsigned char *)ctx->d_straggler.p : (const unsigned char *)nullptr);
        if (ctx->aosPaths) Pc.misc = {ctx->d_miscCompact.p, 1};
        miscCopied = true;
        return PPG_OK;
    };
    // mode 0: every path; 1: paths not flagged as stragglers; 2: the paths of the dense list
    auto launchCommit = [&](hipStream_t st, int mode) {
        const unsigned char *nv8 = ctx->d_nv8.p;  // (mode 1: the stragglers' bytes are 0; mode 2: by list position, written just before)
        if (mode == 2) hipLaunchKernelGGL(k_commit_prepare_list, dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, P, dense, ctx->d_total.p, ctx->d_nv8.p);
        const unsigned int *list = mode == 2 ? dense : nullptr;
        const unsigned long long *listN = mode == 2 ? ctx->d_total.p : nullptr;
        const PathState &PP = mode == 2 ? P : Pc;  // the stragglers' words changed in the tail: read them in place
        CommitLaunch a{grid, st, PP, T, R, Q, nv8, list, listN, ctx->d_splat.p, ctx->adamFlagShift};
        if (ctx->sortedCommit) ppg_launch_commit_records(ctx->spatialFilter, a);
        else ppg_launch_commit(ctx->spatialFilter, ctx->directionalFilter, a);
    };
    if (tail) {
        if (bouncesRun == 0)  // no wavefront bounce was run: every path of the batch goes to the persistent threads
            hipLaunchKernelGGL(k_iota_total, dim3(gridAll), dim3(PPG_BLOCK), 0, s, dense, P.n_paths, ctx->d_total.p);
        if (overlap) {
            HIP_CHECK(ctx->d_straggler.reserve(P.n_paths));
            HIP_CHECK(hipMemsetAsync(ctx->d_straggler.p, 0, P.n_paths, s));
            hipLaunchKernelGGL(k_mark_list, dim3(grid), dim3(PPG_BLOCK), 0, s, dense, ctx->d_total.p, ctx->d_straggler.p);
        } else if (!splitTail) {
            int rc = launchTail(0u);
            if (rc) return rc;
        }
    }
    if (fastRound) {
        // position of path i's records = exclusive scan of the vertex counts
        HIP_CHECK(ctx->d_adamNv.reserve(P.n_paths)); HIP_CHECK(ctx->d_adamBase.reserve(P.n_paths));
        { int rc = copyMisc(); if (rc) return rc; }
        hipLaunchKernelGGL(k_path_nv, dim3((P.n_paths + 255) / 256), dim3(256), 0, s, Pc, ctx->d_adamNv.p, overlap ? ctx->d_straggler.p : nullptr, (unsigned int)ctx->maxVertices);
        size_t bytes = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, ctx->d_adamNv.p, ctx->d_adamBase.p, 0u, (size_t)P.n_paths, rocprim::plus<unsigned int>(), s));
        HIP_CHECK(ctx->d_sortTemp.reserve(std::max<size_t>(bytes, 16)));
        HIP_CHECK(rocprim::exclusive_scan((void *)ctx->d_sortTemp.p, bytes, ctx->d_adamNv.p, ctx->d_adamBase.p, 0u, (size_t)P.n_paths, rocprim::plus<unsigned int>(), s));
        HIP_CHECK(hipMemcpyAsync(ctx->h_round + 65, ctx->d_adamBase.p + (P.n_paths - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(ctx->h_round + 66, ctx->d_adamNv.p + (P.n_paths - 1), 4, hipMemcpyDeviceToHost, s));
    }
    size_t nRecordsOwn = 0;
    if (tail || fastRound) {
        if (unbounded) HIP_CHECK(hipMemcpyAsync(ctx->h_round, ctx->d_bounceCounts.p, 64 * 4, hipMemcpyDeviceToHost, s));
        if (splitTail) HIP_CHECK(hipMemcpyAsync(ctx->h_round + 68, ctx->d_total.p, 8, hipMemcpyDeviceToHost, s));  // the paths handed to k_tail
        HIP_CHECK(hipStreamSynchronize(s));  // the one host round trip of a batch of unbounded paths / of a round
        if (tail) {
            // the survival curve of this batch sizes the schedule of the next one
            int ran = bouncesRun;
            for (int b = 0; b < bouncesRun; ++b) if (ctx->h_round[b] < stopBelow) { ran = b + 1; break; }
            ctx->prevBounces = ran; ctx->prevPaths = P.n_paths;
            for (int b = 0; b < ran; ++b) ctx->prevLive[b] = ctx->h_round[b];
            if (ctx->debugBatch) {
                fprintf(stderr, "[ppg batch] iter %d paths %u built %d final %d launched %d ran %d stop<%u live:", ctx->iter, P.n_paths, (int)ctx->isBuilt, (int)ctx->isFinalIter, bouncesRun, ran, stopBelow);
                for (int b = 0; b < ran; ++b) fprintf(stderr, " %u", ctx->h_round[b]);
                fprintf(stderr, "\n");
            }
        }
        if (fastRound) {
            nRecordsOwn = (size_t)ctx->h_round[65] + ctx->h_round[66];
            nRecords = nRecordsOwn + nDeferredIn;
            if (nRecords > 0xfffffff0ull) { ctx->error = "too many Adam records in one round"; return PPG_ERR_NOMEM; }
            HIP_CHECK(ctx->d_adamKeys[0].reserve(std::max<size_t>(1, nRecords))); HIP_CHECK(ctx->d_adamRecs.reserve(std::max<size_t>(1, nRecords)));
            if (ctx->sortedCommit) HIP_CHECK(ctx->d_splat.reserve(std::max<size_t>(1, nRecords)));
            if (nRecords) HIP_CHECK(hipMemsetAsync(ctx->d_adamKeys[0].p, 0xff, nRecords * 8, s));
            T = ctx->devTree();
        }
    }
    if (adamRound && ctx->adamFast && !fastRound && nDeferredIn) {  // an empty round (a cancelled rank kept in step) that still owes the previous round's stragglers
        nRecords = nDeferredIn;
        HIP_CHECK(ctx->d_adamKeys[0].reserve(nRecords)); HIP_CHECK(ctx->d_adamRecs.reserve(nRecords));
        if (ctx->sortedCommit) HIP_CHECK(ctx->d_splat.reserve(nRecords));
        HIP_CHECK(hipMemsetAsync(ctx->d_adamKeys[0].p, 0xff, nRecords * 8, s));
        T = ctx->devTree();
    }
    if (commit) { int rc = copyMisc(); if (rc) return rc; }
    unsigned int nStrag = 0;
    if (splitTail) {
        // k_tail up to depth deferDepth; what is alive then is written to the straggler set (at most every path handed over)
        const size_t handed = (size_t)(((unsigned long long)ctx->h_round[69] << 32) | ctx->h_round[68]);
        HIP_CHECK(G.rec.reserve(std::max<size_t>(1, handed) * 8)); HIP_CHECK(G.orig.reserve(std::max<size_t>(1, handed)));
        HIP_CHECK(G.count.reserve(2)); HIP_CHECK(G.ticket.reserve(1));
        HIP_CHECK(hipMemsetAsync(G.count.p, 0, 16, s));
        if (overlap) HIP_CHECK(hipEventRecord(ctx->evFork, s));
        { int rc = launchTail(deferDepth, handed); if (rc) return rc; }
        if (overlap) {
            if (beside) {
                HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
                launchCommit(ctx->stream2, 1);
                HIP_CHECK(hipEventRecord(ctx->evJoin, ctx->stream2));
                HIP_CHECK(hipStreamWaitEvent(s, ctx->evJoin, 0));
            } else timedLaunch(ctx, ctx->sortedCommit ? "k_commit_records" : "k_commit", P.n_paths, [&] { launchCommit(s, 1); });
        }
        HIP_CHECK(hipMemcpyAsync(ctx->h_round + 70, G.count.p, 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));  // the second host round trip of a batch with a split tail: how many stragglers
        nStrag = ctx->h_round[70];
        if (ctx->debugBatch) fprintf(stderr, "[ppg batch] iter %d stragglers %u of %zu handed over at depth %u\n", ctx->iter, nStrag, handed, deferDepth);
        // (the previous batch's set is consumed before this batch's stragglers start: one event, one kept copy of the radiance)
        if (nStrag) { int rc = drainStragglers(ctx, nRecordsOwn); if (rc) return rc; }
        if (nStrag) {
            int rc = launchStragglers(ctx, P, S, T, R, nStrag, commit, deferRecords, beside, (smallScene ? 4 : 0) | (neeOn ? 2 : 0) | (fullMats ? 1 : 0), ldsBytes);
            if (rc) return rc;
        }
        if (overlap) {
            if (beside) launchCommit(s, 2);
            else timedLaunch(ctx, ctx->sortedCommit ? "k_commit_records" : "k_commit", 0, [&] { launchCommit(s, 2); });
        }
    } else if (overlap) {
        HIP_CHECK(hipEventRecord(ctx->evFork, s));
        int rc = launchTail(0u);
        if (rc) return rc;
        HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
        launchCommit(ctx->stream2, 1);
        HIP_CHECK(hipEventRecord(ctx->evJoin, ctx->stream2));
        HIP_CHECK(hipStreamWaitEvent(s, ctx->evJoin, 0));
        launchCommit(s, 2);
    } else if (commit) {
        timedLaunch(ctx, ctx->sortedCommit ? "k_commit_records" : "k_commit", P.n_paths, [&] { launchCommit(s, 0); });
    }
    // what an earlier batch still owes (its film kernel, its stragglers' commit) comes before this batch's own sort and film kernel
    if (!(splitTail && nStrag)) { int rc = drainStragglers(ctx, nRecordsOwn); if (rc) return rc; }  // (a batch with stragglers of its own consumed it above)
    if (adamRound) {
        if (!ctx->adamFast) {
            HIP_CHECK(hipMemcpyAsync(ctx->h_round + 64, ctx->d_adamCount.p, 8, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (ctx->h_round[65]) { ctx->error = "Adam record buffer overflow (box spatial filter / next-event estimation with a bsdfSamplingFractionLoss)"; return PPG_ERR_NOMEM; }
            nRecords = ctx->h_round[64];
        }
        int rc = applyAdamRound(ctx, nRecords);
        ctx->adamActive = false;
        if (rc) return rc;
    }
    if (P.n_paths > 0) {
        // the batch's film kernel — at once, or, when stragglers of this batch are still running, from the copy of its paths' radiance once
        // they have ended (drainStragglers): the sums a pixel's samples go through are the same either way
        PathState Pf = P;
        const bool defer = splitTail && nStrag > 0;
        if (defer) Pf.li = {ctx->d_liKeep.p, 1};
        const int sppBatch = ctx->sppPerPass * batch;
        const bool hasGl = gl != nullptr;
        const GroupLaunch glv = gl ? *gl : GroupLaunch{};
        auto film = [ctx, Pf, sppBatch, hasGl, glv] {
            hipStream_t st = ctx->stream;
            timedLaunch(ctx, "k_film", Pf.n_pix, [&] {
                if (hasGl) hipLaunchKernelGGL(k_film_groups, dim3((Pf.n_pix + 255) / 256), dim3(256), 0, st, Pf, sppBatch, glv.groupPasses * (unsigned int)ctx->sppPerPass,
                                              ctx->d_partials.p + (ctx->shardWorld > 1 ? 4 * (size_t)ctx->nPixAll : 0), glv.slot0, glv.slotStride, ctx->nPixAll);
                else hipLaunchKernelGGL(k_film, dim3((Pf.n_pix + 255) / 256), dim3(256), 0, st, Pf, sppBatch, ctx->d_image.p, ctx->d_sq.p, ctx->d_imageW.p,
                                        ctx->d_film.p, ctx->d_filmW.p);
            });
            if (hasGl && glv.addCount) (void)addGroups(ctx, 0, glv.addCount);
        };
        if (defer) ctx->stragPrev().film = film; else film();  // (launchStragglers made this batch's set the previous one)
    }
    HIP_CHECK(hipGetLastError());
    return PPG_OK;
}

// The passes of a FINAL iteration (spp budget), include/ppg.h "Final iteration: groups of passes": groups of ppg_final_group_passes(numPasses)
// passes, each summed on its own (k_film_groups) and added to image / film in group order (k_add_groups).  One GPU renders them all, several
// groups per launch; rank r of a sharded render renders groups r, r + world, ... over the WHOLE film — nothing is recorded in a final
// iteration (GP:2150-2154) and the sampler is keyed by (pixel, sample index), so the groups are independent, and a rank then pays 1 / world of
// the iteration's tails instead of all of them on its tiles — and leaves its slots for the exchange (ppg_final_partials).
int addGroups(ppg_ctx *ctx, unsigned int first, unsigned int count) {
    const unsigned int n = ctx->nPixAll;
    hipLaunchKernelGGL(k_add_groups, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->d_partials.p + (ctx->shardWorld > 1 ? 4 * (size_t)n : 0), first, count,
                       ctx->d_image.p, ctx->d_sq.p, ctx->d_imageW.p, ctx->d_film.p, ctx->d_filmW.p);
    return PPG_OK;
}
// How the groups of a sharded final iteration are dealt (include/ppg.h "Final iteration: groups of passes"): whole groups to the ranks when
// there are at least two per rank, otherwise every group to every rank ON ITS TILES — a 13-pass final iteration is ONE group, which one rank
// would render alone.  The sums a pixel's samples go through are the same either way (a group's partial per pixel, the partials in group
// order), so the picture does not depend on the choice.
static bool finalGroupsByRank(unsigned int nGroups, unsigned int world) { return world > 1 && nGroups >= 2u * world; }

int renderFinalGroups(ppg_ctx *ctx, int numPasses) {
    const unsigned int G = (unsigned int)ppg_final_group_passes(numPasses), nGroups = ((unsigned int)numPasses + G - 1) / G;
    const unsigned int world = (unsigned int)ctx->shardWorld, rank = (unsigned int)ctx->shardRank;
    const bool byRank = finalGroupsByRank(nGroups, world);  // else: all groups, this rank's tiles (world == 1: the whole film)
    const size_t n = ctx->nPixAll;
    std::vector<unsigned int> mine;
    for (unsigned int g = byRank ? rank : 0u; g < nGroups; g += byRank ? world : 1u) mine.push_back(g);
    auto passesOf = [&](unsigned int g) { return std::min(G, (unsigned int)numPasses - g * G); };
    // Passes per launch: maxBatchFinal.  PPG_FINAL_HALVES=1: when this rank's share of the iteration would fit ONE launch, half of it, so that the
    // stragglers of the first half finish beside the second half's paths ("Stragglers") — measured and NOT the default: each half hands its own
    // 2 M paths to k_tail (the floor below which wavefront bounces no longer fill the GPU), and that crowd phase, 10-12 ms of throughput on
    // KITCHEN, is then paid twice to hide 4 ms of a lone path (driver's command: 146 vs 154 Msamples/s, profiles/r06_experiments.json).  How the
    // passes are cut into launches does not enter any sum (a group's samples are added to its slot in sample order, launch after launch).
    unsigned int launchPasses = (unsigned int)ctx->maxBatchFinal;
    {
        unsigned int minePasses = 0;
        for (unsigned int g : mine) minePasses += passesOf(g);
        const size_t pixels = byRank || world == 1 ? n : (size_t)ctx->nPix;
        if (ctx->tuneFinalHalves && ctx->maxDepth < 0 && ctx->tuneSplitDepth > 0 && !ctx->timer.enabled && !ctx->tuneNoOverlap && minePasses >= 2 && minePasses <= launchPasses &&
            pixels * (size_t)ctx->sppPerPass * (size_t)(minePasses / 2) >= ((size_t)1 << 21)) {
            launchPasses = (minePasses + 1) / 2;
            if (G < launchPasses) launchPasses = std::max(G, launchPasses / G * G);  // whole groups per launch
        }
    }
    const unsigned int perLaunch = std::max(1u, launchPasses / G);  // whole groups per launch (G <= launchPasses), else parts of one group
    const unsigned int slots = world > 1 ? nGroups : std::max(perLaunch, std::max(1u, (unsigned int)ctx->maxBatchFinal / G));
    const size_t floats = (world > 1 ? 4 * n : 0) + (size_t)slots * 7 * n;
    if (ctx->d_partials.cap < floats || ctx->partialSlots != slots) {
        HIP_CHECK(ctx->d_partials.reserve(floats));
        ctx->partialSlots = slots;
    }
    HIP_CHECK(hipMemsetAsync(ctx->d_partials.p, 0, floats * 4, ctx->stream));
    const unsigned int firstPassAbs = (unsigned int)ctx->passesRendered;
    const unsigned int step = byRank ? world : 1u;  // distance between this rank's consecutive groups
    const uint64_t pixelsMine = byRank || world == 1 ? (uint64_t)n : (uint64_t)ctx->nPix;
    bool stop = false;
    for (size_t m = 0; m < mine.size() && !stop;) {
        if (ctx->seesCancel()) break;
        if (G <= launchPasses) {
            const size_t cnt = std::min<size_t>(perLaunch, mine.size() - m);
            unsigned int batch = 0;
            for (size_t q = 0; q < cnt; ++q) batch += passesOf(mine[m + q]);
            // (one GPU: the launch's slots are added to image and film right behind its film kernel)
            GroupLaunch gl{firstPassAbs + mine[m] * G, G, step * G, world > 1 ? mine[m] : 0u, world > 1 ? step : 1u, byRank, world == 1 ? (unsigned int)cnt : 0u};
            int rc = renderBatch(ctx, (int)batch, false, &gl, m + cnt >= mine.size());
            if (rc) return rc;
            ctx->samplesLocal += pixelsMine * batch * ctx->sppPerPass;
            m += cnt;
        } else {  // a group larger than a launch: its parts accumulate into the same slot one after the other
            const unsigned int g = mine[m], total = passesOf(g);
            for (unsigned int done = 0; done < total;) {
                if (ctx->seesCancel()) { stop = true; break; }
                const unsigned int batch = std::min(launchPasses, total - done);
                const bool lastPart = done + batch >= total;
                GroupLaunch gl{firstPassAbs + g * G + done, batch, batch, world > 1 ? g : 0u, 1u, byRank, (world == 1 && lastPart) ? 1u : 0u};
                int rc = renderBatch(ctx, (int)batch, false, &gl, lastPart && m + 1 >= mine.size());
                if (rc) return rc;
                ctx->samplesLocal += pixelsMine * batch * ctx->sppPerPass;
                done += batch;
            }
            ++m;
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));  // bound the launch queue; lets ppg_cancel() take effect
    }
    { int rc = flushStragglers(ctx, false); if (rc) return rc; }
    ctx->passesRendered += numPasses; ctx->passesRenderedThisIter += numPasses; ctx->passesLocal += numPasses;
    if (world > 1) { ctx->partialsPending = true; ctx->pendingGroups = nGroups; }
    return ctx->seesCancel() ? PPG_ERR_CANCELLED : PPG_OK;
}

// The owned pixels grouped by region: region of a pixel = spiral rank of its 32x32 block * regions / blocks (include/ppg.h "Rounds by image
// region"); inside a group in the order of the owned-pixel list (row-major), so that a wave is still 64 neighbouring pixels of a row.
int buildRegionLists(ppg_ctx *ctx, int regions) {
    if (ctx->regionsBuilt == regions) return PPG_OK;
    const int bs = 32, bx = (ctx->W + bs - 1) / bs, by = (ctx->H + bs - 1) / bs;
    std::vector<int> rank((size_t)bx * by);
    ppg_spiral_block_ranks(bx, by, rank.data());
    std::vector<unsigned int> own(ctx->nPix);
    if (ctx->nPix) HIP_CHECK(hipMemcpy(own.data(), ctx->d_pixels.p, (size_t)ctx->nPix * 4, hipMemcpyDeviceToHost));
    std::vector<std::vector<unsigned int>> lists((size_t)regions);
    for (unsigned int p : own) {
        const int x = (int)(p % (unsigned int)ctx->W), y = (int)(p / (unsigned int)ctx->W);
        const int r = (int)((long long)rank[(size_t)(y / bs) * bx + (x / bs)] * regions / (bx * by));
        lists[(size_t)r].push_back(p);
    }
    ctx->regionOffset.assign((size_t)regions + 1, 0u);
    std::vector<unsigned int> flat;
    flat.reserve(own.size());
    for (int r = 0; r < regions; ++r) { ctx->regionOffset[(size_t)r] = (unsigned int)flat.size(); flat.insert(flat.end(), lists[(size_t)r].begin(), lists[(size_t)r].end()); }
    ctx->regionOffset[(size_t)regions] = (unsigned int)flat.size();
    HIP_CHECK(ctx->d_regionPixels.reserve(std::max<size_t>(1, flat.size())));
    if (!flat.empty()) HIP_CHECK(hipMemcpyAsync(ctx->d_regionPixels.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (`flat` is pageable host memory)
    ctx->regionsBuilt = regions;
    return PPG_OK;
}

int renderPassesNoStat(ppg_ctx *ctx, int numPasses) {  // GP:1217-1286
    size_t n = (size_t)ctx->W * ctx->H;
    HIP_CHECK(hipMemsetAsync(ctx->d_image.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_sq.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_imageW.p, 0, n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_stats.p, 0, sizeof(BlockStats) * (size_t)ctx->nBlocks, ctx->stream));
    HIP_CHECK(ctx->d_tailLongest.reserve(PPG_TAIL_LOG + 1));
    HIP_CHECK(hipMemsetAsync(ctx->d_tailLongest.p, 0, (PPG_TAIL_LOG + 1) * 4, ctx->stream));
    ctx->tailLaunches = 0;
    ctx->passStart = std::chrono::steady_clock::now();
    ctx->passesLocal = 0;
    // rounds of the sampling-fraction optimiser (include/ppg.h): fractions frozen during a round, its records applied afterwards
    const bool rounds = ctx->loss != LOSS_NONE && ctx->isBuilt && !ctx->isFinalIter;
    const int roundPasses = rounds ? (int)ppg_adam_round_passes(ctx->sppPerPass, ctx->W, ctx->H, numPasses) : (ctx->isFinalIter ? ctx->maxBatchFinal : ctx->maxBatch);
    ctx->samplesLocal = 0;
    ctx->partialsPending = false; ctx->partialsExported = false;
    if (ctx->isFinalIter && ctx->budgetType == 0 && numPasses > 0) return renderFinalGroups(ctx, numPasses);
    // Cancelled while a round hook is installed (a sharded render with a learned sampling fraction): the other ranks enter the hook of
    // EVERY remaining round of this call, so this rank keeps entering it too — with empty rounds (no paths, no records; the host marks its
    // status word) — instead of leaving for an exchange the others are not in.  They all see the status and abort together.
    // rounds by image region (include/ppg.h ppg_set_adam_regions): in a call of at most PPG_ADAM_REGION_MAX_PASSES passes a round is one pass
    // over one group of blocks
    int regions = 0;
    if (rounds && ctx->adamRegions > 1 && numPasses <= PPG_ADAM_REGION_MAX_PASSES) {
        regions = std::min(ctx->adamRegions, ((ctx->W + 31) / 32) * ((ctx->H + 31) / 32));
        int rc = buildRegionLists(ctx, regions);
        if (rc) return rc;
    }
    bool drain = false;
    for (int i = 0; i < numPasses;) {
        if (ctx->seesCancel()) {
            if (rounds && ctx->passHook) drain = true;
            else {
                // (a sharded time budget: the other ranks are about to ask the stop hook after the batch they are rendering — this rank asks it
                // too, once, so that they meet in the same exchange; the host's hook carries its status word there and every rank stops)
                if (ctx->budgetType == 1 && ctx->stopHook) (void)ctx->stopHook(ctx->stopHookUser, PPG_STOP_CANCELLED);
                break;
            }
        }
        const int batch = regions ? 1 : std::min(roundPasses, numPasses - i);
        int rc = PPG_OK;
        if (regions) {
            for (int r = 0; r < regions && !rc; ++r) {
                const unsigned int first = ctx->regionOffset[(size_t)r], count = ctx->regionOffset[(size_t)r + 1] - first;
                // (a group without pixels of this rank — or a cancelled rank kept in step — is an empty round: the hook calls match across ranks)
                rc = renderBatch(ctx, drain ? 0 : 1, true, nullptr, i + 1 >= numPasses && r + 1 == regions, ctx->d_regionPixels.p + first, drain ? 0u : count);
            }
        } else rc = renderBatch(ctx, drain ? 0 : batch, rounds, nullptr, i + batch >= numPasses);
        if (rc) return rc;
        ctx->passesRendered += batch; ctx->passesRenderedThisIter += batch; ctx->passesLocal += batch;
        ctx->samplesLocal += (uint64_t)ctx->nPix * batch * ctx->sppPerPass;
        i += batch;
        if (ctx->budgetType == 1) {  // seconds: the reference checks after every finished pass (GP:1259-1262)
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            int stop = (int)elapsedSeconds(ctx->startTime) > ctx->budget ? 1 : 0;  // `progress = (int) elapsed; shouldAbort = progress > m_budget`
            if (ctx->stopHook) stop = ctx->stopHook(ctx->stopHookUser, stop);           // sharded: the decision of rank 0 for all
            if (stop) break;
        } else if ((i & 63) < batch) {
            HIP_CHECK(hipStreamSynchronize(ctx->stream));  // bound the launch queue; also lets ppg_cancel() take effect
        }
    }
    // what the last batch still owes ("Stragglers"); in a sharded render with rounds whose stragglers' records are deferred, the round of the
    // last stragglers' records is one more call of the round hook on EVERY rank (include/ppg.h "STRAGGLERS")
    {
        const bool deferRounds = rounds && ctx->maxDepth < 0 && ctx->spatialFilter != SF_BOX && !(ctx->doNee && ctx->nee == NEE_KICKSTART);
        int rc = flushStragglers(ctx, deferRounds && ctx->passHook != nullptr);
        if (rc) return rc;
    }
    return ctx->seesCancel() ? PPG_ERR_CANCELLED : PPG_OK;
}

int finishPasses(ppg_ctx *ctx, ppg_pass_stats *st) {  // GP:1288-1328
    if (ctx->partialsPending) { ctx->error = "the groups of this final iteration were not exchanged: ppg_final_partials / ppg_final_partials_commit before ppg_finish_passes"; return PPG_ERR_STATE; }
    const int n = ctx->W * ctx->H;
    const int N = ctx->passesLocal * ctx->sppPerPass;
    if (ctx->sampleCombination == 2) {  // inversevar: m_images.push_back(image->clone())
        ctx->images.emplace_back();
        HIP_CHECK(ctx->images.back().reserve(3 * (size_t)n));
        hipLaunchKernelGGL(k_normalise, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->d_image.p, ctx->d_imageW.p, ctx->images.back().p);
    }
    hipLaunchKernelGGL(k_variance, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->W, N, ctx->d_image.p, ctx->d_sq.p, ctx->d_imageW.p, ctx->d_var.p, ctx->d_lum.p);
    float *lum = ctx->h_lum;  // pinned
    if (ctx->h_statsCap < (size_t)ctx->nBlocks) {
        if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
        ctx->h_stats = nullptr; ctx->h_statsCap = 0;
        HIP_CHECK(hipHostMalloc((void **)&ctx->h_stats, (size_t)ctx->nBlocks * sizeof(BlockStats), hipHostMallocDefault));
        ctx->h_statsCap = (size_t)ctx->nBlocks;
    }
    BlockStats *bs = ctx->h_stats;
    HIP_CHECK(hipMemcpyAsync(lum, ctx->d_lum.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(bs, ctx->d_stats.p, (size_t)ctx->nBlocks * sizeof(BlockStats), hipMemcpyDeviceToHost, ctx->stream));
    unsigned int tailLongest[PPG_TAIL_LOG];
    const unsigned int nTails = std::min<unsigned int>(ctx->tailLaunches, PPG_TAIL_LOG);
    if (nTails) HIP_CHECK(hipMemcpyAsync(tailLongest, ctx->d_tailLongest.p, nTails * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BlockStats c{};
    for (int k = 0; k < ctx->nBlocks; ++k) { const BlockStats &x = bs[k]; c.rays += x.rays; c.path_len += x.path_len; c.committed += x.committed; c.bvh_nodes += x.bvh_nodes; c.bvh_tris += x.bvh_tris; c.max_len = std::max(c.max_len, x.max_len); c.shade_common += x.shade_common; }
    if (ctx->debugBatch) fprintf(stderr, "[ppg passes] iter %d passes %d rays %llu path_len_sum %llu longest path finished by k_tail %llu\n", ctx->iter, ctx->passesLocal, (unsigned long long)c.rays, (unsigned long long)c.path_len, (unsigned long long)c.max_len);
    ctx->bvhNodesVisited += c.bvh_nodes; ctx->bvhTrisTested += c.bvh_tris; ctx->shadeCommonRays += c.shade_common;
    for (unsigned int k = 0; k < nTails; ++k) ctx->tailLongestSum += tailLongest[k];
    float variance = 0;  // summed in the reference's x-major order (GP:1303-1311)
    for (int k = 0; k < n; ++k) variance += lum[k];  // k = x * H + y
    variance /= (float)ctx->W * ctx->H * (N - 1);
    if (ctx->sampleCombination == 2) ctx->variances.push_back(variance);
    ctx->lastVariance = variance;
    ppg_pass_stats s{};
    s.seconds = elapsedSeconds(ctx->passStart);
    s.passes_rendered_total = ctx->passesRendered; s.passes_rendered_local = ctx->passesLocal; s.variance = variance;
    s.samples = ctx->samplesLocal;
    s.rays = c.rays; s.path_length_sum = c.path_len; s.vertices_committed = c.committed;
    ctx->lastStats = s;
    if (st) *st = s;
    ctx->timer.resolve();
    return PPG_OK;
}

int allocFilm(ppg_ctx *ctx) {
    size_t n = (size_t)ctx->W * ctx->H;
    HIP_CHECK(ctx->d_image.reserve(3 * n)); HIP_CHECK(ctx->d_sq.reserve(3 * n)); HIP_CHECK(ctx->d_imageW.reserve(n));
    HIP_CHECK(ctx->d_film.reserve(3 * n)); HIP_CHECK(ctx->d_filmW.reserve(n)); HIP_CHECK(ctx->d_var.reserve(3 * n));
    HIP_CHECK(ctx->d_lum.reserve(n)); HIP_CHECK(ctx->d_tmp.reserve(3 * n));
    if (ctx->h_lumCap < n) {
        if (ctx->h_lum) (void)hipHostFree(ctx->h_lum);
        ctx->h_lum = nullptr; ctx->h_lumCap = 0;
        HIP_CHECK(hipHostMalloc((void **)&ctx->h_lum, n * sizeof(float), hipHostMallocDefault));
        ctx->h_lumCap = n;
    }
    return PPG_OK;
}

int beginRender(ppg_ctx *ctx) {  // GP:1519-1550
    HIP_CHECK(hipSetDevice(ctx->device));
    ctx->quiesce();
    if (!ctx->pathsReady) {  // buffers are sized by scene + shard; normally done by ppg_set_scene / ppg_set_shard
        int rc = allocPaths(ctx);
        if (rc) return rc;
        if ((rc = allocFilm(ctx))) return rc;
        ctx->pathsReady = true;
    }
    // new STree(scene->getAABB()), cubified (GP:850-860)
    float maxSize = 0;
    for (int a = 0; a < 3; ++a) maxSize = ppg_max(maxSize, ctx->aabbMax[a] - ctx->aabbMin[a]);
    {
        float sx = ctx->aabbMax[0] - ctx->aabbMin[0], sy = ctx->aabbMax[1] - ctx->aabbMin[1], sz = ctx->aabbMax[2] - ctx->aabbMin[2];
        maxSize = ppg_max(ppg_max(sx, sy), sz);
    }
    for (int a = 0; a < 3; ++a) {
        ctx->treeMin[a] = ctx->aabbMin[a];
        ctx->treeMax[a] = ctx->aabbMin[a] + maxSize;
        ctx->treeExt[a] = ctx->treeMax[a] - ctx->treeMin[a];  // AABB::getExtents() = max - min
    }
    ctx->snodes.assign(1, HostSNode());
    ctx->hdr.assign(1, LeafHdr{});
    // the initial sampling D-tree: one node, all sums 0 (DTree(), GP:376-381)
    HIP_CHECK(ctx->d_snodes[0].reserve(1));
    HIP_CHECK(hipMemsetAsync(ctx->d_snodes[0].p, 0, sizeof(SNode), ctx->stream));
    ctx->cur = 0;
    ctx->hdr[0].s_base = 0; ctx->hdr[0].s_num = 1;
    {   // the root node goes to the device once; from here on the device owns the S-tree (refineDevice) and the host mirrors it
        int rc = uploadTree(ctx, true);
        if (rc) return rc;
        const unsigned int root = 0;
        HIP_CHECK(ctx->d_dfs[0].reserve(1));
        HIP_CHECK(hipMemcpy(ctx->d_dfs[0].p, &root, 4, hipMemcpyHostToDevice));
        ctx->dfsCur = 0; ctx->dfsCount = 1;
    }
    ctx->nSamplingNodes = 1; ctx->nBuildingNodes = 0;
    ctx->iter = 0; ctx->isFinalIter = false; ctx->isBuilt = false;
    size_t n = (size_t)ctx->W * ctx->H;
    HIP_CHECK(hipMemsetAsync(ctx->d_film.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_filmW.p, 0, n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_var.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_image.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_sq.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_imageW.p, 0, n * 4, ctx->stream));
    ctx->images.clear(); ctx->variances.clear();
#ifdef PPG_PROBE
    HIP_CHECK(ctx->d_probe.reserve(PPG_PROBE_SLOTS));
    HIP_CHECK(hipMemsetAsync(ctx->d_probe.p, 0, PPG_PROBE_SLOTS * 8, ctx->stream));
#endif
    ctx->startTime = std::chrono::steady_clock::now();
    ctx->passesRendered = 0; ctx->passesRenderedThisIter = 0;
    ctx->treeAlive = true;
    // ppg_cancel() is sticky: a cancel no render has acted on yet — it arrived while the scene was being set up, say, seconds of BVH build —
    // cancels THIS render; the flag is consumed here (it used to be cleared, and that cancel was lost).  One the previous render DID act on
    // (it returned PPG_ERR_CANCELLED, or left through a failing hook) is spent.
    {
        const bool pending = ctx->cancelled.exchange(false), spent = ctx->cancelSeen;
        ctx->cancelSeen = false;
        if (pending && !spent) { ctx->error = "cancelled before the render began"; return PPG_ERR_CANCELLED; }
    }
    return PPG_OK;
}

int beginIteration(ppg_ctx *ctx, bool isFinal) {  // GP:1378-1381
    ctx->isFinalIter = isFinal;
    size_t n = (size_t)ctx->W * ctx->H;
    HIP_CHECK(hipMemsetAsync(ctx->d_film.p, 0, 3 * n * 4, ctx->stream));
    HIP_CHECK(hipMemsetAsync(ctx->d_filmW.p, 0, n * 4, ctx->stream));
    return resetSDTree(ctx);
}

int dumpSDTreeFile(ppg_ctx *ctx, const char *path);

int endIteration(ppg_ctx *ctx) {  // GP:1417-1422
    if (ctx->dumpSDTree && !ctx->isFinalIter && !ctx->dumpPrefix.empty()) {
        char buf[1024];
        snprintf(buf, sizeof buf, "%s-%02d.sdt", ctx->dumpPrefix.c_str(), ctx->iter);
        int rc = dumpSDTreeFile(ctx, buf);
        if (rc) return rc;
    }
    ++ctx->iter;
    ctx->passesRenderedThisIter = 0;
    return PPG_OK;
}

int endRender(ppg_ctx *ctx) {  // GP:1567-1582
    ctx->joinTree();
#ifdef PPG_PROBE
    if (ctx->d_probe.p) {
        unsigned long long h[PPG_PROBE_SLOTS];
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(h, ctx->d_probe.p, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ppg probe] lone_bounces %llu coop_rays %llu coop_steps %llu cycles:", h[20], h[22], h[21]);
        for (int k = 0; k <= 11; ++k) fprintf(stderr, " s%d=%llu", k, h[k]);
        fprintf(stderr, "\n");
    }
#endif
    if (ctx->sampleCombination == 2 && !ctx->images.empty()) {
        const int n = ctx->W * ctx->H;
        HIP_CHECK(hipMemsetAsync(ctx->d_film.p, 0, 3 * (size_t)n * 4, ctx->stream));
        std::vector<float> ones((size_t)n, 1.0f);
        HIP_CHECK(hipMemcpyAsync(ctx->d_filmW.p, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        size_t begin = ctx->images.size() - std::min(ctx->images.size(), (size_t)4);
        float totalWeight = 0;
        for (size_t i = begin; i < ctx->variances.size(); ++i) totalWeight += 1.0f / ctx->variances[i];
        for (size_t i = begin; i < ctx->images.size(); ++i) {
            float mult = 1.0f / ctx->variances[i] / totalWeight;
            hipLaunchKernelGGL(k_axpy, dim3((3 * n + 255) / 256), dim3(256), 0, ctx->stream, 3 * n, mult, ctx->images[i].p, ctx->d_film.p);
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PPG_OK;
}

bool doNeeWithSpp(const ppg_ctx *ctx, int spp) {  // GP:1331-1340
    switch (ctx->nee) {
        case NEE_NEVER: return false;
        case NEE_KICKSTART: return spp < 128;
        default: return true;
    }
}

int renderSPP(ppg_ctx *ctx) {  // GP:1342-1426
    size_t sampleCount = (size_t)ctx->budget;
    int nPasses = (int)std::ceil(sampleCount / (float)ctx->sppPerPass);
    float currentVarAtEnd = std::numeric_limits<float>::infinity();
    while (ctx->passesRendered < nPasses) {
        const int sppRendered = ctx->passesRendered * ctx->sppPerPass;
        ctx->doNee = doNeeWithSpp(ctx, sppRendered);
        int remainingPasses = nPasses - ctx->passesRendered;
        int passesThisIteration = std::min(remainingPasses, 1 << ctx->iter);
        if (remainingPasses - passesThisIteration < 2 * passesThisIteration) passesThisIteration = remainingPasses;
        int rc = beginIteration(ctx, passesThisIteration >= remainingPasses);
        if (rc) return rc;
        ppg_pass_stats st;
        if ((rc = renderPassesNoStat(ctx, passesThisIteration))) return rc;
        if ((rc = finishPasses(ctx, &st))) return rc;
        float variance = st.variance;
        const float lastVarAtEnd = currentVarAtEnd;
        currentVarAtEnd = passesThisIteration * variance / remainingPasses;
        remainingPasses -= passesThisIteration;
        if (ctx->sampleCombination == 1 && remainingPasses > 0 &&
            (remainingPasses < passesThisIteration || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
            ctx->isFinalIter = true;
            if ((rc = renderPassesNoStat(ctx, remainingPasses))) return rc;
            if ((rc = finishPasses(ctx, &st))) return rc;
        }
        if ((rc = buildSDTree(ctx, nullptr))) return rc;
        if ((rc = endIteration(ctx))) return rc;
    }
    return PPG_OK;
}

int renderTime(ppg_ctx *ctx) {  // GP:1434-1514
    float nSeconds = ctx->budget;
    float currentVarAtEnd = std::numeric_limits<float>::infinity();
    float elapsed = 0;
    while (elapsed < nSeconds) {
        const int sppRendered = ctx->passesRendered * ctx->sppPerPass;
        ctx->doNee = doNeeWithSpp(ctx, sppRendered);
        float remainingTime = nSeconds - elapsed;
        const int passesThisIteration = 1 << ctx->iter;
        const auto startIter = std::chrono::steady_clock::now();
        int rc = beginIteration(ctx, false);
        if (rc) return rc;
        ppg_pass_stats st;
        if ((rc = renderPassesNoStat(ctx, passesThisIteration))) return rc;
        if ((rc = finishPasses(ctx, &st))) return rc;
        float variance = st.variance;
        const float secondsIter = elapsedSeconds(startIter);
        const float lastVarAtEnd = currentVarAtEnd;
        currentVarAtEnd = secondsIter * variance / remainingTime;
        remainingTime -= secondsIter;
        if (ctx->sampleCombination == 1 && remainingTime > 0 &&
            (remainingTime < secondsIter || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
            ctx->isFinalIter = true;
            do {
                if ((rc = renderPassesNoStat(ctx, passesThisIteration))) return rc;
                if ((rc = finishPasses(ctx, &st))) return rc;
                elapsed = elapsedSeconds(ctx->startTime);
            } while (elapsed < nSeconds);
        }
        if ((rc = buildSDTree(ctx, nullptr))) return rc;
        if ((rc = endIteration(ctx))) return rc;
        elapsed = elapsedSeconds(ctx->startTime);
    }
    return PPG_OK;
}

// gather one kind of D-trees per S-tree leaf into host arrays (expands shared sampling blocks)
int gatherDTrees(ppg_ctx *ctx, int which, std::vector<float> &sums, std::vector<uint16_t> &children, std::vector<uint64_t> *fixed) {
    sums.clear(); children.clear(); if (fixed) fixed->clear();
    if (which == 0) {
        std::vector<SNode> pool(ctx->nSamplingNodes);
        if (!pool.empty()) HIP_CHECK(hipMemcpy(pool.data(), ctx->d_snodes[ctx->cur].p, pool.size() * sizeof(SNode), hipMemcpyDeviceToHost));
        for (unsigned int leaf : ctx->leaves) {
            const LeafHdr &h = ctx->hdr[leaf];
            for (unsigned int n = 0; n < h.s_num; ++n)
                for (int j = 0; j < 4; ++j) {
                    sums.push_back(pool[h.s_base + n].sum[j]); children.push_back(pool[h.s_base + n].child[j]);
                    if (fixed) fixed->push_back(0);
                }
        }
    } else {
        std::vector<ushort4> ch(ctx->nBuildingNodes);
        std::vector<unsigned long long> acc(ctx->nBuildingNodes * 4);
        if (!ch.empty()) {
            HIP_CHECK(hipMemcpy(ch.data(), ctx->d_bchild.p, ch.size() * sizeof(ushort4), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(acc.data(), ctx->d_bacc.p, acc.size() * 8, hipMemcpyDeviceToHost));
        }
        std::vector<LeafHdr> dh(ctx->hdr.size());
        HIP_CHECK(hipMemcpy(dh.data(), ctx->d_hdr.p, dh.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost));
        for (unsigned int leaf : ctx->leaves) {
            const LeafHdr &h = dh[leaf];
            for (unsigned int n = 0; n < h.b_num; ++n) {
                const unsigned short cc[4] = {ch[h.b_base + n].x, ch[h.b_base + n].y, ch[h.b_base + n].z, ch[h.b_base + n].w};
                for (int j = 0; j < 4; ++j) {
                    uint64_t a = acc[(size_t)(h.b_base + n) * 4 + j];
                    sums.push_back(cc[j] == 0 ? ppg_from_fixed(a) : 0.0f); children.push_back(cc[j]);
                    if (fixed) fixed->push_back(a);
                }
            }
        }
    }
    return PPG_OK;
}

int dumpSDTreeFile(ppg_ctx *ctx, const char *path) {  // GP:1191-1208 + STree::dump GP:945-951 + DTreeWrapper::dump GP:699-711
    std::vector<SNode> pool(ctx->nSamplingNodes);
    if (!pool.empty()) HIP_CHECK(hipMemcpy(pool.data(), ctx->d_snodes[ctx->cur].p, pool.size() * sizeof(SNode), hipMemcpyDeviceToHost));
    FILE *f = fopen(path, "wb");
    if (!f) { ctx->error = std::string("cannot open ") + path; return PPG_ERR_INVALID; }
    fwrite(ctx->scene.cam.c2w, 4, 16, f);
    // STreeNode::forEachLeaf (GP:796-813): depth-first, child 0 before child 1
    struct E { size_t idx; float p[3], s[3]; };
    std::vector<E> st;
    E root; root.idx = 0;
    for (int a = 0; a < 3; ++a) { root.p[a] = ctx->treeMin[a]; root.s[a] = ctx->treeMax[a] - ctx->treeMin[a]; }
    st.push_back(root);
    while (!st.empty()) {
        E e = st.back(); st.pop_back();
        const HostSNode &n = ctx->snodes[e.idx];
        if (n.isLeaf()) {
            const LeafHdr &h = ctx->hdr[e.idx];
            if (h.s_statw > 0) {
                float mean = 0;
                if (h.s_statw != 0) { const float factor = 1 / (PPG_PI_F * 4 * h.s_statw); mean = factor * h.s_sum; }
                float hd[7] = {e.p[0], e.p[1], e.p[2], e.s[0], e.s[1], e.s[2], mean};
                fwrite(hd, 4, 7, f);
                uint64_t sw = (uint64_t)h.s_statw, nn = h.s_num;
                fwrite(&sw, 8, 1, f); fwrite(&nn, 8, 1, f);
                for (unsigned int k = 0; k < h.s_num; ++k)
                    for (int j = 0; j < 4; ++j) { fwrite(&pool[h.s_base + k].sum[j], 4, 1, f); fwrite(&pool[h.s_base + k].child[j], 2, 1, f); }
            }
        } else {
            E c0 = e, c1 = e;
            c0.s[n.axis] /= 2; c1.s[n.axis] = c0.s[n.axis];
            c0.idx = n.child[0]; c1.idx = n.child[1];
            c1.p[n.axis] += c1.s[n.axis];
            st.push_back(c1); st.push_back(c0);
        }
    }
    fclose(f);
    return PPG_OK;
}

}  // namespace

// ================================================================================================
// C-ABI (include/ppg.h)
// ================================================================================================
extern "C" {

void ppg_config_default(ppg_config *cfg) {
    memset(cfg, 0, sizeof *cfg);
    cfg->nee = "never"; cfg->sampleCombination = "automatic"; cfg->spatialFilter = "nearest"; cfg->directionalFilter = "nearest";
    cfg->bsdfSamplingFractionLoss = "none"; cfg->sdTreeMaxMemory = -1; cfg->sTreeThreshold = 12000; cfg->dTreeThreshold = 0.01f;
    cfg->bsdfSamplingFraction = 0.5f; cfg->sppPerPass = 4; cfg->budgetType = "seconds"; cfg->budget = 300.0f; cfg->dumpSDTree = 0;
    cfg->rrDepth = 5; cfg->maxDepth = -1; cfg->strictNormals = 0; cfg->hideEmitters = 0; cfg->seed = 0; cfg->device = 0; cfg->dumpPrefix = nullptr;
}

const char *ppg_description(void) { return "Guided path tracer"; }

int32_t ppg_adam_round_passes(int32_t spp_per_pass, int32_t width, int32_t height, int32_t n_passes) {
    const uint64_t perPass = (uint64_t)std::max(1, spp_per_pass) * (uint64_t)std::max(1, width) * (uint64_t)std::max(1, height);
    int32_t r = 1;
    while (r * 2 <= PPG_ADAM_ROUND_MAX_PASSES && r * 4 <= n_passes && (uint64_t)(r * 2) * perPass <= PPG_ADAM_ROUND_MAX_PATHS) r *= 2;
    return r;
}

int ppg_create(const ppg_config *cfg, ppg_ctx **out) {
    if (!cfg || !out) { g_createError = "null argument"; return PPG_ERR_INVALID; }
    std::unique_ptr<ppg_ctx> c(new ppg_ctx());
#define PARSE(field, dflt, target, ...)                                                                                  \
    c->target = parseEnum(cfg->field, dflt, {__VA_ARGS__});                                                              \
    if (c->target < 0) { g_createError = std::string("invalid value for '" #field "': ") + (cfg->field ? cfg->field : ""); return PPG_ERR_INVALID; }
    PARSE(nee, "never", nee, "never", "kickstart", "always")
    PARSE(sampleCombination, "automatic", sampleCombination, "discard", "automatic", "inversevar")
    PARSE(spatialFilter, "nearest", spatialFilter, "nearest", "stochastic", "box")
    PARSE(directionalFilter, "nearest", directionalFilter, "nearest", "box")
    PARSE(bsdfSamplingFractionLoss, "none", loss, "none", "kl", "var")
    PARSE(budgetType, "seconds", budgetType, "spp", "seconds")
#undef PARSE
    c->sdTreeMaxMemory = cfg->sdTreeMaxMemory; c->sTreeThreshold = cfg->sTreeThreshold; c->dTreeThreshold = cfg->dTreeThreshold;
    c->bsdfSamplingFraction = cfg->bsdfSamplingFraction; c->sppPerPass = cfg->sppPerPass; c->budget = cfg->budget;
    c->dumpSDTree = cfg->dumpSDTree != 0; c->rrDepth = cfg->rrDepth; c->maxDepth = cfg->maxDepth;
    c->strictNormals = cfg->strictNormals != 0; c->hideEmitters = cfg->hideEmitters != 0; c->seed = cfg->seed; c->device = cfg->device;
    if (cfg->dumpPrefix) c->dumpPrefix = cfg->dumpPrefix;
    {   // tuning switches (performance experiments only; results do not depend on them)
        if (const char *e = getenv("PPG_TAIL_THRESHOLD")) c->tailThreshold = (unsigned int)std::max(0ll, atoll(e));
        if (const char *e = getenv("PPG_TAIL_MIN")) c->tailMin = (unsigned int)std::max(1ll, atoll(e));
        if (const char *e = getenv("PPG_TAIL_DIV")) c->tailDiv = (unsigned int)std::max(1ll, atoll(e));
        if (const char *e = getenv("PPG_BATCH_PATHS")) c->tuneBatchPaths = (size_t)std::max(1ll, atoll(e));
        if (const char *e = getenv("PPG_BLOCKS")) c->tuneBlocks = std::max(1, atoi(e));
        c->tuneForceBvh = getenv("PPG_FORCE_BVH") != nullptr;
        c->tuneFuse = getenv("PPG_FUSE") != nullptr;
        c->tuneNoSort = getenv("PPG_NO_SORT") != nullptr;
        c->tuneNoSplit = getenv("PPG_NO_SPLIT") != nullptr;
        c->tuneNoSortFirst = getenv("PPG_NO_SORT_FIRST") != nullptr;
        c->tuneNoOverlap = getenv("PPG_NO_OVERLAP") != nullptr;
        c->tuneNoSortedCommit = getenv("PPG_NO_SORTED_COMMIT") != nullptr;
        c->tuneAdamUnordered = getenv("PPG_ADAM_UNORDERED") != nullptr;
        c->tuneNoAside = getenv("PPG_NO_ASIDE") != nullptr;
        if (const char *e = getenv("PPG_SPLAT_LDS_NODES")) c->tuneSplatLdsNodes = (unsigned int)std::max(0, std::min((int)PPG_SPLAT_NODES, atoi(e)));
        if (const char *e = getenv("PPG_BULK_BOUNCES")) c->tuneBulkBounces = std::max(0, atoi(e));
        if (const char *e = getenv("PPG_BOUNCE_MARGIN")) c->bounceMargin = std::max(0, atoi(e));
        c->debugBatch = getenv("PPG_DEBUG_BATCH") != nullptr;
        if (const char *e = getenv("PPG_TAIL_BLOCKS")) c->tuneTailBlocks = std::max(1, atoi(e));
        if (const char *e = getenv("PPG_FINAL_BATCH")) c->tuneFinalBatch = std::max(1, atoi(e));
        if (const char *e = getenv("PPG_PATH_LAYOUT")) c->tunePathLayout = !strcmp(e, "aos") ? 2 : (!strcmp(e, "pack") ? 3 : (!strcmp(e, "soa") ? 1 : 0));
        if (const char *e = getenv("PPG_SORT_KERNEL")) c->tuneSortKernel = atoi(e) != 0;
        if (const char *e = getenv("PPG_BLOCKS_SMALL")) c->tuneBlocksSmall = std::max(0, atoi(e));
        if (const char *e = getenv("PPG_SMALL_PATHS")) c->tuneSmallPaths = (size_t)std::max(0ll, atoll(e));
        if (const char *e = getenv("PPG_BVH_LEAF")) c->tuneBvhLeaf = std::max(1, std::min(8, atoi(e)));
        if (const char *e = getenv("PPG_BVH_PAD")) c->tuneBvhPad = (float)atof(e);
        if (const char *e = getenv("PPG_SPLIT_DEPTH")) c->tuneSplitDepth = std::max(0, atoi(e));
        c->tuneFinalHalves = getenv("PPG_FINAL_HALVES") != nullptr;
    }
    if (c->sppPerPass <= 0) { g_createError = "sppPerPass must be > 0"; return PPG_ERR_INVALID; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_createError = std::string("no HIP device available: ") + hipGetErrorString(e); return PPG_ERR_DEVICE; }
    if (c->device < 0 || c->device >= ndev) { g_createError = "device ordinal out of range"; return PPG_ERR_INVALID; }
    if ((e = hipSetDevice(c->device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess || (e = createSecondStream(&c->stream2)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamCreate(&c->stream3)) != hipSuccess || (e = hipEventCreateWithFlags(&c->evTreeFork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamCreate(&c->stream4)) != hipSuccess || (e = hipEventCreateWithFlags(&c->evStragFork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->evStragDone, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->evSplatDone, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&c->evAdamDone, hipEventDisableTiming)) != hipSuccess) {
        g_createError = std::string("HIP init failed: ") + hipGetErrorString(e);
        return PPG_ERR_DEVICE;
    }
    *out = c.release();
    return PPG_OK;
}

void ppg_destroy(ppg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    g_blockCache.settle();
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx->stream4) (void)hipStreamSynchronize(ctx->stream4);
    g_blockCache.settle();
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
    for (hipEvent_t ev : {ctx->evStragFork, ctx->evStragDone}) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {ctx->evTreeFork, ctx->evSplatDone, ctx->evAdamDone}) if (ev) (void)hipEventDestroy(ev);
    if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
    if (ctx->evJoin) (void)hipEventDestroy(ctx->evJoin);
    ctx->timer.resolve();
    for (auto e : ctx->timer.pool) (void)hipEventDestroy(e);
    if (ctx->h_lum) (void)hipHostFree(ctx->h_lum);
    if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
    if (ctx->h_round) (void)hipHostFree(ctx->h_round);
    hipStream_t s = ctx->stream;
    delete ctx;
    if (s) (void)hipStreamDestroy(s);
}

const char *ppg_last_error(const ppg_ctx *ctx) { return ctx ? ctx->error.c_str() : g_createError.c_str(); }

int ppg_set_scene(ppg_ctx *ctx, const ppg_scene *s) {
    // a scene needs materials and at least one primitive; triangle arrays may be absent when it consists of analytic spheres only
    if (!s || !s->materials || (s->n_triangles == 0 && s->n_spheres == 0) ||
        (s->n_triangles > 0 && (!s->positions || !s->indices || !s->tri_material || !s->tri_emitter))) {
        ctx->error = "incomplete scene";
        return PPG_ERR_INVALID;
    }
    HIP_CHECK(hipSetDevice(ctx->device));
    ctx->quiesce();
    for (uint32_t t = 0; t < s->n_triangles; ++t) {
        if (s->tri_material[t] >= s->n_materials || s->tri_emitter[t] >= (int32_t)s->n_emitters) { ctx->error = "index out of range"; return PPG_ERR_INVALID; }
        if (s->materials[s->tri_material[t]].type < 0 || s->materials[s->tri_material[t]].type > PPG_BSDF_LAST) { ctx->error = "unsupported BSDF type"; return PPG_ERR_INVALID; }
        for (int k = 0; k < 3; ++k) if (s->indices[3 * t + k] >= s->n_vertices) { ctx->error = "vertex index out of range"; return PPG_ERR_INVALID; }
    }
    std::vector<int> emitterSphere(s->n_emitters, -1);  // per emitter: the analytic sphere carrying it
    if (s->n_spheres) {
        if (!s->spheres) { ctx->error = "spheres: n_spheres > 0 but no array"; return PPG_ERR_INVALID; }
        std::vector<int> users(s->n_emitters, 0);
        for (uint32_t t = 0; t < s->n_triangles; ++t) if (s->tri_emitter[t] >= 0) users[s->tri_emitter[t]] = 1;
        for (uint32_t k = 0; k < s->n_spheres; ++k) {
            const ppg_sphere &sp = s->spheres[k];
            if (!(sp.radius > 0)) { ctx->error = "sphere: radius must be > 0"; return PPG_ERR_INVALID; }
            if (sp.material >= s->n_materials || sp.emitter >= (int32_t)s->n_emitters) { ctx->error = "index out of range"; return PPG_ERR_INVALID; }
            if (s->materials[sp.material].type < 0 || s->materials[sp.material].type > PPG_BSDF_LAST) { ctx->error = "unsupported BSDF type"; return PPG_ERR_INVALID; }
            if (sp.emitter >= 0 && users[sp.emitter]++) { ctx->error = "sphere: its emitter is shared with another shape"; return PPG_ERR_INVALID; }
            if (sp.emitter >= 0) emitterSphere[sp.emitter] = (int)k;
        }
    }
    // Scene::getAABB(): kd-tree box enlarged by MTS_KD_AABB_EPSILON (gkdtree.h:1213-1220) + sensor position (scene.cpp:386-414)
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t t = 0; t < 3 * (size_t)s->n_triangles; ++t)
        for (int a = 0; a < 3; ++a) { float v = s->positions[3 * s->indices[t] + a]; mn[a] = ppg_min(mn[a], v); mx[a] = ppg_max(mx[a], v); }
    for (uint32_t k = 0; k < s->n_spheres; ++k)  // Sphere::getAABB, sphere.cpp:152-157
        for (int a = 0; a < 3; ++a) { mn[a] = ppg_min(mn[a], s->spheres[k].center[a] - s->spheres[k].radius); mx[a] = ppg_max(mx[a], s->spheres[k].center[a] + s->spheres[k].radius); }
    const float eps = 1e-3f;
    for (int a = 0; a < 3; ++a) {
        ctx->aabbMin[a] = mn[a] - ((mx[a] - mn[a]) * eps + eps);
        ctx->aabbMax[a] = mx[a] + ((mx[a] - ctx->aabbMin[a]) * eps + eps);
        float c = s->camera.camera_to_world[4 * a + 3];
        ctx->aabbMin[a] = ppg_min(ctx->aabbMin[a], c);
        ctx->aabbMax[a] = ppg_max(ctx->aabbMax[a], c);
    }
    float ext = 0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, mx[a] - mn[a]);
    if (!(ext < 1e9f)) { ctx->error = "scene extent of 1e9 units or more (or not finite)"; return PPG_ERR_INVALID; }  // BvhBuilder::quantise
    BvhBuilder bb;
    // Box padding: the BVH must return what brute force returns.  A slab test carries a few ulps of error in t, i.e. up to
    // ~4 * 2^-24 * (distance travelled) in position; 2e-6 * (scene extent) leaves an 8x margin.  (1e-4 * extent, the first
    // choice, made the leaf boxes of a finely tessellated model under a 100 m sky dome several triangles thick: 4x slower.)
    const float padRel = ctx->tuneBvhPad;
    bb.maxLeaf = ctx->tuneBvhLeaf;
    bb.run(s->positions, s->indices, s->n_triangles, padRel * ext + 1e-30f);
    std::vector<float4> tris(3 * (size_t)s->n_triangles), nrm, accel(3 * (size_t)s->n_triangles);
    if (s->normals) nrm.resize(tris.size());
    for (uint32_t k = 0; k < s->n_triangles; ++k) {
        uint32_t t = bb.order[k];
        {   // TriAccel::load (triaccel.h:62-97), same float operations as the oracle's
            const float *A = s->positions + 3 * s->indices[3 * t], *B = s->positions + 3 * s->indices[3 * t + 1], *C = s->positions + 3 * s->indices[3 * t + 2];
            static const int waldModulo[4] = {1, 2, 0, 1};
            const float b[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]}, c[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
            const float N[3] = {c[1] * b[2] - c[2] * b[1], c[2] * b[0] - c[0] * b[2], c[0] * b[1] - c[1] * b[0]};
            int kk = 0;
            for (int j = 0; j < 3; j++) if (ppg_abs(N[j]) > ppg_abs(N[kk])) kk = j;
            const int u = waldModulo[kk], v = waldModulo[kk + 1];
            const float n_k = N[kk], denom = b[u] * c[v] - b[v] * c[u];
            float n_u = 0, n_v = 0, n_d = 0, a_u = 0, a_v = 0, b_nu = 0, b_nv = 0, c_nu = 0, c_nv = 0;
            if (denom == 0) {
                kk = 3;
            } else {
                n_u = N[u] / n_k; n_v = N[v] / n_k;
                n_d = (A[0] * N[0] + A[1] * N[1] + A[2] * N[2]) / n_k;
                b_nu = b[u] / denom; b_nv = -b[v] / denom;
                a_u = A[u]; a_v = A[v];
                c_nu = c[v] / denom; c_nv = -c[u] / denom;
            }
            accel[3 * k + 0] = make_float4(n_u, n_v, n_d, __builtin_bit_cast(float, kk));
            accel[3 * k + 1] = make_float4(a_u, a_v, b_nu, b_nv);
            accel[3 * k + 2] = make_float4(c_nu, c_nv, 0.0f, __builtin_bit_cast(float, (int)t));
        }
        for (int v = 0; v < 3; ++v) {
            const float *p = s->positions + 3 * s->indices[3 * t + v];
            float w = v == 0 ? __builtin_bit_cast(float, (int)s->tri_material[t]) : (v == 1 ? __builtin_bit_cast(float, (int)s->tri_emitter[t]) : __builtin_bit_cast(float, (int)t));
            tris[3 * k + v] = make_float4(p[0], p[1], p[2], w);
            if (s->normals) { const float *n = s->normals + 3 * s->indices[3 * t + v]; nrm[3 * k + v] = make_float4(n[0], n[1], n[2], 0); }
        }
    }
    // material table: 4 x float4 per BSDF = (reflectance, type) (specular, alpha) (eta, flags) (k, fdrInt) (opacity, rtrans slice); the lean kernels read
    // only the first.  configure()-time normalisation as in the plugins: "none" conductor = (eta 0, k 1) (conductor.cpp:171-173),
    // GGX alpha clamped (microfacet.h:135), plastic's internal diffuse Fresnel reflectance (plastic.cpp:191-193)
    std::vector<float4> mats(PPG_MAT_STRIDE * (size_t)s->n_materials), ems(std::max<uint32_t>(1, s->n_emitters));
    ctx->fullMaterials = false;
    bool hasNull = false;
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        ppg_material m = s->materials[i];
        if (m.type == PPG_BSDF_DIFFUSE && m.flags == PPG_MAT_TWOSIDED) { m.type = PPG_BSDF_TWOSIDED_DIFFUSE; m.flags = 0; }
        if (m.type == PPG_BSDF_TWOSIDED_DIFFUSE) m.flags &= ~PPG_MAT_TWOSIDED;
        if (m.type == PPG_BSDF_MIRROR) for (int c = 0; c < 3; ++c) { m.eta[c] = 0.0f; m.k[c] = 1.0f; }
        if (m.type == PPG_BSDF_ROUGHCONDUCTOR || m.type == PPG_BSDF_ROUGHDIELECTRIC || m.type == PPG_BSDF_ROUGHPLASTIC) m.alpha = ppg_max(m.alpha, 1e-4f);
        if ((m.type == PPG_BSDF_PLASTIC || m.type == PPG_BSDF_DIELECTRIC || m.type == PPG_BSDF_THINDIELECTRIC || m.type == PPG_BSDF_ROUGHDIELECTRIC) && !(m.eta[0] > 0)) { ctx->error = "plastic / dielectric need eta[0] = intIOR / extIOR > 0"; return PPG_ERR_INVALID; }
        float fdrInt = m.type == PPG_BSDF_PLASTIC ? ppg_fresnel_diffuse_reflectance(1 / m.eta[0]) : 0.0f;
        if (m.type == PPG_BSDF_ROUGHPLASTIC) {  // Fdr = 1 - internal diffuse transmittance, the last entry of the slice (roughplastic.cpp:372)
            if (m.rtrans < 0 || (uint32_t)m.rtrans >= s->n_rtrans || !s->rtrans || s->rtrans_samples < 2) { ctx->error = "roughplastic: material.rtrans is not a slice of scene.rtrans"; return PPG_ERR_INVALID; }
            if (!(m.eta[0] > 0)) { ctx->error = "plastic / dielectric need eta[0] = intIOR / extIOR > 0"; return PPG_ERR_INVALID; }
            fdrInt = 1 - s->rtrans[(size_t)m.rtrans * (s->rtrans_samples + 1) + s->rtrans_samples];
        } else m.rtrans = 0;
        if (m.type > PPG_BSDF_MIRROR || m.flags != 0) ctx->fullMaterials = true;
        if (m.type == PPG_BSDF_THINDIELECTRIC || (m.flags & PPG_MAT_MASK)) hasNull = true;
        mats[PPG_MAT_STRIDE * i + 0] = make_float4(m.reflectance[0], m.reflectance[1], m.reflectance[2], (float)m.type);
        mats[PPG_MAT_STRIDE * i + 1] = make_float4(m.specular[0], m.specular[1], m.specular[2], m.alpha);
        mats[PPG_MAT_STRIDE * i + 2] = make_float4(m.eta[0], m.eta[1], m.eta[2], __builtin_bit_cast(float, m.flags));
        mats[PPG_MAT_STRIDE * i + 3] = make_float4(m.k[0], m.k[1], m.k[2], fdrInt);
        mats[PPG_MAT_STRIDE * i + 4] = make_float4(m.opacity[0], m.opacity[1], m.opacity[2], __builtin_bit_cast(float, m.rtrans));
        {   // bitmap on the diffuse reflectance / bump map around the BSDF
            const uint32_t ta = m.texture & 0xffffu, tb = m.texture >> 16;
            if (ta > s->n_textures || tb > s->n_textures || (m.texture && !s->textures)) { ctx->error = "material.texture: index out of range"; return PPG_ERR_INVALID; }
            if (ta && m.type != PPG_BSDF_DIFFUSE && m.type != PPG_BSDF_TWOSIDED_DIFFUSE && m.type != PPG_BSDF_PLASTIC && m.type != PPG_BSDF_ROUGHPLASTIC) {
                ctx->error = "material.texture: only the diffuse reflectance of diffuse / plastic / roughplastic can carry a bitmap"; return PPG_ERR_INVALID;
            }
            if (m.texture) ctx->fullMaterials = true;  // the texture code lives in the FULL kernel variants
            mats[PPG_MAT_STRIDE * i + 5] = make_float4(__builtin_bit_cast(float, m.texture), 0.0f, 0.0f, 0.0f);
        }
    }
    for (uint32_t k = 0; k < s->n_spheres; ++k)
        if (s->materials[s->spheres[k].material].texture) { ctx->error = "sphere: textured BSDFs are only supported on triangle meshes"; return PPG_ERR_INVALID; }
    ctx->scene.uvs = nullptr; ctx->scene.textures = nullptr;
    if (s->texcoords) {
        std::vector<float2> uv(3 * (size_t)s->n_triangles);
        for (uint32_t k = 0; k < s->n_triangles; ++k) {
            const uint32_t t = bb.order[k];
            for (int v = 0; v < 3; ++v) { const float *q = s->texcoords + 2 * (size_t)s->indices[3 * t + v]; uv[3 * (size_t)k + v] = make_float2(q[0], q[1]); }
        }
        HIP_CHECK(ctx->d_uvs.reserve(uv.size()));
        HIP_CHECK(hipMemcpy(ctx->d_uvs.p, uv.data(), uv.size() * sizeof(float2), hipMemcpyHostToDevice));
        ctx->scene.uvs = ctx->d_uvs.p;
    }
    ctx->d_texTexels.clear();
    if (s->n_textures) {
        if (!s->textures) { ctx->error = "textures: n_textures > 0 but no array"; return PPG_ERR_INVALID; }
        std::vector<DevTex> table(s->n_textures);
        ctx->d_texTexels.resize(s->n_textures);
        for (uint32_t i = 0; i < s->n_textures; ++i) {
            const ppg_texture &t = s->textures[i];
            if (!t.rgb || t.width == 0 || t.height == 0 || t.width > 0x7fff || t.height > 0x7fff || t.wrap_u < 0 || t.wrap_u > PPG_WRAP_ONE || t.wrap_v < 0 || t.wrap_v > PPG_WRAP_ONE) {
                ctx->error = "texture: needs pixels, 0 < width, height < 32768 and valid wrap modes"; return PPG_ERR_INVALID;
            }
            const size_t n = (size_t)t.width * t.height;
            std::vector<float4> px(n);
            for (size_t k = 0; k < n; ++k) px[k] = make_float4(t.rgb[3 * k], t.rgb[3 * k + 1], t.rgb[3 * k + 2], 0.0f);
            HIP_CHECK(ctx->d_texTexels[i].reserve(n));
            HIP_CHECK(hipMemcpy(ctx->d_texTexels[i].p, px.data(), n * sizeof(float4), hipMemcpyHostToDevice));
            DevTex &d = table[i];
            d.texels = ctx->d_texTexels[i].p; d.w = (int)t.width; d.h = (int)t.height; d.su = t.uv_scale[0]; d.sv = t.uv_scale[1]; d.ou = t.uv_offset[0]; d.ov = t.uv_offset[1];
            d.wrap_u = t.wrap_u; d.wrap_v = t.wrap_v; d.nearest = t.nearest != 0; d.pad = 0;
        }
        HIP_CHECK(ctx->d_textures.reserve(table.size()));
        HIP_CHECK(hipMemcpy(ctx->d_textures.p, table.data(), table.size() * sizeof(DevTex), hipMemcpyHostToDevice));
        ctx->scene.textures = ctx->d_textures.p;
    }
    for (uint32_t i = 0; i < s->n_emitters; ++i) ems[i] = make_float4(s->emitters[i].radiance[0], s->emitters[i].radiance[1], s->emitters[i].radiance[2], 0);
    HIP_CHECK(ctx->d_tris.reserve(std::max<size_t>(tris.size(), 3)));
    HIP_CHECK(hipMemcpy(ctx->d_tris.p, tris.data(), tris.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_CHECK(ctx->d_accel.reserve(std::max<size_t>(accel.size(), 3)));
    HIP_CHECK(hipMemcpy(ctx->d_accel.p, accel.data(), accel.size() * sizeof(float4), hipMemcpyHostToDevice));
    {   // small scenes are traced by brute force from LDS: the same records grouped by projection axis (degenerate ones dropped,
        // the rest padded with never-hit records so that n_tris records can always be staged), leaf-order index in record[2].z
        std::vector<float4> small(accel.size(), make_float4(0, 0, 0, __builtin_bit_cast(float, 3)));
        int n[3] = {0, 0, 0};
        if (s->n_triangles <= 64) {
            size_t w = 0;
            for (int axis = 0; axis < 3; ++axis)
                for (uint32_t k = 0; k < s->n_triangles; ++k)
                    if (__builtin_bit_cast(int, accel[3 * k].w) == axis) {
                        small[3 * w] = accel[3 * k]; small[3 * w + 1] = accel[3 * k + 1]; small[3 * w + 2] = accel[3 * k + 2];
                        small[3 * w + 2].z = __builtin_bit_cast(float, (int)k);
                        ++w; ++n[axis];
                    }
        }
        HIP_CHECK(ctx->d_accelSmall.reserve(std::max<size_t>(small.size(), 3)));
        HIP_CHECK(hipMemcpy(ctx->d_accelSmall.p, small.data(), small.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (int axis = 0; axis < 3; ++axis) ctx->scene.small_n[axis] = n[axis];
        ctx->scene.accel_small = ctx->d_accelSmall.p;
    }
    if (s->normals) { HIP_CHECK(ctx->d_normals.reserve(nrm.size())); HIP_CHECK(hipMemcpy(ctx->d_normals.p, nrm.data(), nrm.size() * sizeof(float4), hipMemcpyHostToDevice)); }
    HIP_CHECK(ctx->d_bvh.reserve(std::max<size_t>(bb.nodes.size(), 1)));
    HIP_CHECK(hipMemcpy(ctx->d_bvh.p, bb.nodes.data(), bb.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    HIP_CHECK(ctx->d_bvh4.reserve(bb.nodes4q.size()));
    HIP_CHECK(hipMemcpy(ctx->d_bvh4.p, bb.nodes4q.data(), bb.nodes4q.size() * sizeof(Bvh4QNode), hipMemcpyHostToDevice));
    HIP_CHECK(ctx->d_bvhTop.reserve(std::max<size_t>(2, bb.topCut.size() / 4)));
    if (!bb.topCut.empty()) HIP_CHECK(hipMemcpy(ctx->d_bvhTop.p, bb.topCut.data(), bb.topCut.size() * sizeof(float), hipMemcpyHostToDevice));
    const int nTop = getenv("PPG_NO_TOPCUT") ? 0 : (int)(bb.topCut.size() / 8);
    if (s->n_rtrans && s->rtrans) {
        const size_t n = (size_t)s->n_rtrans * (s->rtrans_samples + 1);
        HIP_CHECK(ctx->d_rtrans.reserve(n));
        HIP_CHECK(hipMemcpy(ctx->d_rtrans.p, s->rtrans, n * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_CHECK(ctx->d_materials.reserve(mats.size()));
    HIP_CHECK(hipMemcpy(ctx->d_materials.p, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_CHECK(ctx->d_emitters.reserve(ems.size()));
    HIP_CHECK(hipMemcpy(ctx->d_emitters.p, ems.data(), ems.size() * sizeof(float4), hipMemcpyHostToDevice));
    ctx->scene.em_texels = nullptr; ctx->scene.em_w = ctx->scene.em_h = 0;
    if (s->envmap) {  // EnvironmentMap::configure (envmap.cpp:255-322): the same float running sums as the oracle's
        const ppg_envmap &em = *s->envmap;
        if (s->environment) { ctx->error = "envmap: a scene has one environment emitter (`environment` is set as well)"; return PPG_ERR_INVALID; }
        if (!em.rgb || em.width == 0 || em.height == 0 || em.width > 0xFFFF || em.height > 0xFFFF) { ctx->error = "envmap: needs pixels and 0 < width, height < 65536"; return PPG_ERR_INVALID; }
        const int w = (int)em.width, h = (int)em.height;
        std::vector<float4> tex((size_t)w * h);
        std::vector<float> cdfCols((size_t)(w + 1) * h, 0.0f), cdfRows(h + 1, 0.0f), rowWeights(h, 0.0f);
        size_t colPos = 0, rowPos = 0;
        float rowSum = 0.0f;
        cdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            float colSum = 0;
            cdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) {
                const float *px = em.rgb + 3 * ((size_t)y * w + x);
                tex[(size_t)y * w + x] = make_float4(px[0], px[1], px[2], 0.0f);
                colSum += px[0] * 0.212671f + px[1] * 0.715160f + px[2] * 0.072169f;
                cdfCols[colPos++] = colSum;
            }
            const float normalization = 1.0f / colSum;
            for (int x = 1; x < w; ++x) cdfCols[colPos - x - 1] *= normalization;
            cdfCols[colPos - 1] = 1.0f;
            float weight, cosUnused;
            ppg_sincos((y + 0.5f) * PPG_PI_F / h, &weight, &cosUnused);
            rowWeights[y] = weight;
            rowSum += colSum * weight;
            cdfRows[rowPos++] = rowSum;
        }
        const float norm = 1.0f / rowSum;
        for (int y = 1; y < h; ++y) cdfRows[rowPos - y - 1] *= norm;
        cdfRows[rowPos - 1] = 1.0f;
        if (!(rowSum > 0) || !std::isfinite(rowSum)) { ctx->error = "envmap: the environment map is completely black or holds nan / inf (envmap.cpp:308-312)"; return PPG_ERR_INVALID; }
        HIP_CHECK(ctx->d_emTexels.reserve(tex.size())); HIP_CHECK(hipMemcpy(ctx->d_emTexels.p, tex.data(), tex.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emCdfRows.reserve(cdfRows.size())); HIP_CHECK(hipMemcpy(ctx->d_emCdfRows.p, cdfRows.data(), cdfRows.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emCdfCols.reserve(cdfCols.size())); HIP_CHECK(hipMemcpy(ctx->d_emCdfCols.p, cdfCols.data(), cdfCols.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emRowWeights.reserve(rowWeights.size())); HIP_CHECK(hipMemcpy(ctx->d_emRowWeights.p, rowWeights.data(), rowWeights.size() * 4, hipMemcpyHostToDevice));
        DevScene &E = ctx->scene;
        E.em_texels = ctx->d_emTexels.p; E.em_cdf_rows = ctx->d_emCdfRows.p; E.em_cdf_cols = ctx->d_emCdfCols.p; E.em_row_weights = ctx->d_emRowWeights.p;
        E.em_w = w; E.em_h = h; E.em_scale = em.scale;
        E.em_norm = 1.0f / (rowSum * (2 * PPG_PI_F / w) * (PPG_PI_F / h));
        E.em_px = 2 * PPG_PI_F / w; E.em_py = PPG_PI_F / h;
        memcpy(E.em_R, em.to_world, sizeof E.em_R);
    }
    {   // luminaire sampling tables: TriMesh::prepareSamplingTable (trimesh.cpp:388-403) per emitter (= the triangles
        // carrying its id, in index order) and Scene::configure's emitter pmf (scene.cpp:375-380, samplingWeight = 1);
        // float running sums and DiscreteDistribution::normalize() (pmf.h:101-114) exactly as the oracle builds them
        const uint32_t ne = s->n_emitters;
        std::vector<std::vector<uint32_t>> emTris(ne);
        for (uint32_t t = 0; t < s->n_triangles; ++t) if (s->tri_emitter[t] >= 0) emTris[s->tri_emitter[t]].push_back(t);
        std::vector<float> selCdf(1, 0.0f), areaCdf;
        std::vector<int4> info(std::max<uint32_t>(1, ne));
        std::vector<float4> etris, enrm;
        auto normalize = [](std::vector<float> &cdf, size_t first, size_t entries, float &sum) {
            sum = cdf[first + entries - 1];
            if (sum > 0) {
                float normalization = 1.0f / sum;
                for (size_t i = 1; i < entries; ++i) cdf[first + i] *= normalization;
                cdf[first + entries - 1] = 1.0f;
                return normalization;
            }
            return 0.0f;
        };
        for (uint32_t e = 0; e < ne; ++e) {
            const size_t first = areaCdf.size();
            areaCdf.push_back(0.0f);
            const int firstTri = (int)(etris.size() / 3);
            for (uint32_t t : emTris[e]) {
                const float *p0 = s->positions + 3 * s->indices[3 * t], *p1 = s->positions + 3 * s->indices[3 * t + 1], *p2 = s->positions + 3 * s->indices[3 * t + 2];
                const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
                const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
                const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
                areaCdf.push_back(areaCdf.back() + 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz));  // Triangle::surfaceArea, triangle.cpp:61-67
                for (int v = 0; v < 3; ++v) {
                    const float *p = s->positions + 3 * s->indices[3 * t + v];
                    etris.push_back(make_float4(p[0], p[1], p[2], 0));
                    if (s->normals) { const float *n = s->normals + 3 * s->indices[3 * t + v]; enrm.push_back(make_float4(n[0], n[1], n[2], 0)); }
                }
            }
            float area = 0, inv = -1;
            if (!emTris[e].empty()) { normalize(areaCdf, first, emTris[e].size() + 1, area); inv = 1.0f / area; }
            info[e] = make_int4(firstTri, (int)emTris[e].size(), (int)first, __builtin_bit_cast(int, inv));
            if (emitterSphere[e] >= 0) info[e].y = -(emitterSphere[e] + 1);  // sampled analytically (sphere_sample_direct)
            selCdf.push_back(selCdf.back() + 1.0f);
        }
        if (s->environment || s->envmap) selCdf.push_back(selCdf.back() + 1.0f);  // the environment emitter is the last one
        float selSum = 0, selNorm = 0;
        if (selCdf.size() > 1) selNorm = normalize(selCdf, 0, selCdf.size(), selSum);
        if (etris.empty()) etris.push_back(make_float4(0, 0, 0, 0));
        if (areaCdf.empty()) areaCdf.push_back(0.0f);
        HIP_CHECK(ctx->d_emSel.reserve(selCdf.size())); HIP_CHECK(hipMemcpy(ctx->d_emSel.p, selCdf.data(), selCdf.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emArea.reserve(areaCdf.size())); HIP_CHECK(hipMemcpy(ctx->d_emArea.p, areaCdf.data(), areaCdf.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emInfo.reserve(info.size())); HIP_CHECK(hipMemcpy(ctx->d_emInfo.p, info.data(), info.size() * sizeof(int4), hipMemcpyHostToDevice));
        HIP_CHECK(ctx->d_emTris.reserve(etris.size())); HIP_CHECK(hipMemcpy(ctx->d_emTris.p, etris.data(), etris.size() * sizeof(float4), hipMemcpyHostToDevice));
        if (!enrm.empty()) { HIP_CHECK(ctx->d_emNrm.reserve(enrm.size())); HIP_CHECK(hipMemcpy(ctx->d_emNrm.p, enrm.data(), enrm.size() * sizeof(float4), hipMemcpyHostToDevice)); }
        DevScene &S = ctx->scene;
        if (s->environment || s->envmap) {  // Constant / EnvironmentMap: bounding sphere of createShape() (constant.cpp:67-78, envmap.cpp:330-355) around Scene::getAABB()
            float c[3], r2 = 0;
            for (int a = 0; a < 3; ++a) { c[a] = (ctx->aabbMax[a] + ctx->aabbMin[a]) * 0.5f; }
            const float dx = c[0] - ctx->aabbMax[0], dy = c[1] - ctx->aabbMax[1], dz = c[2] - ctx->aabbMax[2];
            r2 = dx * dx + dy * dy + dz * dz;
            if (s->environment) S.env = make_float4(s->environment[0], s->environment[1], s->environment[2], 1.0f);
            else S.env = make_float4(0, 0, 0, 2.0f);  // image based: DevScene::em_*
            S.bsphere = make_float4(c[0], c[1], c[2], ppg_max(PPG_EPSILON, std::sqrt(r2) * 1.5f));
            ctx->fullMaterials = true;  // the environment code lives in the FULL kernel variants
        } else {
            S.env = make_float4(0, 0, 0, 0); S.bsphere = make_float4(0, 0, 0, 0);
        }
        S.n_emitters = (int)ne; S.em_sel_cdf = ctx->d_emSel.p; S.em_sel_norm = selNorm; S.em_info = ctx->d_emInfo.p;
        S.em_area_cdf = ctx->d_emArea.p; S.em_tris = ctx->d_emTris.p; S.em_normals = enrm.empty() ? nullptr : ctx->d_emNrm.p;
    }
    DevScene &S = ctx->scene;
    S.tris = ctx->d_tris.p; S.accel = ctx->d_accel.p; S.normals = s->normals ? ctx->d_normals.p : nullptr; S.bvh = ctx->d_bvh.p; S.bvh4 = ctx->d_bvh4.p; S.bvh_top = ctx->d_bvhTop.p; S.n_top = nTop;
    S.materials = ctx->d_materials.p; S.emitters = ctx->d_emitters.p; S.n_tris = (int)s->n_triangles; S.has_null = hasNull ? 1 : 0;
    S.rtrans = s->n_rtrans ? ctx->d_rtrans.p : nullptr; S.rtrans_n = (int)s->rtrans_samples;
    S.spheres = nullptr; S.n_spheres = (int)s->n_spheres;
    if (s->n_spheres) {
        std::vector<float4> sph(4 * (size_t)s->n_spheres);
        for (uint32_t k = 0; k < s->n_spheres; ++k) {
            const ppg_sphere &sp = s->spheres[k];
            sph[4 * k + 0] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius);
            sph[4 * k + 1] = make_float4(sp.to_world[0], sp.to_world[1], sp.to_world[2], __builtin_bit_cast(float, (int)sp.material));
            sph[4 * k + 2] = make_float4(sp.to_world[3], sp.to_world[4], sp.to_world[5], __builtin_bit_cast(float, sp.emitter));
            sph[4 * k + 3] = make_float4(sp.to_world[6], sp.to_world[7], sp.to_world[8], __builtin_bit_cast(float, sp.flip_normals ? 1 : 0));
        }
        HIP_CHECK(ctx->d_spheres.reserve(sph.size()));
        HIP_CHECK(hipMemcpy(ctx->d_spheres.p, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice));
        S.spheres = ctx->d_spheres.p;
        ctx->fullMaterials = true;  // the sphere code lives in the FULL kernel variants, on the BVH path
    }
    memcpy(S.cam.s2c, s->camera.sample_to_camera, 64); memcpy(S.cam.c2w, s->camera.camera_to_world, 64);
    S.cam.near_clip = s->camera.near_clip; S.cam.far_clip = s->camera.far_clip;
    S.cam.width = s->camera.width; S.cam.height = s->camera.height;
    S.cam.inv_w = 1.0f / (float)s->camera.width; S.cam.inv_h = 1.0f / (float)s->camera.height;
    ctx->W = s->camera.width; ctx->H = s->camera.height;
    {   // LDS budget of k_trace: 32 KB keeps 5 workgroups per CU resident
        const size_t budget = 32 * 1024;
        ctx->ldsNodes = (int)std::min<size_t>(bb.nodes.size(), budget / 64);
        size_t left = budget - (size_t)ctx->ldsNodes * 64;
        ctx->ldsTris = ((size_t)s->n_triangles * 48 <= left) ? (int)s->n_triangles : 0;
    }
    // Every array the path kernels walk must be there before a launch can chase it: an unset pointer here is a hung GPU, not a wrong pixel
    // (round 5 lost 40 GPU minutes to five assignments that an edit had turned into a comment).
    if (!S.materials || !S.bvh4 || (S.n_tris > 0 && (!S.tris || !S.accel || !S.bvh)) || (S.n_spheres > 0 && !S.spheres)) {
        ctx->error = "internal: device scene incomplete";
        return PPG_ERR_STATE;
    }
    ctx->haveScene = true;
    ctx->treeAlive = false;
    ctx->pathsReady = false;
    {   // scene preprocessing also sizes the per-pass buffers (path state, vertex slots, queues, film)
        int rc = allocPaths(ctx);
        if (rc) return rc;
        if ((rc = allocFilm(ctx))) return rc;
        if ((rc = presizeRounds(ctx))) return rc;
        ctx->pathsReady = true;
    }
    return PPG_OK;
}

int ppg_set_shard(ppg_ctx *ctx, int32_t rank, int32_t world, int32_t tile_size) {
    (void)hipSetDevice(ctx->device);
    ctx->quiesce();
    if (world < 1 || rank < 0 || rank >= world || tile_size < 1) { ctx->error = "bad shard"; return PPG_ERR_INVALID; }
    ctx->shardRank = rank; ctx->shardWorld = world; ctx->tileSize = tile_size;
    ctx->pathsReady = false;
    if (ctx->haveScene) {
        int rc = allocPaths(ctx);
        if (rc) return rc;
        if ((rc = allocFilm(ctx))) return rc;
        if ((rc = presizeRounds(ctx))) return rc;
        ctx->pathsReady = true;
    }
    return PPG_OK;
}

// every entry point re-selects the context's device: the caller (torch / RCCL in a multi-GPU process) may have changed it
// ... and names the context's stream as the one buffer growth is ordered on (StreamScope)
#define NEED_SCENE (void)hipSetDevice(ctx->device); StreamScope scope_(ctx->stream); if (!ctx->haveScene) { ctx->error = "no scene"; return PPG_ERR_STATE; }
#define NEED_TREE (void)hipSetDevice(ctx->device); StreamScope scope_(ctx->stream); if (!ctx->treeAlive) { ctx->error = "render not begun"; return PPG_ERR_STATE; } ctx->joinTree();

int ppg_begin_render(ppg_ctx *ctx) { NEED_SCENE return beginRender(ctx); }
int ppg_begin_iteration(ppg_ctx *ctx, int32_t is_final) { NEED_TREE return beginIteration(ctx, is_final != 0); }
int ppg_set_final(ppg_ctx *ctx, int32_t is_final) { ctx->isFinalIter = is_final != 0; return PPG_OK; }
int ppg_set_do_nee(ppg_ctx *ctx, int32_t do_nee) { ctx->doNee = do_nee != 0; return PPG_OK; }
int ppg_render_passes_nostat(ppg_ctx *ctx, int32_t n) { NEED_TREE return renderPassesNoStat(ctx, n); }
int ppg_finish_passes(ppg_ctx *ctx, ppg_pass_stats *st) { NEED_TREE return finishPasses(ctx, st); }
int ppg_render_passes(ppg_ctx *ctx, int32_t n, ppg_pass_stats *st) {
    NEED_TREE
    int rc = renderPassesNoStat(ctx, n);
    if (rc && rc != PPG_ERR_CANCELLED) return rc;
    int rc2 = finishPasses(ctx, st);
    return rc ? rc : rc2;
}
int ppg_build_sdtree(ppg_ctx *ctx, ppg_tree_stats *st) { NEED_TREE return buildSDTree(ctx, st); }
int ppg_end_iteration(ppg_ctx *ctx) { NEED_TREE return endIteration(ctx); }
int ppg_end_render(ppg_ctx *ctx) { NEED_TREE return endRender(ctx); }
int ppg_cancel(ppg_ctx *ctx) { ctx->cancelled.store(true); return PPG_OK; }
int ppg_debug_build_bvh(const float *positions, const uint32_t *indices, uint32_t n_triangles, float pad_abs, int32_t max_leaf,
                        void *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes, uint32_t *order_out) {
    if (!positions || !indices || !n_nodes || n_triangles == 0) return PPG_ERR_INVALID;
    BvhBuilder bb;
    bb.maxLeaf = std::max(1, std::min(8, (int)max_leaf));
    bb.run(positions, indices, n_triangles, pad_abs);
    *n_nodes = (uint32_t)bb.nodes4q.size();
    if (nodes_out) memcpy(nodes_out, bb.nodes4q.data(), std::min<size_t>(nodes_cap, bb.nodes4q.size()) * sizeof(Bvh4QNode));
    if (order_out) memcpy(order_out, bb.order.data(), (size_t)n_triangles * sizeof(uint32_t));
    return PPG_OK;
}
int ppg_release_cached_memory(int32_t device) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return PPG_ERR_DEVICE;
    g_blockCache.trim();
    (void)hipSetDevice(prev);
    return PPG_OK;
}

int ppg_render(ppg_ctx *ctx) {  // GP:1516-1585
    NEED_SCENE
    int rc = beginRender(ctx);
    if (rc) return rc;
    rc = ctx->budgetType == 0 ? renderSPP(ctx) : renderTime(ctx);
    if (rc) return rc;
    return endRender(ctx);
}

int ppg_read_film(ppg_ctx *ctx, float *rgb) {
    NEED_SCENE
    const int n = ctx->W * ctx->H;
    if (!ctx->d_film.p || !ctx->d_tmp.p) { ctx->error = "nothing rendered"; return PPG_ERR_STATE; }
    hipLaunchKernelGGL(k_normalise, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, ctx->d_film.p, ctx->d_filmW.p, ctx->d_tmp.p);
    HIP_CHECK(hipMemcpyAsync(rgb, ctx->d_tmp.p, 3 * (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PPG_OK;
}
int ppg_read_variance(ppg_ctx *ctx, float *rgb) {
    NEED_SCENE
    if (!ctx->d_var.p) { ctx->error = "nothing rendered"; return PPG_ERR_STATE; }
    HIP_CHECK(hipMemcpy(rgb, ctx->d_var.p, 3 * (size_t)ctx->W * ctx->H * 4, hipMemcpyDeviceToHost));
    return PPG_OK;
}
int ppg_dump_sdtree(ppg_ctx *ctx, const char *path) { NEED_TREE return dumpSDTreeFile(ctx, path); }

int ppg_sdtree_info_get(ppg_ctx *ctx, ppg_sdtree_info *info) {
    NEED_TREE
    memset(info, 0, sizeof *info);
    info->n_stree_nodes = (uint32_t)ctx->snodes.size();
    std::vector<LeafHdr> dh(ctx->hdr);
    if (ctx->d_hdr.p && ctx->d_hdr.cap >= dh.size()) HIP_CHECK(hipMemcpy(dh.data(), ctx->d_hdr.p, dh.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ctx->snodes.size(); ++i)
        if (ctx->snodes[i].isLeaf()) { info->n_leaves++; info->n_sampling_nodes += ctx->hdr[i].s_num; info->n_building_nodes += dh[i].b_num; }
    for (int a = 0; a < 3; ++a) { info->aabb_min[a] = ctx->treeMin[a]; info->aabb_max[a] = ctx->treeMax[a]; }
    info->iter = ctx->iter; info->is_built = ctx->isBuilt;
    return PPG_OK;
}

int ppg_sdtree_read_stree(ppg_ctx *ctx, int32_t *axis, uint32_t *children) {
    NEED_TREE
    for (size_t i = 0; i < ctx->snodes.size(); ++i) { axis[i] = ctx->snodes[i].axis; children[2 * i] = ctx->snodes[i].child[0]; children[2 * i + 1] = ctx->snodes[i].child[1]; }
    return PPG_OK;
}

int ppg_sdtree_read_dtree_headers(ppg_ctx *ctx, int32_t which, uint64_t *offset, uint32_t *num_nodes, int32_t *max_depth, float *sum, double *stat_weight) {
    NEED_TREE
    std::vector<LeafHdr> dh(ctx->hdr.size());
    std::vector<unsigned long long> bw(ctx->hdr.size(), 0);
    { int rc = foldWeights(ctx); if (rc) return rc; }
    if (ctx->d_hdr.p) {
        HIP_CHECK(hipMemcpy(dh.data(), ctx->d_hdr.p, dh.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(bw.data(), ctx->d_bweight.p, bw.size() * 8, hipMemcpyDeviceToHost));
    } else dh = ctx->hdr;
    uint64_t off = 0;
    for (size_t i = 0; i < ctx->snodes.size(); ++i) {
        if (!ctx->snodes[i].isLeaf()) { offset[i] = 0; num_nodes[i] = 0; max_depth[i] = 0; sum[i] = 0; stat_weight[i] = 0; continue; }
        const LeafHdr &h = dh[i];
        offset[i] = off;
        if (which == 0) { num_nodes[i] = h.s_num; max_depth[i] = h.s_depth; sum[i] = h.s_sum; stat_weight[i] = h.s_statw; off += h.s_num; }
        else {
            num_nodes[i] = h.b_num; max_depth[i] = h.b_depth; sum[i] = 0.0f;
            stat_weight[i] = (double)bw[i] / 16777216.0;  // the accumulator of the current iteration
            off += h.b_num;
        }
    }
    return PPG_OK;
}

int ppg_sdtree_read_dtree_nodes(ppg_ctx *ctx, int32_t which, float *sums, uint16_t *children, uint64_t *fixed_sums) {
    NEED_TREE
    std::vector<float> s; std::vector<uint16_t> c; std::vector<uint64_t> f;
    int rc = gatherDTrees(ctx, which, s, c, fixed_sums ? &f : nullptr);
    if (rc) return rc;
    memcpy(sums, s.data(), s.size() * 4);
    memcpy(children, c.data(), c.size() * 2);
    if (fixed_sums) memcpy(fixed_sums, f.data(), f.size() * 8);
    return PPG_OK;
}

int ppg_sdtree_read_adam(ppg_ctx *ctx, float *theta) {
    NEED_TREE
    std::vector<LeafHdr> dh(ctx->hdr.size());
    if (ctx->d_hdr.p) HIP_CHECK(hipMemcpy(dh.data(), ctx->d_hdr.p, dh.size() * sizeof(LeafHdr), hipMemcpyDeviceToHost));
    else dh = ctx->hdr;
    for (size_t i = 0; i < dh.size(); ++i) theta[i] = dh[i].theta;
    return PPG_OK;
}

int ppg_sdtree_stat_buffers(ppg_ctx *ctx, void **dev_sums, uint64_t *n_sums, void **dev_weights, uint64_t *n_weights) {
    NEED_TREE
    { int rc = foldWeights(ctx); if (rc) return rc; }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *dev_sums = ctx->d_bacc.p; *n_sums = ctx->nBuildingNodes * 4;
    *dev_weights = ctx->d_bweight.p; *n_weights = ctx->snodes.size();
    return PPG_OK;
}
int ppg_film_buffers(ppg_ctx *ctx, void **dev_rgb_sum, void **dev_weight) {
    NEED_TREE
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *dev_rgb_sum = ctx->d_film.p; *dev_weight = ctx->d_filmW.p;
    return PPG_OK;
}
int ppg_image_buffers(ppg_ctx *ctx, void **dev_image, void **dev_sq_image, void **dev_weight) {
    NEED_TREE
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *dev_image = ctx->d_image.p; *dev_sq_image = ctx->d_sq.p; *dev_weight = ctx->d_imageW.p;
    return PPG_OK;
}

int ppg_set_stop_hook(ppg_ctx *ctx, ppg_stop_hook hook, void *user) { ctx->stopHook = hook; ctx->stopHookUser = user; return PPG_OK; }
int ppg_exchange_stream(ppg_ctx *ctx, void **hip_stream) { *hip_stream = (void *)ctx->stream; return PPG_OK; }
int32_t ppg_final_group_passes(int32_t n_passes) {
    const int32_t n = std::max(1, n_passes);
    return 16 * ((n + 1023) / 1024);
}
int ppg_final_partials(ppg_ctx *ctx, void **dev, uint64_t *n_floats) {
    NEED_TREE
    *dev = nullptr; *n_floats = 0;
    if (!ctx->partialsPending) return PPG_OK;  // not a sharded final iteration: image / squared image / weights are exchanged as usual
    const size_t n = ctx->nPixAll;
    if (!ctx->partialsExported) {  // the film of this iteration so far (tile-sharded training passes of an `automatic` render) travels in the head
        HIP_CHECK(hipMemcpyAsync(ctx->d_partials.p, ctx->d_film.p, 3 * n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(ctx->d_partials.p + 3 * n, ctx->d_filmW.p, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ctx->partialsExported = true;
    }
    *dev = ctx->d_partials.p; *n_floats = 4 * n + (uint64_t)ctx->pendingGroups * 7 * n;
    return PPG_OK;
}
int ppg_final_partials_commit(ppg_ctx *ctx) {
    NEED_TREE
    if (!ctx->partialsPending || !ctx->partialsExported) { ctx->error = "ppg_final_partials_commit: nothing to commit"; return PPG_ERR_STATE; }
    const size_t n = ctx->nPixAll;
    HIP_CHECK(hipMemcpyAsync(ctx->d_film.p, ctx->d_partials.p, 3 * n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(ctx->d_filmW.p, ctx->d_partials.p + 3 * n, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    addGroups(ctx, 0, ctx->pendingGroups);
    HIP_CHECK(hipGetLastError());
    ctx->partialsPending = false; ctx->partialsExported = false;
    return PPG_OK;
}
int ppg_set_pass_hook(ppg_ctx *ctx, ppg_pass_hook hook, void *user) { ctx->passHook = hook; ctx->passHookUser = user; return PPG_OK; }
int ppg_adam_records(ppg_ctx *ctx, void **dev_records, uint64_t *n) {
    NEED_TREE
    if (!ctx->inHook) { ctx->error = "ppg_adam_records: only valid inside the round hook"; return PPG_ERR_STATE; }
    *dev_records = ctx->hookReplaced ? (void *)ctx->d_adamRecs.p : (void *)ctx->d_adamRecsOut.p;
    *n = ctx->hookCount;
    return PPG_OK;
}
int ppg_hook_phase(ppg_ctx *ctx, int32_t *phase) {
    if (!ctx || !phase) return PPG_ERR_INVALID;
    *phase = ctx->hookPhase;
    return PPG_OK;
}
int ppg_adam_records_by_owner(ppg_ctx *ctx, int32_t world, void **dev_records, uint64_t *counts) {
    if (!ctx || !dev_records || !counts || world < 1) return PPG_ERR_INVALID;
    if (!ctx->inHook || ctx->hookPhase != 0) { ctx->error = "ppg_adam_records_by_owner: only valid in phase 0 of the round hook"; return PPG_ERR_STATE; }
    HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned int nNodes = (unsigned int)ctx->snodes.size();
    const unsigned int seg = (nNodes + (unsigned int)world - 1u) / (unsigned int)world;
    const unsigned int n = (unsigned int)ctx->hookCount;  // the valid records, in key order: d_adamKeys[1] / d_adamRecsOut
    HIP_CHECK(ctx->d_ownerBounds.reserve((size_t)world + 1));
    std::vector<unsigned long long> b((size_t)world + 1, 0ull);
    if (n) {
        hipLaunchKernelGGL(k_owner_bounds, dim3(((unsigned int)world + 1u + 63u) / 64u), dim3(64), 0, ctx->stream, ctx->d_adamKeys[1].p, n, seg, (unsigned int)world, ctx->d_ownerBounds.p);
        HIP_CHECK(hipMemcpyAsync(b.data(), ctx->d_ownerBounds.p, b.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    for (int32_t r = 0; r < world; ++r) counts[r] = b[(size_t)r + 1] - b[(size_t)r];
    *dev_records = ctx->d_adamRecsOut.p;
    ctx->ownerMode = true;
    return PPG_OK;
}
int ppg_adam_state(ppg_ctx *ctx, int32_t world, void **dev_state, uint64_t *segment) {
    if (!ctx || !dev_state || !segment || world < 1) return PPG_ERR_INVALID;
    if (!ctx->inHook || ctx->hookPhase != 1) { ctx->error = "ppg_adam_state: only valid in phase 1 of the round hook"; return PPG_ERR_STATE; }
    HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned int nNodes = (unsigned int)ctx->snodes.size();
    const size_t seg = ((size_t)nNodes + (size_t)world - 1) / (size_t)world;
    const size_t words = (size_t)world * seg * 6;
    HIP_CHECK(ctx->d_adamState.reserve(std::max<size_t>(words, 6)));
    HIP_CHECK(hipMemsetAsync(ctx->d_adamState.p, 0, words * 4, ctx->stream));
    hipLaunchKernelGGL((k_adam_state<false>), dim3((nNodes + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->d_hdr.p, nNodes, ctx->d_adamState.p);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *dev_state = ctx->d_adamState.p; *segment = seg;
    return PPG_OK;
}
int ppg_adam_state_commit(ppg_ctx *ctx) {
    if (!ctx) return PPG_ERR_INVALID;
    if (!ctx->inHook || ctx->hookPhase != 1 || !ctx->d_adamState.p) { ctx->error = "ppg_adam_state_commit: call ppg_adam_state first"; return PPG_ERR_STATE; }
    HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned int nNodes = (unsigned int)ctx->snodes.size();
    hipLaunchKernelGGL((k_adam_state<true>), dim3((nNodes + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->d_hdr.p, nNodes, ctx->d_adamState.p);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PPG_OK;
}
int ppg_adam_records_replace(ppg_ctx *ctx, const void *dev_records, uint64_t n) {
    NEED_TREE
    if (!ctx->inHook) { ctx->error = "ppg_adam_records_replace: only valid inside the round hook"; return PPG_ERR_STATE; }
    if (n > 0xfffffff0ull) { ctx->error = "too many Adam records in one round"; return PPG_ERR_NOMEM; }
    if (n && dev_records == (const void *)ctx->d_adamRecs.p) { ctx->error = "ppg_adam_records_replace: source aliases the destination"; return PPG_ERR_INVALID; }
    HIP_CHECK(ctx->d_adamRecs.reserve(std::max<size_t>(1, (size_t)n)));
    if (n) HIP_CHECK(hipMemcpyAsync(ctx->d_adamRecs.p, dev_records, (size_t)n * sizeof(AdamRec), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->hookReplaced = true; ctx->hookCount = n;
    return PPG_OK;
}

int ppg_query_pdf(ppg_ctx *ctx, uint32_t n, const float *positions, const float *dirs, float *pdf_out) {
    NEED_TREE
    if (n == 0) return PPG_OK;
    DevBuf<float> dp, dd, dout;
    HIP_CHECK(dp.reserve(3 * (size_t)n)); HIP_CHECK(dd.reserve(3 * (size_t)n)); HIP_CHECK(dout.reserve(n));
    HIP_CHECK(hipMemcpy(dp.p, positions, 12 * (size_t)n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dd.p, dirs, 12 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_query_pdf, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->devTree(), n, dp.p, dd.p, dout.p);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipMemcpy(pdf_out, dout.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    return PPG_OK;
}
int ppg_query_sample(ppg_ctx *ctx, uint32_t n, const float *positions, uint64_t seed, float *dirs_out) {
    NEED_TREE
    if (n == 0) return PPG_OK;
    DevBuf<float> dp, dout;
    HIP_CHECK(dp.reserve(3 * (size_t)n)); HIP_CHECK(dout.reserve(3 * (size_t)n));
    HIP_CHECK(hipMemcpy(dp.p, positions, 12 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_query_sample, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->devTree(), n, dp.p, (unsigned long long)seed, dout.p);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipMemcpy(dirs_out, dout.p, 12 * (size_t)n, hipMemcpyDeviceToHost));
    return PPG_OK;
}

int ppg_set_adam_regions(ppg_ctx *ctx, int32_t regions) {  // include/ppg.h "Rounds by image region"
    if (!ctx || regions < 0 || regions > 4096) { if (ctx) ctx->error = "ppg_set_adam_regions: 0 .. 4096"; return PPG_ERR_INVALID; }
    ctx->adamRegions = regions >= 2 ? regions : 0;
    return PPG_OK;
}

int ppg_debug_set_defer_depth(ppg_ctx *ctx, int32_t depth) {  // include/ppg_testhooks.h
    if (!ctx || depth < 1 || depth > 64) return PPG_ERR_INVALID;
    ctx->deferDepthAdam = depth;
    return PPG_OK;
}

int ppg_enable_kernel_timing(ppg_ctx *ctx, int32_t enable) {
    ctx->timer.reset();
    ctx->bvhNodesVisited = ctx->bvhTrisTested = 0; ctx->tailLongestSum = 0; ctx->shadeCommonRays = 0;
    ctx->timer.enabled = enable != 0;
    return PPG_OK;
}
int ppg_kernel_times(ppg_ctx *ctx, ppg_kernel_time *out, uint32_t cap, uint32_t *n) {
    ctx->timer.resolve();
    uint32_t k = 0;
    for (size_t i = 0; i < ctx->timer.names.size() && k < cap; ++i, ++k) {
        out[k].name = ctx->timer.names[i].c_str(); out[k].ms = ctx->timer.ms[i]; out[k].launches = ctx->timer.launches[i]; out[k].units = ctx->timer.units[i];
        // the two launches over a sorted slice were both booked with the slice's rays: the common classes' share was counted by k_sort_slices
        if (ctx->timer.names[i] == "k_shade<common>") out[k].units = ctx->shadeCommonRays;
        else if (ctx->timer.names[i] == "k_shade<rest>") out[k].units = ctx->timer.units[i] > ctx->shadeCommonRays ? ctx->timer.units[i] - ctx->shadeCommonRays : 0;
    }
    // two pseudo entries for the roofline of k_trace on BVH scenes: `units` = BVH4 nodes visited / triangles tested by its launches
    if (k + 2 <= cap && ctx->bvhNodesVisited) {
        out[k].name = "bvh_nodes_visited"; out[k].ms = 0; out[k].launches = 0; out[k].units = ctx->bvhNodesVisited; ++k;
        out[k].name = "bvh_triangles_tested"; out[k].ms = 0; out[k].launches = 0; out[k].units = ctx->bvhTrisTested; ++k;
    }
    // ... and one for k_tail: `units` = sum over its launches of the longest path (bounces) each finished — a launch of k_tail cannot be
    // shorter than its longest path's chain of dependent bounces
    if (k + 1 <= cap && ctx->tailLongestSum) { out[k].name = "tail_longest_paths_sum"; out[k].ms = 0; out[k].launches = 0; out[k].units = ctx->tailLongestSum; ++k; }
    *n = k;
    return PPG_OK;
}

}  // extern "C"
