/*
 * guided_path_hip.h — C++ host side above the C-ABI: the integrator object a C++ application holds instead of
 * mitsuba's GuidedPathTracer (guided_path.cpp:1012-2419, cited GP:line).
 *
 *   GuidedPathTracerHIP gpt(props);          // props: the reference's property names (GP:1014-1085)
 *   gpt.render(scene, log);                  // render() — drives renderSPP / renderTime phase by phase and
 *                                            //            prints the reference's log lines (GP:1176-1186, 1325, 1376)
 *   gpt.cancel();                            // from another thread (GP:1643-1648)
 *   gpt.film();                              // weight-normalised RGB
 *
 * Everything heavy happens behind include/ppg.h in libppg_hip.so; this file only sequences the phases.
 */
#ifndef GUIDED_PATH_HIP_H
#define GUIDED_PATH_HIP_H

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ppg.h"

namespace ppg {

// Flat scene owned by the host (what ppg_set_scene copies from)
struct SceneData {
    std::vector<float> positions, normals;
    std::vector<uint32_t> indices, triMaterial;
    std::vector<int32_t> triEmitter;
    std::vector<ppg_material> materials;
    std::vector<ppg_emitter> emitters;
    ppg_camera camera{};
    bool hasEnvironment = false;  // constant environment emitter
    float environment[3] = {0, 0, 0};
    std::vector<float> rtrans;    // rough-transmittance slices of the roughplastic materials, rtransSamples + 1 floats each
    uint32_t rtransSamples = 0;
    std::vector<ppg_sphere> spheres;  // analytic spheres
    bool hasEnvmap = false;           // image-based environment emitter: envmap (pixels in envmapRgb)
    mutable ppg_envmap envmap{};
    std::vector<float> envmapRgb;
    std::vector<float> texcoords;                  // per-vertex texture coordinates (NaN rows: none) or empty
    mutable std::vector<ppg_texture> textures;     // bitmap textures; their pixels in texturePixels
    std::vector<std::vector<float>> texturePixels;

    ppg_scene view() const {
        ppg_scene s{};
        s.n_vertices = (uint32_t)(positions.size() / 3); s.positions = positions.data();
        s.normals = normals.empty() ? nullptr : normals.data();
        s.n_triangles = (uint32_t)(indices.size() / 3); s.indices = indices.data();
        s.tri_material = triMaterial.data(); s.tri_emitter = triEmitter.data();
        s.n_materials = (uint32_t)materials.size(); s.materials = materials.data();
        s.n_emitters = (uint32_t)emitters.size(); s.emitters = emitters.data();
        s.camera = camera;
        s.environment = hasEnvironment ? environment : nullptr;
        if (hasEnvmap) { envmap.rgb = envmapRgb.data(); s.envmap = &envmap; }
        if (!spheres.empty()) { s.n_spheres = (uint32_t)spheres.size(); s.spheres = spheres.data(); }
        if (rtransSamples && !rtrans.empty()) { s.n_rtrans = (uint32_t)(rtrans.size() / (rtransSamples + 1)); s.rtrans_samples = rtransSamples; s.rtrans = rtrans.data(); }
        s.texcoords = texcoords.empty() ? nullptr : texcoords.data();
        for (size_t i = 0; i < textures.size(); ++i) textures[i].rgb = texturePixels[i].data();
        if (!textures.empty()) { s.n_textures = (uint32_t)textures.size(); s.textures = textures.data(); }
        return s;
    }
};

// Properties: string map with the typed getters of mitsuba/core/properties.h as far as this plugin uses them
class Properties {
public:
    std::map<std::string, std::string> values;
    std::string getString(const std::string &k, const std::string &d) const { auto it = values.find(k); return it == values.end() ? d : it->second; }
    int getInteger(const std::string &k, int d) const { auto it = values.find(k); return it == values.end() ? d : std::stoi(it->second); }
    float getFloat(const std::string &k, float d) const { auto it = values.find(k); return it == values.end() ? d : std::stof(it->second); }
    bool getBoolean(const std::string &k, bool d) const {
        auto it = values.find(k);
        if (it == values.end()) return d;
        return it->second == "true" || it->second == "1";
    }
};

// The exchange points of a tile-sharded multi-GPU render (one process per GPU): what render() calls between the phases of an iteration.
// host/rccl_reducer.h implements it over RCCL.
class Reducer {
public:
    virtual ~Reducer() {}
    virtual void reduceImages(ppg_ctx *ctx, int width, int height) = 0;   // before the variance estimate of performRenderPasses (GP:1288-1313)
    virtual void reduceSDTree(ppg_ctx *ctx) = 0;                          // before buildSDTree (GP:1115)
    virtual void reduceAdamRecords(ppg_ctx *ctx) = 0;                     // round hook of the sampling-fraction optimiser (include/ppg.h)
    virtual void reduceFilm(ppg_ctx *ctx, int width, int height) = 0;     // before the film is read (not with inverse-variance combination)
    // a final iteration's groups of passes (include/ppg.h "Final iteration: groups of passes"): all-reduce the n floats at dev, then ppg_final_partials_commit
    virtual void reduceFinalPartials(ppg_ctx *ctx, void *dev, uint64_t nFloats) = 0;
    virtual double broadcast(double value) = 0;                            // rank 0's value on every rank (the clock readings of a time budget)
    // the stop decision of a time budget (include/ppg.h ppg_set_stop_hook): rank 0's `localStop`, or "stop" if any rank's status word is set —
    // a cancelled rank meets the others HERE, and they all leave the batch loop for the image exchange, where the status aborts the render
    virtual int stopDecision(int localStop) = 0;
    virtual void beginRender() {}                                         // a new render() starts: forget the status of the previous one
    virtual void setLocalStatus(int status) { (void)status; }             // != 0: this rank was cancelled / failed — announced to the others in the next exchange
    virtual int rank() const = 0;
    virtual int world() const = 0;
};

class GuidedPathTracerHIP {
public:
    typedef std::function<void(const std::string &)> LogFn;  // (not "Log": that is a macro of Mitsuba's logger.h, and the plug-in includes this header)

    explicit GuidedPathTracerHIP(const Properties &props) {  // GP:1014-1085 + MonteCarloIntegrator (integrator.cpp:190-225)
        ppg_config_default(&m_cfg);
        m_str[0] = props.getString("nee", "never"); m_cfg.nee = m_str[0].c_str();
        m_str[1] = props.getString("sampleCombination", "automatic"); m_cfg.sampleCombination = m_str[1].c_str();
        m_str[2] = props.getString("spatialFilter", "nearest"); m_cfg.spatialFilter = m_str[2].c_str();
        m_str[3] = props.getString("directionalFilter", "nearest"); m_cfg.directionalFilter = m_str[3].c_str();
        m_str[4] = props.getString("bsdfSamplingFractionLoss", "none"); m_cfg.bsdfSamplingFractionLoss = m_str[4].c_str();
        m_str[5] = props.getString("budgetType", "seconds"); m_cfg.budgetType = m_str[5].c_str();
        m_str[6] = props.getString("dumpPrefix", ""); m_cfg.dumpPrefix = m_str[6].empty() ? nullptr : m_str[6].c_str();
        m_cfg.sdTreeMaxMemory = props.getInteger("sdTreeMaxMemory", -1);
        m_cfg.sTreeThreshold = props.getInteger("sTreeThreshold", 12000);
        m_cfg.dTreeThreshold = props.getFloat("dTreeThreshold", 0.01f);
        m_cfg.bsdfSamplingFraction = props.getFloat("bsdfSamplingFraction", 0.5f);
        m_cfg.sppPerPass = props.getInteger("sppPerPass", 4);
        m_cfg.budget = props.getFloat("budget", 300.0f);
        m_cfg.dumpSDTree = props.getBoolean("dumpSDTree", false);
        m_cfg.rrDepth = props.getInteger("rrDepth", 5);
        m_cfg.maxDepth = props.getInteger("maxDepth", -1);
        m_cfg.strictNormals = props.getBoolean("strictNormals", false);
        m_cfg.hideEmitters = props.getBoolean("hideEmitters", false);
        m_cfg.seed = (uint64_t)std::stoull(props.getString("seed", "0"));
        m_cfg.device = props.getInteger("device", 0);
        if (ppg_create(&m_cfg, &m_ctx) != PPG_OK) throw std::runtime_error(std::string("ppg_create: ") + ppg_last_error(nullptr));
        // not a property of the reference: rounds by image region in the early iterations (include/ppg.h ppg_set_adam_regions), 0 = off
        const int regions = props.getInteger("adamRegions", 0);
        if (regions && ppg_set_adam_regions(m_ctx, regions) != PPG_OK) throw std::runtime_error(std::string("adamRegions: ") + ppg_last_error(m_ctx));
    }
    ~GuidedPathTracerHIP() { ppg_destroy(m_ctx); }
    GuidedPathTracerHIP(const GuidedPathTracerHIP &) = delete;
    GuidedPathTracerHIP &operator=(const GuidedPathTracerHIP &) = delete;

    void cancel() { m_cancelled.store(true); ppg_cancel(m_ctx); }  // GP:1643-1648

    // render(): GP:1516-1585.  Returns false when cancelled, throws on errors.  With a reducer the image is sharded by 32x32 tiles over
    // reducer->world() ranks (every rank calls render() on the same scene) and the film is complete on every rank afterwards.
    bool render(const SceneData &scene, const LogFn &log = LogFn(), Reducer *reducer = nullptr) {
        m_reducer = reducer;
        // (cancel() is sticky in the library: one that arrived before this call — or during ppg_set_scene below — cancels this render,
        // ppg_begin_render consumes it; m_cancelled, which the hooks read, is reset when the render is over)
        struct Reset { std::atomic<bool> &f; ~Reset() { f.store(false); } } reset{m_cancelled};
        m_filmComplete = false;
        if (reducer) reducer->beginRender();
        const bool spp = std::string(m_cfg.budgetType) == "spp";
        ppg_scene sv = scene.view();
        check(ppg_set_scene(m_ctx, &sv), "ppg_set_scene");
        if (reducer) check(ppg_set_shard(m_ctx, reducer->rank(), reducer->world(), 32), "ppg_set_shard");
        m_w = scene.camera.width; m_h = scene.camera.height;
        {
            const int rc = ppg_begin_render(m_ctx);
            if (reducer) {
                // Sharded: a cancel() that arrived before this call cancels THIS rank's render here, before any exchange — the peers would wait for
                // it in their first collective.  So every sharded render starts with one status exchange (the all-reduce of stopDecision with
                // "no stop" from every rank): a rank that cannot start says so, and all of them leave together.
                if (rc != PPG_OK) reducer->setLocalStatus(1);
                const int bad = reducer->stopDecision(0);
                if (rc == PPG_ERR_CANCELLED) return false;  // GP:1584
                check(rc, "ppg_begin_render");
                if (bad) throw std::runtime_error("render aborted: a rank could not start (cancelled or failed before its first exchange)");
            } else {
                if (rc == PPG_ERR_CANCELLED) return false;  // GP:1584
                check(rc, "ppg_begin_render");
            }
        }
        if (reducer && std::string(m_cfg.bsdfSamplingFractionLoss) != "none") check(ppg_set_pass_hook(m_ctx, &GuidedPathTracerHIP::roundHook, this), "ppg_set_pass_hook");
        // a time budget, sharded: every decision taken by a clock (GP:1259-1262, 1434-1514) is rank 0's, so that all ranks render the same passes
        check(ppg_set_stop_hook(m_ctx, (reducer && !spp) ? &GuidedPathTracerHIP::stopHook : nullptr, this), "ppg_set_stop_hook");
        say(log, fmt("Starting render job (%ix%i, MI355X%s) ..", m_w, m_h, reducer ? fmt(", rank %d of %d", reducer->rank(), reducer->world()).c_str() : ""));
        bool ok = spp ? renderSPP(log) : renderTime(log);
        if (ok && reducer && !m_filmComplete && std::string(m_cfg.sampleCombination) != "inversevar") reducer->reduceFilm(m_ctx, m_w, m_h);
        if (ok) check(ppg_end_render(m_ctx), "ppg_end_render");
        return ok;
    }

    std::vector<float> film() {
        std::vector<float> rgb((size_t)m_w * m_h * 3);
        check(ppg_read_film(m_ctx, rgb.data()), "ppg_read_film");
        return rgb;
    }
    ppg_ctx *context() { return m_ctx; }

private:
    static std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2))) {
        char buf[1024];
        va_list ap; va_start(ap, f); vsnprintf(buf, sizeof buf, f, ap); va_end(ap);
        return buf;
    }
    static void say(const LogFn &log, const std::string &s) { if (log) log(s); }
    void check(int rc, const char *what) {
        rethrowHookError();
        if (rc != PPG_OK && rc != PPG_ERR_CANCELLED) throw std::runtime_error(std::string(what) + ": " + ppg_last_error(m_ctx));
    }

    // performRenderPasses, GP:1210-1329
    bool passes(int n, ppg_pass_stats &st, const LogFn &log, bool finalGroups = false) {
        say(log, fmt("Rendering %d render passes.", n));
        int rc;
        if (!m_reducer) {
            rc = ppg_render_passes(m_ctx, n, &st);
            check(rc, "ppg_render_passes");
        } else {  // the ranks' image tiles are summed before the variance is estimated from them
            rc = ppg_render_passes_nostat(m_ctx, n);
            // cancelled or failed here: the other ranks must not wait for this one in their next collective — tell them in it
            if (rc != PPG_OK || m_hookError) m_reducer->setLocalStatus(1);
            try {
                // Which exchange follows is decided by what EVERY rank knows — final flag, budget type, pass count — never by this rank's
                // own outcome: a cancelled or failed rank still joins the collective the others are in (with its status word set).
                if (finalGroups && m_reducer->world() > 1) {
                    // a final iteration rendered in whole groups of passes by rank: their partial images instead of the tiles' image buffers
                    const uint64_t px = (uint64_t)m_w * m_h, G = (uint64_t)ppg_final_group_passes(n);
                    const uint64_t expect = 4 * px + (((uint64_t)n + G - 1) / G) * 7 * px;
                    void *dev = nullptr;
                    uint64_t nFloats = 0;
                    if (ppg_final_partials(m_ctx, &dev, &nFloats) != PPG_OK || nFloats != expect) { m_reducer->setLocalStatus(1); dev = nullptr; }
                    m_reducer->reduceFinalPartials(m_ctx, dev, expect);
                    m_filmComplete = true;
                } else m_reducer->reduceImages(m_ctx, m_w, m_h);
            } catch (...) { if (!m_hookError) m_hookError = std::current_exception(); }
            if (rc == PPG_ERR_CANCELLED || m_cancelled.load()) { m_hookError = nullptr; return false; }  // (the exchanges told the others; they leave with an error)
            check(rc, "ppg_render_passes_nostat");
            rethrowHookError();
            check(ppg_finish_passes(m_ctx, &st), "ppg_finish_passes");
        }
        const float ttuv = (float)st.seconds * st.variance, stuv = st.passes_rendered_local * m_cfg.sppPerPass * st.variance;
        say(log, fmt("%.2f seconds, Total passes: %d, Var: %f, TTUV: %f, STUV: %f.", st.seconds, st.passes_rendered_total, st.variance, ttuv, stuv));
        m_passesRendered = st.passes_rendered_total;
        return rc == PPG_OK;
    }
    void build(const LogFn &log, bool nothingRecorded = false) {  // buildSDTree, GP:1115-1189
        if (m_reducer && !nothingRecorded) m_reducer->reduceSDTree(m_ctx);  // (an iteration that was final from its first pass records nothing)
        say(log, "Building distributions for sampling.");
        ppg_tree_stats t;
        check(ppg_build_sdtree(m_ctx, &t), "ppg_build_sdtree");
        say(log, fmt("Distribution statistics:\n  Depth         = [%d, %f, %d]\n  Mean radiance = [%f, %f, %f]\n  Node count    = [%llu, %f, %llu]\n"
                     "  Stat. weight  = [%f, %f, %f]\n",
                     t.min_depth, t.avg_depth, t.max_depth, t.min_mean_radiance, t.avg_mean_radiance, t.max_mean_radiance,
                     (unsigned long long)t.min_nodes, t.avg_nodes, (unsigned long long)t.max_nodes, t.min_stat_weight, t.avg_stat_weight,
                     t.max_stat_weight));
    }
    bool doNeeWithSpp(int spp) const {  // GP:1331-1340
        const std::string nee = m_cfg.nee;
        return nee == "never" ? false : (nee == "kickstart" ? spp < 128 : true);
    }

    bool renderSPP(const LogFn &log) {  // GP:1342-1426
        const size_t sampleCount = (size_t)m_cfg.budget;
        const int nPasses = (int)std::ceil(sampleCount / (float)m_cfg.sppPerPass);
        float currentVarAtEnd = std::numeric_limits<float>::infinity();
        const bool automatic = std::string(m_cfg.sampleCombination) == "automatic";
        int iter = 0;
        m_passesRendered = 0;
        while (m_passesRendered < nPasses) {
            const int sppRendered = m_passesRendered * m_cfg.sppPerPass;
            ppg_set_do_nee(m_ctx, doNeeWithSpp(sppRendered));
            int remainingPasses = nPasses - m_passesRendered;
            int passesThisIteration = std::min(remainingPasses, 1 << iter);
            if (remainingPasses - passesThisIteration < 2 * passesThisIteration) passesThisIteration = remainingPasses;
            say(log, fmt("ITERATION %d, %d passes", iter, passesThisIteration));
            say(log, "Resetting distributions for sampling.");
            bool isFinal = passesThisIteration >= remainingPasses;
            const bool recorded = !isFinal;  // training passes of this iteration go into the building tree — also when FINAL passes follow them (GP:1400-1411)
            check(ppg_begin_iteration(m_ctx, isFinal), "ppg_begin_iteration");
            m_filmComplete = false;
            ppg_pass_stats st;
            if (!passes(passesThisIteration, st, log, isFinal)) return false;
            const float lastVarAtEnd = currentVarAtEnd;
            currentVarAtEnd = passesThisIteration * st.variance / remainingPasses;
            say(log, fmt("Extrapolated var:\n  Last:    %f\n  Current: %f\n", lastVarAtEnd, currentVarAtEnd));
            remainingPasses -= passesThisIteration;
            if (automatic && remainingPasses > 0 && (remainingPasses < passesThisIteration || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
                say(log, fmt("FINAL %d passes", remainingPasses));
                ppg_set_final(m_ctx, 1);
                isFinal = true;
                if (!passes(remainingPasses, st, log, true)) return false;
            }
            build(log, !recorded);
            check(ppg_end_iteration(m_ctx), "ppg_end_iteration");
            ++iter;
        }
        return true;
    }

    bool renderTime(const LogFn &log) {  // GP:1434-1514
        const float nSeconds = m_cfg.budget;
        float currentVarAtEnd = std::numeric_limits<float>::infinity();
        const bool automatic = std::string(m_cfg.sampleCombination) == "automatic";
        const auto start = std::chrono::steady_clock::now();
        auto elapsed = [&](std::chrono::steady_clock::time_point t0) {
            const float local = (float)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000;
            return m_reducer ? (float)m_reducer->broadcast(local) : local;  // sharded: rank 0's clock
        };
        float elapsedSeconds = 0;
        int iter = 0;
        m_passesRendered = 0;
        while (elapsedSeconds < nSeconds) {
            const int sppRendered = m_passesRendered * m_cfg.sppPerPass;
            ppg_set_do_nee(m_ctx, doNeeWithSpp(sppRendered));
            float remainingTime = nSeconds - elapsedSeconds;
            const int passesThisIteration = 1 << iter;
            say(log, fmt("ITERATION %d, %d passes", iter, passesThisIteration));
            const auto startIter = std::chrono::steady_clock::now();
            check(ppg_begin_iteration(m_ctx, 0), "ppg_begin_iteration");
            ppg_pass_stats st;
            if (!passes(passesThisIteration, st, log)) return false;
            const float secondsIter = elapsed(startIter);
            const float lastVarAtEnd = currentVarAtEnd;
            currentVarAtEnd = secondsIter * st.variance / remainingTime;
            remainingTime -= secondsIter;
            if (automatic && remainingTime > 0 && (remainingTime < secondsIter || (sppRendered > 256 && currentVarAtEnd > lastVarAtEnd))) {
                say(log, fmt("FINAL %f seconds", remainingTime));
                ppg_set_final(m_ctx, 1);
                do {
                    if (!passes(passesThisIteration, st, log)) return false;
                    elapsedSeconds = elapsed(start);
                } while (elapsedSeconds < nSeconds);
            }
            build(log);
            check(ppg_end_iteration(m_ctx), "ppg_end_iteration");
            ++iter;
            elapsedSeconds = elapsed(start);
        }
        return true;
    }

    static int stopHook(void *self, int localStop) {
        GuidedPathTracerHIP *g = static_cast<GuidedPathTracerHIP *>(self);
        // (PPG_STOP_CANCELLED: the library's render was cancelled — also by a ppg_cancel() that did not come through cancel())
        if (g->m_cancelled.load() || localStop == PPG_STOP_CANCELLED) g->m_reducer->setLocalStatus(1);
        try { return g->m_reducer->stopDecision(localStop); } catch (...) { if (!g->m_hookError) g->m_hookError = std::current_exception(); return 1; }
    }
    static int roundHook(void *self) {  // C callback: no exception may cross the C-ABI
        GuidedPathTracerHIP *g = static_cast<GuidedPathTracerHIP *>(self);
        // (a cancelled rank stays in step with the others' round hooks — the library keeps calling this one with empty rounds — and says so
        // in the exchange: every rank then sees the status and they abort together, none left waiting in a collective)
        if (g->m_cancelled.load()) g->m_reducer->setLocalStatus(1);
        try { g->m_reducer->reduceAdamRecords(g->m_ctx); return 0; } catch (...) { if (!g->m_hookError) g->m_hookError = std::current_exception(); return 1; }
    }
    void rethrowHookError() { if (m_hookError) { std::exception_ptr e = m_hookError; m_hookError = nullptr; std::rethrow_exception(e); } }

    Reducer *m_reducer = nullptr;
    std::atomic<bool> m_cancelled{false};
    bool m_filmComplete = false;  // the exchange of a final iteration's groups left the complete film on every rank
    std::exception_ptr m_hookError;
    ppg_config m_cfg;
    ppg_ctx *m_ctx = nullptr;
    std::string m_str[7];
    int m_w = 0, m_h = 0, m_passesRendered = 0;
};

}  // namespace ppg
#endif
