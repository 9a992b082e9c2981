/*
 * plugin_core.h — everything the Mitsuba plug-in (mitsuba_plugin/guided_path_hip.cpp) does that touches no Mitsuba type: the mapping of
 * the integrator's properties to ppg_config, and the control flow of Integrator::render() / cancel() above the C-ABI —
 *
 *     configure(props)                       GuidedPathTracer(const Properties &), GP:1014-1085 (names and defaults verbatim)
 *     render(scene, destinationFile)         ppg_create → ppg_set_scene → ppg_render, GP:1516-1585; the SD-tree dumps go to
 *                                            "<destinationFile>-NN.sdt" (GP:1191-1195), which is only known HERE — so the context is created
 *                                            here and not in the constructor (round 3's shim created it in the constructor and then changed a
 *                                            prefix the context had already copied: dumpSDTree through the plug-in wrote nothing)
 *     cancel()                               Integrator::cancel (integrator.h:84, GP:1643-1648), from any thread, at any time
 *     readFilm(rgb)                          the weight-normalised picture for film->setBitmap
 *
 * The plug-in cannot be compiled where Mitsuba's headers are missing; this header can, and the stand-alone driver runs the same code
 * (`ppg_render --plugin-core`), which is how tests/test_cpp_host.py exercises it.  Errors are returned (code + message), never thrown: the
 * plug-in reports them with Log(EError, ..), which may or may not throw.
 */
#ifndef PPG_PLUGIN_CORE_H
#define PPG_PLUGIN_CORE_H

#include <atomic>
#include <mutex>
#include <string>

#include "../../include/ppg.h"

namespace ppg {

class PluginCore {
public:
    PluginCore() {
        ppg_config_default(&m_cfg);
        const char *d[6] = {m_cfg.nee, m_cfg.sampleCombination, m_cfg.spatialFilter, m_cfg.directionalFilter, m_cfg.bsdfSamplingFractionLoss, m_cfg.budgetType};
        for (int k = 0; k < 6; ++k) m_s[k] = d[k] ? d[k] : "";
        pointStrings();
    }
    ~PluginCore() { ppg_ctx *c = m_ctx.exchange(nullptr); if (c) ppg_destroy(c); }
    PluginCore(const PluginCore &) = delete;
    PluginCore &operator=(const PluginCore &) = delete;

    // P: anything with Mitsuba's typed getters (mitsuba::Properties; ppg::Properties of guided_path_hip.h)
    template <class P> void configure(const P &props) {
        m_s[0] = props.getString("nee", "never");
        m_s[1] = props.getString("sampleCombination", "automatic");
        m_s[2] = props.getString("spatialFilter", "nearest");
        m_s[3] = props.getString("directionalFilter", "nearest");
        m_s[4] = props.getString("bsdfSamplingFractionLoss", "none");
        m_s[5] = props.getString("budgetType", "seconds");
        m_cfg.sdTreeMaxMemory = props.getInteger("sdTreeMaxMemory", -1);
        m_cfg.sTreeThreshold = props.getInteger("sTreeThreshold", 12000);
        m_cfg.dTreeThreshold = props.getFloat("dTreeThreshold", 0.01f);
        m_cfg.bsdfSamplingFraction = props.getFloat("bsdfSamplingFraction", 0.5f);
        m_cfg.sppPerPass = props.getInteger("sppPerPass", 4);
        m_cfg.budget = props.getFloat("budget", 300.0f);
        m_cfg.dumpSDTree = props.getBoolean("dumpSDTree", false);
        m_cfg.rrDepth = props.getInteger("rrDepth", 5);  // MonteCarloIntegrator, integrator.cpp:192-218
        m_cfg.maxDepth = props.getInteger("maxDepth", -1);
        m_cfg.strictNormals = props.getBoolean("strictNormals", false);
        m_cfg.hideEmitters = props.getBoolean("hideEmitters", false);
        m_cfg.device = props.getInteger("device", 0);
        pointStrings();
    }
    void setSeed(uint64_t seed) { m_cfg.seed = seed; }
    const ppg_config &config() const { return m_cfg; }

    // PPG_OK, PPG_ERR_CANCELLED, or an error code with `err` set.  May be called again (a new context per render, like a new RenderJob).
    int render(const ppg_scene &scene, const std::string &destinationFile, std::string &err) {
        err.clear();
        {   // a context of an earlier render() is replaced (under the lock: cancel() must not reach a context that is being destroyed)
            std::lock_guard<std::mutex> lock(m_mutex);
            ppg_ctx *old = m_ctx.exchange(nullptr);
            if (old) ppg_destroy(old);
        }
        m_dump = m_cfg.dumpSDTree ? destinationFile : std::string();   // "<dest>-NN.sdt", GP:1191-1195
        pointStrings();
        m_cfg.dumpPrefix = m_dump.empty() ? nullptr : m_dump.c_str();
        ppg_ctx *ctx = nullptr;
        int rc = ppg_create(&m_cfg, &ctx);
        if (rc != PPG_OK) { err = ppg_last_error(nullptr); return rc; }  // e.g. an unknown enum string: where GP:1023.. Assert(false)
        {
            std::lock_guard<std::mutex> lock(m_mutex);  // cancel() either sees the context or has set the flag before this point
            m_ctx.store(ctx);
            if (m_cancelRequested.exchange(false)) return PPG_ERR_CANCELLED;
        }
        m_hasFilm = false;
        rc = ppg_set_scene(ctx, &scene);  // (a cancel() during these seconds of BVH build stays set in the context: ppg_render returns at once)
        if (rc != PPG_OK) { err = ppg_last_error(ctx); return rc; }
        m_hasFilm = true;  // (the film exists from here on: black if the cancel came before the first pass)
        rc = ppg_render(ctx);
        if (rc != PPG_OK && rc != PPG_ERR_CANCELLED) err = ppg_last_error(ctx);
        return rc;
    }
    // is there a picture to read?  Not after a cancel() that arrived before the context had a scene: the plug-in then returns false (GP:1584)
    // without touching the film
    bool hasFilm() const { return m_hasFilm; }

    void cancel() {
        std::lock_guard<std::mutex> lock(m_mutex);
        ppg_ctx *c = m_ctx.load();
        if (c) ppg_cancel(c); else m_cancelRequested.store(true);  // before render() has a context: render() returns at once
    }

    int readFilm(float *rgb, std::string &err) {
        ppg_ctx *c = m_ctx.load();
        if (!c) { err = "no render has been started"; return PPG_ERR_INVALID; }
        const int rc = ppg_read_film(c, rgb);
        if (rc != PPG_OK) err = ppg_last_error(c);
        return rc;
    }
    ppg_ctx *context() { return m_ctx.load(); }

private:
    void pointStrings() {
        m_cfg.nee = m_s[0].c_str(); m_cfg.sampleCombination = m_s[1].c_str(); m_cfg.spatialFilter = m_s[2].c_str();
        m_cfg.directionalFilter = m_s[3].c_str(); m_cfg.bsdfSamplingFractionLoss = m_s[4].c_str(); m_cfg.budgetType = m_s[5].c_str();
    }
    ppg_config m_cfg;
    std::string m_s[6], m_dump;
    std::atomic<ppg_ctx *> m_ctx{nullptr};
    std::atomic<bool> m_cancelRequested{false};
    bool m_hasFilm = false;
    std::mutex m_mutex;
};

}  // namespace ppg
#endif
