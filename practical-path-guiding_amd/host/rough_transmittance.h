/* Rough-transmittance slices for the `roughplastic` BSDF (include/ppg.h: ppg_scene.rtrans) — the C++ twin of
 * ppg_host/rtrans.py.
 *
 * Mitsuba tabulates the transmittance through a rough dielectric boundary in data/microfacet/{beckmann,ggx}.dat and
 * RoughPlastic::configure() (roughplastic.cpp:285-305) reduces the table to one curve over the warped incident cosine plus the
 * diffuse transmittance from the inside.  Restated here: the file layout (rtrans.h:81-146), setEta / setAlpha / evalDiffuse
 * (rtrans.h:233-400) and the cubic interpolation they use (spline.cpp:23-60, 236-452), all in float like the reference's
 * SINGLE_PRECISION build.  The tables are Mitsuba's data: read from the operator's Mitsuba tree, never bundled. */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ppg {

// spline.cpp:23-60 on [0, 1], extrapolate = false
inline float evalCubicInterp1D(float x, const float *values, size_t size) {
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    float t = (x * (float)(size - 1)) / 1.0f;
    const size_t k = std::min((size_t)t, size - 2);
    const float f0 = values[k], f1 = values[k + 1];
    const float d0 = k > 0 ? 0.5f * (values[k + 1] - values[k - 1]) : values[k + 1] - values[k];
    const float d1 = k + 2 < size ? 0.5f * (values[k + 2] - values[k]) : values[k + 1] - values[k];
    t = t - (float)k;
    const float t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

// spline.cpp:244-288 / 387-431: knot and the four node weights along one dimension; false if p is outside [0, 1]
inline bool cubicKnotWeights(float p, size_t size, size_t &knot, float w[4]) {
    if (!(p >= 0.0f && p <= 1.0f)) return false;
    float t = (p * (float)(size - 1)) / 1.0f;
    knot = std::min((size_t)t, size - 2);
    t = t - (float)knot;
    const float t2 = t * t, t3 = t2 * t;
    w[0] = 0.0f; w[1] = 2 * t3 - 3 * t2 + 1; w[2] = -2 * t3 + 3 * t2; w[3] = 0.0f;
    const float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
    if (knot > 0) { w[2] += 0.5f * d0; w[0] -= 0.5f * d0; }
    else { w[2] += d0; w[1] -= d0; }
    if (knot + 2 < size) { w[3] += 0.5f * d1; w[1] -= 0.5f * d1; }
    else { w[2] += d1; w[1] -= d1; }
    return true;
}
inline float evalCubicInterp2D(float px, float py, const float *values, size_t sx, size_t sy) {  // spline.cpp:236-304
    size_t kx, ky;
    float wx[4], wy[4];
    if (!cubicKnotWeights(px, sx, kx, wx) || !cubicKnotWeights(py, sy, ky, wy)) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y)
        for (int x = -1; x <= 2; ++x) {
            const float wxy = wx[x + 1] * wy[y + 1];
            if (wxy == 0) continue;
            result += values[(ky + y) * sx + kx + x] * wxy;
        }
    return result;
}
inline float evalCubicInterp3D(float px, float py, float pz, const float *values, size_t sx, size_t sy, size_t sz) {  // spline.cpp:379-451
    size_t kx, ky, kz;
    float wx[4], wy[4], wz[4];
    if (!cubicKnotWeights(px, sx, kx, wx) || !cubicKnotWeights(py, sy, ky, wy) || !cubicKnotWeights(pz, sz, kz, wz)) return 0.0f;
    float result = 0.0f;
    for (int z = -1; z <= 2; ++z)
        for (int y = -1; y <= 2; ++y) {
            const float wyz = wy[y + 1] * wz[z + 1];
            for (int x = -1; x <= 2; ++x) {
                const float wxyz = wx[x + 1] * wyz;
                if (wxyz == 0) continue;
                result += values[((kz + z) * sy + (ky + y)) * sx + kx + x] * wxyz;
            }
        }
    return result;
}

class RoughTransmittance {
public:
    explicit RoughTransmittance(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(path + " not found (data/microfacet of a Mitsuba tree)");
        char header[17];
        uint64_t sizes[3];
        float range[4];
        if (!f.read(header, 17) || memcmp(header, "MTS_TRANSMITTANCE", 17) != 0 || !f.read((char *)sizes, 24) || !f.read((char *)range, 16))
            throw std::runtime_error(path + ": not a rough-transmittance data file");
        m_eta = sizes[0]; m_alpha = sizes[1]; m_theta = sizes[2];
        m_etaMin = range[0]; m_etaMax = range[1]; m_alphaMin = range[2]; m_alphaMax = range[3];
        std::vector<float> raw(2 * m_eta * m_alpha * (m_theta + 1));
        if (!f.read((char *)raw.data(), raw.size() * 4)) throw std::runtime_error(path + ": truncated rough-transmittance data file");
        m_trans.resize(2 * m_eta * m_alpha * m_theta); m_diff.resize(2 * m_eta * m_alpha);
        size_t a = 0, b = 0, c = 0;
        for (size_t i = 0; i < 2 * m_eta; ++i)
            for (size_t j = 0; j < m_alpha; ++j) {
                for (size_t k = 0; k < m_theta; ++k) m_trans[a++] = raw[c++];
                m_diff[b++] = raw[c++];
            }
    }
    size_t thetaSamples() const { return m_theta; }
    void check(float alpha, float eta) const {  // rtrans.h:402-420
        if (eta < 1) eta = 1 / eta;
        if (eta < m_etaMin || eta > m_etaMax) throw std::runtime_error("relative IOR " + std::to_string(eta) + " is outside the tabulated range");
        if (alpha < m_alphaMin || alpha > m_alphaMax) throw std::runtime_error("roughness alpha = " + std::to_string(alpha) + " is outside the tabulated range");
    }
    void setEta(float eta) {  // rtrans.h:299-349
        const float *trans = m_trans.data(), *diff = m_diff.data();
        if (eta < 1) { trans += m_eta * m_alpha * m_theta; diff += m_eta * m_alpha; eta = 1.0f / eta; }
        if (eta < m_etaMin) eta = m_etaMin;
        const float warpedEta = std::pow((eta - m_etaMin) / (m_etaMax - m_etaMin), 0.25f);
        std::vector<float> newTrans(m_alpha * m_theta), newDiff(m_alpha);
        const float dAlpha = 1.0f / (m_alpha - 1), dTheta = 1.0f / (m_theta - 1);
        for (size_t i = 0; i < m_alpha; ++i) {
            for (size_t j = 0; j < m_theta; ++j) newTrans[i * m_theta + j] = evalCubicInterp3D(j * dTheta, i * dAlpha, warpedEta, trans, m_theta, m_alpha, m_eta);
            newDiff[i] = evalCubicInterp2D(i * dAlpha, warpedEta, diff, m_alpha, m_eta);
        }
        m_trans.swap(newTrans); m_diff.swap(newDiff); m_etaFixed = true;
    }
    void setAlpha(float alpha) {  // rtrans.h:357-400
        const float warpedAlpha = std::pow((alpha - m_alphaMin) / (m_alphaMax - m_alphaMin), 0.25f);
        std::vector<float> newTrans(m_theta), newDiff(1);
        const float dTheta = 1.0f / (m_theta - 1);
        for (size_t i = 0; i < m_theta; ++i) newTrans[i] = evalCubicInterp2D(i * dTheta, warpedAlpha, m_trans.data(), m_theta, m_alpha);
        newDiff[0] = evalCubicInterp1D(warpedAlpha, m_diff.data(), m_alpha);
        m_trans.swap(newTrans); m_diff.swap(newDiff); m_alphaFixed = true;
    }
    float evalDiffuse(float alpha) const {  // rtrans.h:249-258 (eta fixed)
        float result;
        if (m_alphaFixed) result = m_diff[0];
        else result = evalCubicInterp1D(std::pow((alpha - m_alphaMin) / (m_alphaMax - m_alphaMin), 0.25f), m_diff.data(), m_alpha);
        return std::min(1.0f, std::max(0.0f, result));
    }
    const std::vector<float> &trans() const { return m_trans; }

private:
    size_t m_eta = 0, m_alpha = 0, m_theta = 0;
    float m_etaMin = 0, m_etaMax = 0, m_alphaMin = 0, m_alphaMax = 0;
    bool m_etaFixed = false, m_alphaFixed = false;
    std::vector<float> m_trans, m_diff;
};

/* What RoughPlastic::configure() precomputes for (distribution, alpha, eta), in ppg_scene.rtrans layout: thetaSamples values of the
 * external transmittance, then the internal diffuse transmittance.  dataDir: the `data` directory of a Mitsuba tree ("" →
 * $PPG_MITSUBA_DATA). */
inline std::vector<float> roughplasticSlice(const std::string &distribution, float alpha, float eta, std::string dataDir) {
    if (dataDir.empty()) { const char *e = std::getenv("PPG_MITSUBA_DATA"); if (e) dataDir = e; }
    if (dataDir.empty())
        throw std::runtime_error("roughplastic needs Mitsuba's data/microfacet tables: pass --data-dir or set PPG_MITSUBA_DATA to the `data` directory of a Mitsuba tree");
    static std::map<std::string, RoughTransmittance> tables;
    const std::string path = dataDir + "/microfacet/" + distribution + ".dat";
    auto it = tables.find(path);
    if (it == tables.end()) it = tables.emplace(path, RoughTransmittance(path)).first;
    alpha = std::max(alpha, 1e-4f);  // MicrofacetDistribution clamps alpha (microfacet.h:135)
    it->second.check(alpha, eta);
    RoughTransmittance ext = it->second, internal = it->second;
    ext.setEta(eta); internal.setEta(1 / eta);
    ext.setAlpha(alpha);
    std::vector<float> slice = ext.trans();
    slice.push_back(internal.evalDiffuse(alpha));
    return slice;
}

}  // namespace ppg
