/*
 * scene_xml.h — C++ loader for the subset of Mitsuba 0.6 scene XML the reference's bundled scenes use, so that
 * `ppg_render scene.xml` reads what `mitsuba scene.xml` reads:
 *
 *   integrator  guided_path (properties of guided_path.cpp:1014-1085 and integrator.cpp:192-218)
 *   sensor      perspective (fov, fovAxis, nearClip, farClip, toWorld), film hdrfilm (width, height; rfilter box)
 *   shapes      sphere (center, radius, toWorld = rotation x uniform scale, flipNormals; analytic), obj (filename, toWorld, faceNormals, flipNormals, flipTexCoords, collapse), rectangle (toWorld, flipNormals)
 *   bsdfs       diffuse, conductor, roughconductor / roughdielectric / roughplastic (ggx / beckmann, isotropic; roughplastic reads Mitsuba's data/microfacet tables), plastic, dielectric, thindielectric, mask (constant opacity),
 *               twosided(BRDF) — top level with id, nested, or <ref id>
 *   emitters    area (nested in a shape), constant (environment), envmap (latitude-longitude .exr / .pfm / .hdr; filename, scale, toWorld = rotation),
 *               sunsky (baked into an envmap at load time, host/sunsky.h; its tables are parsed from the operator's Mitsuba source tree)
 *   values      <spectrum>, <rgb>, <srgb>; <transform> of translate / rotate / scale / lookAt / matrix; <default> and $name
 *
 * Anything else throws std::runtime_error naming the plugin.  Same semantics as ppg_host/mitsuba_xml.py (the two are tested against each
 * other); the OBJ reader follows shapes/obj.cpp:198-342 and TriMesh::computeNormals (trimesh.cpp:608-676).  Header only.
 */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "guided_path_hip.h"
#include "rough_transmittance.h"
#include "hdr_image.h"

namespace ppg {

// ------------------------------------------------------------------------------------------------ minimal XML
struct XmlNode {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<XmlNode> children;
    const std::string *attr(const std::string &k) const {
        for (auto &a : attrs) if (a.first == k) return &a.second;
        return nullptr;
    }
    std::string get(const std::string &k, const std::string &d = "") const { auto *a = attr(k); return a ? *a : d; }
    const XmlNode *child(const std::string &t) const {
        for (auto &c : children) if (c.tag == t) return &c;
        return nullptr;
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string &text) : s(text), p(0) {}
    XmlNode parseDocument() {
        skipMisc();
        XmlNode root = parseElement();
        return root;
    }

private:
    const std::string &s;
    size_t p;
    [[noreturn]] void fail(const std::string &m) const { throw std::runtime_error("XML: " + m + " at offset " + std::to_string(p)); }
    void skipWs() { while (p < s.size() && std::isspace((unsigned char)s[p])) ++p; }
    bool starts(const char *t) const { return s.compare(p, strlen(t), t) == 0; }
    void skipMisc() {
        for (;;) {
            skipWs();
            if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) fail("unterminated <?"); p = e + 2; }
            else if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("unterminated comment"); p = e + 3; }
            else if (starts("<!")) { size_t e = s.find('>', p); if (e == std::string::npos) fail("unterminated <!"); p = e + 1; }
            else break;
        }
    }
    static std::string unescape(const std::string &v) {
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] != '&') { o += v[i]; continue; }
            static const std::pair<const char *, char> ent[] = {{"&quot;", '"'}, {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&apos;", '\''}};
            bool done = false;
            for (auto &e : ent) if (v.compare(i, strlen(e.first), e.first) == 0) { o += e.second; i += strlen(e.first) - 1; done = true; break; }
            if (!done) o += v[i];
        }
        return o;
    }
    std::string name() {
        size_t b = p;
        while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.')) ++p;
        if (p == b) fail("name expected");
        return s.substr(b, p - b);
    }
    XmlNode parseElement() {
        if (p >= s.size() || s[p] != '<') fail("'<' expected");
        ++p;
        XmlNode n;
        n.tag = name();
        for (;;) {
            skipWs();
            if (p >= s.size()) fail("unterminated tag");
            if (s[p] == '/') { if (s.compare(p, 2, "/>") != 0) fail("'/>' expected"); p += 2; return n; }
            if (s[p] == '>') { ++p; break; }
            std::string k = name();
            skipWs();
            if (p >= s.size() || s[p] != '=') fail("'=' expected");
            ++p; skipWs();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) fail("quoted value expected");
            char q = s[p++];
            size_t e = s.find(q, p);
            if (e == std::string::npos) fail("unterminated attribute");
            n.attrs.emplace_back(k, unescape(s.substr(p, e - p)));
            p = e + 1;
        }
        for (;;) {  // content
            size_t lt = s.find('<', p);
            if (lt == std::string::npos) fail("unterminated element <" + n.tag + ">");
            p = lt;
            if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("unterminated comment"); p = e + 3; continue; }
            if (starts("</")) {
                p += 2;
                std::string t = name();
                if (t != n.tag) fail("</" + t + "> closes <" + n.tag + ">");
                skipWs();
                if (p >= s.size() || s[p] != '>') fail("'>' expected");
                ++p;
                return n;
            }
            n.children.push_back(parseElement());
        }
    }
};

// ------------------------------------------------------------------------------------------------ small float algebra
struct Mat4 {
    float m[16];
    static Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
    Mat4 operator*(const Mat4 &o) const {
        Mat4 r{};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float a = 0; for (int k = 0; k < 4; ++k) a += m[4 * i + k] * o.m[4 * k + j]; r.m[4 * i + j] = a; }
        return r;
    }
};
struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalized(V3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
inline V3 xfPoint(const Mat4 &m, V3 p) {
    float x = m.m[0] * p.x + m.m[1] * p.y + m.m[2] * p.z + m.m[3], y = m.m[4] * p.x + m.m[5] * p.y + m.m[6] * p.z + m.m[7];
    float z = m.m[8] * p.x + m.m[9] * p.y + m.m[10] * p.z + m.m[11], w = m.m[12] * p.x + m.m[13] * p.y + m.m[14] * p.z + m.m[15];
    if (w != 1) { x /= w; y /= w; z /= w; }
    return {x, y, z};
}
// inverse transpose of the upper 3x3 in double (Transform::operator()(Normal))
inline void normalMatrix(const Mat4 &m, double out[9]) {
    const double a = m.m[0], b = m.m[1], c = m.m[2], d = m.m[4], e = m.m[5], f = m.m[6], g = m.m[8], h = m.m[9], i = m.m[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det, (f * g - d * i) / det, (a * i - c * g) / det,
                           (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) out[3 * r + cc] = inv[3 * cc + r];  // transpose
}
inline V3 xfNormal(const double nm[9], V3 n) {
    return {(float)(nm[0] * n.x + nm[1] * n.y + nm[2] * n.z), (float)(nm[3] * n.x + nm[4] * n.y + nm[5] * n.z), (float)(nm[6] * n.x + nm[7] * n.y + nm[8] * n.z)};
}

// ------------------------------------------------------------------------------------------------ colour values (ppg_host/spectrum.py)
namespace detail {
static const double kCIE[471][3] = {
#include "cie1931_xyz_1nm.inc"
};
inline double evalInterp(const std::vector<double> &lam, const std::vector<double> &val, double g) {  // incl. the reversed lerp, spectrum.cpp:693-706
    if (lam.size() < 2 || g < lam.front() || g > lam.back()) return 0.0;
    size_t i = std::lower_bound(lam.begin(), lam.end(), g) - lam.begin();
    if (lam[std::min(i, lam.size() - 1)] == g) return val[std::min(i, lam.size() - 1)];
    i = std::max<size_t>(1, std::min(i, lam.size() - 1));
    const double a = lam[i - 1], b = lam[i], fa = val[i - 1], fb = val[i], t = (g - a) / (b - a);
    return (1 - t) * fb + t * fa;
}
}  // namespace detail

// scene-file spectra are zero-extended and clamped (scenehandler.cpp:563-566); the conductor plug-ins read data/ior/*.spd without either
inline void interpolatedToRGB(std::vector<double> lam, std::vector<double> val, float rgb[3], bool zeroExtend = true, bool clamp = true) {
    using namespace detail;
    const double spacing = (lam.back() - lam.front()) / (lam.size() - 1);  // zeroExtend, spectrum.cpp:630-648
    if (zeroExtend && val.front() != 0) { lam.insert(lam.begin(), lam.front() - spacing); val.insert(val.begin(), 0.0); }
    if (zeroExtend && val.back() != 0) { lam.push_back(lam.back() + spacing); val.push_back(0.0); }
    std::vector<double> wl(471), cx(471), cy(471), cz(471);
    for (int i = 0; i < 471; ++i) { wl[i] = 360.0 + i; cx[i] = kCIE[i][0]; cy[i] = kCIE[i][1]; cz[i] = kCIE[i][2]; }
    const double step = 0.00731;  // off-node sampling: exact hits have measure zero
    double X = 0, Y = 0, Z = 0, N = 0, pg = 0, ps[4] = {0, 0, 0, 0};
    bool first = true;
    for (long k = 0;; ++k) {
        const double g = 360.0 + k * step;
        if (!(g < 830.0)) break;
        const double s = evalInterp(lam, val, g), x = evalInterp(wl, cx, g), y = evalInterp(wl, cy, g), z = evalInterp(wl, cz, g);
        const double cur[4] = {s * x, s * y, s * z, y};
        if (!first) { const double h = 0.5 * (g - pg); X += h * (cur[0] + ps[0]); Y += h * (cur[1] + ps[1]); Z += h * (cur[2] + ps[2]); N += h * (cur[3] + ps[3]); }
        first = false; pg = g;
        for (int c = 0; c < 4; ++c) ps[c] = cur[c];
    }
    X /= N; Y /= N; Z /= N;
    const double r = 3.240479 * X - 1.537150 * Y - 0.498535 * Z, gg = -0.969256 * X + 1.875991 * Y + 0.041556 * Z, b = 0.055648 * X - 0.204043 * Y + 1.057311 * Z;
    rgb[0] = (float)(clamp ? std::max(r, 0.0) : r); rgb[1] = (float)(clamp ? std::max(gg, 0.0) : gg); rgb[2] = (float)(clamp ? std::max(b, 0.0) : b);
}

// InterpolatedSpectrum(path), spectrum.cpp:575-600: "wavelength value" lines, # comments; stops at the first malformed line
inline bool readSPD(const std::string &path, std::vector<double> &lam, std::vector<double> &val) {
    std::ifstream sf(path);
    if (!sf) return false;
    std::string line;
    while (std::getline(sf, line)) {
        const size_t a = line.find_first_not_of(" \t\r\n");
        if (a == std::string::npos || line[a] == '#') continue;
        std::istringstream ls(line);
        double l, v;
        if (!(ls >> l >> v)) break;
        lam.push_back(l); val.push_back(v);
    }
    return true;
}

// <blackbody temperature=".." scale=".."/> (scenehandler.cpp:534-547): Planck's law (BlackBodySpectrum::eval, spectrum.cpp:483-495) through
// Spectrum::fromContinuousSpectrum, clamped, times scale
inline void blackbodyToRGB(double temperature, double scale, float rgb[3]) {
    using namespace detail;
    std::vector<double> wl(471), cx(471), cy(471), cz(471);
    for (int i = 0; i < 471; ++i) { wl[i] = 360.0 + i; cx[i] = kCIE[i][0]; cy[i] = kCIE[i][1]; cz[i] = kCIE[i][2]; }
    const double step = 0.00731, cc = 299792458.0, k = 1.3806488e-23, h = 6.62606957e-34;
    double X = 0, Y = 0, Z = 0, N = 0, pg = 0, ps[4] = {0, 0, 0, 0};
    bool first = true;
    for (long n = 0;; ++n) {
        const double g = 360.0 + n * step;
        if (!(g < 830.0)) break;
        const double lambda = g * 1e-9;
        const double s = (2 * h * cc * cc) * std::pow(lambda, -5.0) / ((std::exp((h / k) * cc / (lambda * temperature)) - 1.0) * 1e9);
        const double x = evalInterp(wl, cx, g), y = evalInterp(wl, cy, g), z = evalInterp(wl, cz, g);
        const double cur[4] = {s * x, s * y, s * z, y};
        if (!first) { const double hh = 0.5 * (g - pg); X += hh * (cur[0] + ps[0]); Y += hh * (cur[1] + ps[1]); Z += hh * (cur[2] + ps[2]); N += hh * (cur[3] + ps[3]); }
        first = false; pg = g;
        for (int c = 0; c < 4; ++c) ps[c] = cur[c];
    }
    X /= N; Y /= N; Z /= N;
    const double r = 3.240479 * X - 1.537150 * Y - 0.498535 * Z, gg = -0.969256 * X + 1.875991 * Y + 0.041556 * Z, b = 0.055648 * X - 0.204043 * Y + 1.057311 * Z;
    rgb[0] = (float)std::max(r, 0.0) * (float)scale; rgb[1] = (float)std::max(gg, 0.0) * (float)scale; rgb[2] = (float)std::max(b, 0.0) * (float)scale;
}

}  // namespace ppg
#include "sunsky.h"
namespace ppg {

inline std::vector<double> parseFloats(std::string t) {
    for (char &c : t) if (c == ',') c = ' ';
    std::istringstream is(t);
    std::vector<double> v;
    double x;
    while (is >> x) v.push_back(x);
    return v;
}

inline void parseColour(const std::string &tag, const std::string &value, float rgb[3]) {
    if (tag == "rgb" || tag == "srgb") {
        double c[3];
        std::string v = value;
        v.erase(0, v.find_first_not_of(" \t"));
        if (!v.empty() && v[0] == '#') {
            for (int k = 0; k < 3; ++k) c[k] = std::stoi(v.substr(1 + 2 * k, 2), nullptr, 16) / 255.0;
        } else {
            auto f = parseFloats(v);
            if (f.size() == 1) f = {f[0], f[0], f[0]};
            if (f.size() != 3) throw std::runtime_error("expected 1 or 3 colour components, got '" + value + "'");
            for (int k = 0; k < 3; ++k) c[k] = f[k];
        }
        for (int k = 0; k < 3; ++k) rgb[k] = (float)(tag == "srgb" ? (c[k] <= 0.04045 ? c[k] / 12.92 : std::pow((c[k] + 0.055) / 1.055, 2.4)) : c[k]);
        return;
    }
    if (tag == "spectrum") {
        if (value.find(':') != std::string::npos) {
            std::string t = value;
            for (char &ch : t) if (ch == ',' || ch == ':') ch = ' ';
            auto f = parseFloats(t);
            std::vector<double> lam, val;
            for (size_t k = 0; k + 1 < f.size(); k += 2) { lam.push_back(f[k]); val.push_back(f[k + 1]); }
            interpolatedToRGB(lam, val, rgb);
            return;
        }
        auto f = parseFloats(value);
        if (f.size() == 1) { rgb[0] = rgb[1] = rgb[2] = (float)f[0]; return; }
        if (f.size() == 3) { for (int k = 0; k < 3; ++k) rgb[k] = (float)f[k]; return; }
    }
    throw std::runtime_error("unsupported colour element <" + tag + " value=\"" + value + "\">");
}

// ------------------------------------------------------------------------------------------------ OBJ (shapes/obj.cpp:198-342)
struct Mesh {
    std::vector<V3> positions, normals;  // normals empty = none
    std::vector<uint32_t> indices;
};

inline float unitAngle(V3 u, V3 v) {  // util.h:309-314
    V3 s{v.x + u.x, v.y + u.y, v.z + u.z}, d{v.x - u.x, v.y - u.y, v.z - u.z};
    if (dot(u, v) < 0) return 3.14159265358979323846f - 2 * std::asin(0.5f * std::sqrt(dot(s, s)));
    return 2 * std::asin(0.5f * std::sqrt(dot(d, d)));
}

inline void computeNormals(Mesh &m, bool flip) {  // TriMesh::computeNormals, smooth branch (trimesh.cpp:631-671)
    m.normals.assign(m.positions.size(), V3{0, 0, 0});
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3) {
        V3 n{0, 0, 0};
        for (int i = 0; i < 3; ++i) {
            const V3 &v0 = m.positions[m.indices[t + i]], &v1 = m.positions[m.indices[t + (i + 1) % 3]], &v2 = m.positions[m.indices[t + (i + 2) % 3]];
            V3 sideA = v1 - v0, sideB = v2 - v0;
            if (i == 0) {
                n = cross(sideA, sideB);
                float len = std::sqrt(dot(n, n));
                if (len == 0) break;
                n = {n.x / len, n.y / len, n.z / len};
            }
            float angle = unitAngle(normalized(sideA), normalized(sideB));
            V3 &dst = m.normals[m.indices[t + i]];
            dst = {dst.x + n.x * angle, dst.y + n.y * angle, dst.z + n.z * angle};
        }
    }
    for (V3 &n : m.normals) {
        float len = std::sqrt(dot(n, n));
        if (flip) len *= -1;
        if (len != 0) n = {n.x / len, n.y / len, n.z / len};
        else n = {1, 0, 0};
    }
}

/* TriMesh::rebuildTopology (trimesh.cpp:468-608): vertices re-merged on (position[, uv]); around each, the incident triangles are
 * clustered greedily by face normal (one new vertex per cluster of faces within maxAngle degrees of the cluster's first face), so that
 * the smooth normals computed afterwards stop at creases.  New vertices are numbered in the reference's order. */
inline void rebuildTopology(Mesh &m, const std::vector<std::pair<float, float>> *uvs, float maxAngle) {
    const float dpThresh = std::cos(maxAngle * (float)(M_PI / 180.0));
    const size_t nt = m.indices.size() / 3;
    std::vector<V3> fn(nt);
    for (size_t t = 0; t < nt; ++t) {
        const V3 v0 = m.positions[m.indices[3 * t]], v1 = m.positions[m.indices[3 * t + 1]], v2 = m.positions[m.indices[3 * t + 2]];
        V3 n = cross(v1 - v0, v2 - v0);
        const float l = std::sqrt(dot(n, n));
        fn[t] = l > 2.93873587705571876e-39f ? V3{n.x / l, n.y / l, n.z / l} : V3{0, 0, 0};  // RCPOVERFLOW_FLT
    }
    struct Key { float v[5]; int n; bool operator<(const Key &o) const { return std::lexicographical_compare(v, v + n, o.v, o.v + o.n); } };
    std::multimap<Key, std::pair<uint32_t, bool>> vertexToFace;  // equal keys keep their insertion order (triangle, corner)
    for (size_t t = 0; t < nt; ++t)
        for (int j = 0; j < 3; ++j) {
            const uint32_t v = m.indices[3 * t + j];
            Key k{{m.positions[v].x, m.positions[v].y, m.positions[v].z, uvs ? (*uvs)[v].first : 0.0f, uvs ? (*uvs)[v].second : 0.0f}, uvs ? 5 : 3};
            vertexToFace.insert({k, {(uint32_t)t, false}});
        }
    std::vector<V3> newPositions;
    std::vector<uint32_t> newIndices(m.indices.size(), 0xFFFFFFFFu);
    for (auto it = vertexToFace.begin(); it != vertexToFace.end();) {
        auto end = vertexToFace.upper_bound(it->first);
        const V3 p{it->first.v[0], it->first.v[1], it->first.v[2]};
        for (auto it2 = it; it2 != end; ++it2) {
            if (it2->second.second) continue;
            const V3 n1 = fn[it2->second.first];
            const uint32_t vertexIdx = (uint32_t)newPositions.size();
            newPositions.push_back(p);
            for (auto it3 = it2; it3 != end; ++it3) {
                if (it3->second.second) continue;
                const V3 n2 = fn[it3->second.first];
                if ((n1.x == n2.x && n1.y == n2.y && n1.z == n2.z) || dot(n1, n2) > dpThresh) {
                    for (int i = 0; i < 3; ++i) {
                        const V3 q = m.positions[m.indices[3 * it3->second.first + i]];
                        if (q.x == p.x && q.y == p.y && q.z == p.z) newIndices[3 * it3->second.first + i] = vertexIdx;
                    }
                    it3->second.second = true;
                }
            }
        }
        it = end;
    }
    m.positions.swap(newPositions);
    m.indices.swap(newIndices);
    m.normals.clear();
}

inline std::vector<Mesh> loadOBJ(const std::string &path, const Mat4 &toWorld, bool faceNormals, bool flipNormals, bool flipTexCoords, bool collapse,
                                 float maxSmoothAngle = -1.0f) {
    std::ifstream is(path);
    if (!is) throw std::runtime_error("Wavefront OBJ file '" + path + "' not found!");
    std::vector<V3> V, N;
    std::vector<std::pair<float, float>> UV;
    struct Corner { int p, uv, n; };
    std::vector<Corner> tris;  // 3 per triangle
    std::vector<Mesh> meshes;
    double nm[9];
    normalMatrix(toWorld, nm);
    auto flush = [&]() {
        if (tris.empty()) return;
        std::vector<V3> pw(V.size()), nw(N.size());
        for (size_t i = 0; i < V.size(); ++i) pw[i] = xfPoint(toWorld, V[i]);
        for (size_t i = 0; i < N.size(); ++i) {
            V3 n = xfNormal(nm, N[i]);
            float l = std::sqrt(dot(n, n));
            nw[i] = l != 0 ? V3{n.x / l, n.y / l, n.z / l} : n;
        }
        struct Key { float v[8]; bool operator<(const Key &o) const { return std::lexicographical_compare(v, v + 8, o.v, o.v + 8); } };
        std::map<Key, uint32_t> vmap;
        Mesh m;
        std::vector<V3> vn;
        std::vector<std::pair<float, float>> vuv;
        bool hasNormals = false, hasUVs = false;
        for (const Corner &c : tris) {
            int p = c.p, n = c.n, uv = c.uv;
            if (p < 0) p += (int)V.size() + 1;
            if (n < 0) n += (int)N.size() + 1;
            if (uv < 0) uv += (int)UV.size() + 1;
            if (p <= 0 || p > (int)V.size()) throw std::runtime_error(path + ": vertex index out of bounds");
            if (n > (int)N.size() || uv > (int)UV.size()) throw std::runtime_error(path + ": normal / uv index out of bounds");
            V3 pn = n ? nw[n - 1] : V3{0, 0, 0};
            hasNormals |= n != 0;
            hasUVs |= uv != 0;
            std::pair<float, float> puv = uv ? UV[uv - 1] : std::make_pair(0.0f, 0.0f);
            Key k{{pw[p - 1].x, pw[p - 1].y, pw[p - 1].z, pn.x, pn.y, pn.z, puv.first, puv.second}};
            auto it = vmap.find(k);
            uint32_t id;
            if (it == vmap.end()) { id = (uint32_t)m.positions.size(); vmap[k] = id; m.positions.push_back(pw[p - 1]); vn.push_back(pn); vuv.push_back(puv); }
            else id = it->second;
            m.indices.push_back(id);
        }
        if (maxSmoothAngle >= 0) { rebuildTopology(m, hasUVs ? &vuv : nullptr, maxSmoothAngle); hasNormals = false; }  // obj.cpp:336-343
        if (faceNormals) {
            if (flipNormals) for (size_t t = 0; t + 2 < m.indices.size(); t += 3) std::swap(m.indices[t], m.indices[t + 1]);
        } else if (hasNormals) {
            m.normals = vn;
            if (flipNormals) for (V3 &n : m.normals) n = {-n.x, -n.y, -n.z};
        } else {
            computeNormals(m, flipNormals);
        }
        meshes.push_back(std::move(m));
        tris.clear();
    };
    std::string line, pending;
    auto handle = [&](const std::string &ln) {
        std::istringstream iss(ln);
        std::string buf;
        if (!(iss >> buf)) return;
        if (buf == "v") { V3 p{0, 0, 0}; iss >> p.x >> p.y >> p.z; V.push_back(p); }
        else if (buf == "vn") { V3 n{0, 0, 0}; iss >> n.x >> n.y >> n.z; N.push_back(n); }
        else if (buf == "vt") { float u = 0, v = 0; iss >> u >> v; UV.emplace_back(u, flipTexCoords ? 1 - v : v); }
        else if (buf == "g" && !collapse) flush();
        else if (buf == "usemtl") { if (!collapse) flush(); }
        else if (buf == "f") {
            std::vector<Corner> cs;
            std::string tok;
            while (iss >> tok) {
                Corner c{0, 0, 0};
                size_t a = tok.find('/');
                c.p = std::atoi(tok.substr(0, a).c_str());
                if (a != std::string::npos) {
                    size_t b = tok.find('/', a + 1);
                    std::string suv = tok.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
                    if (!suv.empty()) c.uv = std::atoi(suv.c_str());
                    if (b != std::string::npos && b + 1 < tok.size()) c.n = std::atoi(tok.substr(b + 1).c_str());
                }
                cs.push_back(c);
            }
            if (cs.size() < 3) throw std::runtime_error(path + ": face with fewer than 3 vertices");
            for (size_t k = 1; k + 1 < cs.size(); ++k) { tris.push_back(cs[0]); tris.push_back(cs[k]); tris.push_back(cs[k + 1]); }  // fan
        }
    };
    while (std::getline(is, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == '\t' || line.back() == ' ')) line.pop_back();
        if (!line.empty() && line.back() == '\\') { line.pop_back(); pending += line; continue; }  // obj.cpp:166-188
        handle(pending + line);
        pending.clear();
    }
    if (!pending.empty()) handle(pending);
    flush();
    return meshes;
}

/* shapes/serialized.cpp:148-212 + TriMesh::loadCompressed (trimesh.cpp:175-295): Mitsuba's binary mesh format — header 0x041C, version 3 or
 * 4, then a zlib stream {flags, [name], vertex count, triangle count, positions, [normals], [texcoords], [colours], indices}; several
 * meshes per file are addressed through the offset table at the end (shapeIndex). */
inline Mesh loadSerialized(const std::string &path, const Mat4 &toWorld, int shapeIndex, bool faceNormals, bool flipNormals, float maxSmoothAngle) {
    const std::vector<unsigned char> buf = slurp(path, "serialized");
    auto header = [&](size_t off) {
        uint16_t h[2];
        if (off + 4 > buf.size()) throw std::runtime_error(path + ": encountered an invalid file format!");
        memcpy(h, &buf[off], 4);
        if (h[0] != 0x041C) throw std::runtime_error(path + ": encountered an invalid file format!");
        if (h[1] != 3 && h[1] != 4) throw std::runtime_error(path + ": encountered an incompatible file version!");
        return (int)h[1];
    };
    const int version = header(0);
    size_t start = 0;
    if (shapeIndex != 0) {
        // the offset table at the end of the file (readOffset, trimesh.cpp:272-295); every size comes from the file: check it against the buffer
        // (the reference accepts shapeIndex == count and then reads past the table — an error here)
        if (buf.size() < 8) throw std::runtime_error(path + ": truncated file");
        uint32_t count;
        memcpy(&count, &buf[buf.size() - 4], 4);
        if (shapeIndex < 0 || (uint64_t)shapeIndex >= (uint64_t)count) throw std::runtime_error(path + ": shape index is out of range!");
        const uint64_t entry = version == 4 ? 8 : 4, back = entry * (uint64_t)(count - shapeIndex) + 4;
        if (back > buf.size()) throw std::runtime_error(path + ": corrupt offset table");
        if (version == 4) { uint64_t o; memcpy(&o, &buf[buf.size() - back], 8); if (o > buf.size()) throw std::runtime_error(path + ": corrupt offset table"); start = (size_t)o; }
        else { uint32_t o; memcpy(&o, &buf[buf.size() - back], 4); start = o; }
        if (start + 4 > buf.size()) throw std::runtime_error(path + ": corrupt offset table");
        header(start);
    }
    std::vector<unsigned char> data;
    {
        z_stream zs{};
        if (inflateInit(&zs) != Z_OK) throw std::runtime_error(path + ": zlib failure");
        zs.next_in = const_cast<unsigned char *>(&buf[start + 4]);
        zs.avail_in = (uInt)(buf.size() - start - 4);
        unsigned char chunk[1 << 16];
        int rc;
        do {
            zs.next_out = chunk; zs.avail_out = sizeof chunk;
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error(path + ": corrupt zlib stream"); }
            data.insert(data.end(), chunk, chunk + (sizeof chunk - zs.avail_out));
        } while (rc != Z_STREAM_END);
        inflateEnd(&zs);
    }
    size_t off = 0;
    auto need = [&](size_t n) { if (off + n > data.size()) throw std::runtime_error(path + ": truncated mesh data"); };
    uint32_t flags;
    need(4); memcpy(&flags, &data[off], 4); off += 4;
    if (version == 4) { while (off < data.size() && data[off]) ++off; ++off; }
    uint64_t nv, nt;
    need(16); memcpy(&nv, &data[off], 8); memcpy(&nt, &data[off + 8], 8); off += 16;
    const bool dbl = (flags & 0x2000) != 0;
    if (nv > data.size() || nt > data.size()) throw std::runtime_error(path + ": truncated mesh data");  // every vertex / triangle takes > 1 byte: no overflow below
    auto take = [&](int comps, std::vector<float> &out) {
        const size_t n = (size_t)nv * comps;
        need(n * (dbl ? 8 : 4));
        out.resize(n);
        for (size_t k = 0; k < n; ++k) {
            if (dbl) { double v; memcpy(&v, &data[off + 8 * k], 8); out[k] = (float)v; }
            else memcpy(&out[k], &data[off + 4 * k], 4);
        }
        off += n * (dbl ? 8 : 4);
    };
    std::vector<float> P, N, UVs, C;
    take(3, P);
    if (flags & 0x0001) take(3, N);
    if (flags & 0x0002) take(2, UVs);
    if (flags & 0x0008) take(3, C);
    need((size_t)nt * 12);
    Mesh m;
    m.indices.resize((size_t)nt * 3);
    memcpy(m.indices.data(), &data[off], (size_t)nt * 12);
    for (uint32_t id : m.indices) if (id >= nv) throw std::runtime_error(path + ": vertex index out of bounds");
    bool identity = true;
    for (int k = 0; k < 16; ++k) identity &= toWorld.m[k] == Mat4::identity().m[k];
    double nm[9];
    normalMatrix(toWorld, nm);
    for (size_t i = 0; i < nv; ++i) {
        const V3 p{P[3 * i], P[3 * i + 1], P[3 * i + 2]};
        m.positions.push_back(identity ? p : xfPoint(toWorld, p));
        if (!N.empty()) {
            const V3 n{N[3 * i], N[3 * i + 1], N[3 * i + 2]};
            m.normals.push_back(identity ? n : normalized(xfNormal(nm, n)));
        }
    }
    const double a = toWorld.m[0], b = toWorld.m[1], c = toWorld.m[2], d = toWorld.m[4], e = toWorld.m[5], f = toWorld.m[6], g = toWorld.m[8], h = toWorld.m[9], i9 = toWorld.m[10];
    if (a * (e * i9 - f * h) - b * (d * i9 - f * g) + c * (d * h - e * g) < 0)
        for (size_t t = 0; t + 2 < m.indices.size(); t += 3) std::swap(m.indices[t], m.indices[t + 1]);
    std::vector<std::pair<float, float>> vuv;
    for (size_t i = 0; i + 1 < UVs.size(); i += 2) vuv.emplace_back(UVs[i], UVs[i + 1]);
    if (maxSmoothAngle >= 0) rebuildTopology(m, UVs.empty() ? nullptr : &vuv, maxSmoothAngle);
    if (faceNormals) {
        m.normals.clear();
        if (flipNormals) for (size_t t = 0; t + 2 < m.indices.size(); t += 3) std::swap(m.indices[t], m.indices[t + 1]);
    } else if (!m.normals.empty()) {
        if (flipNormals) for (V3 &n : m.normals) n = {-n.x, -n.y, -n.z};
    } else {
        computeNormals(m, flipNormals);
    }
    return m;
}

/* shapes/ply.cpp: Stanford PLY (ascii / binary little / big endian), vertex x y z [nx ny nz], faces of 3 or 4 indices (a quad becomes
 * (0, 1, 2), (3, 0, 2), ply.cpp:299-311); positions and normals go through toWorld as they are read (:229-233). */
inline Mesh loadPLY(const std::string &path, const Mat4 &toWorld, bool faceNormals, bool flipNormals, float maxSmoothAngle) {
    const std::vector<unsigned char> buf = slurp(path, "ply");
    const std::string all(buf.begin(), buf.end());
    const size_t end = all.find("end_header");
    if (all.compare(0, 3, "ply") != 0 || end == std::string::npos) throw std::runtime_error(path + ": not a PLY file");
    size_t off = all.find('\n', end) + 1;
    struct Prop { std::string name; bool list; std::string t0, t1; };
    struct Elem { std::string name; size_t count; std::vector<Prop> props; };
    std::vector<Elem> elems;
    std::string fmt;
    {
        std::istringstream hs(all.substr(0, end));
        std::string line;
        std::getline(hs, line);
        while (std::getline(hs, line)) {
            std::istringstream ls(line);
            std::string tok;
            if (!(ls >> tok) || tok == "comment" || tok == "obj_info") continue;
            if (tok == "format") ls >> fmt;
            else if (tok == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
            else if (tok == "property" && !elems.empty()) {
                Prop p; std::string a;
                ls >> a;
                if (a == "list") { p.list = true; ls >> p.t0 >> p.t1 >> p.name; }
                else { p.list = false; p.t0 = a; ls >> p.name; }
                elems.back().props.push_back(p);
            }
        }
    }
    if (fmt != "ascii" && fmt != "binary_little_endian" && fmt != "binary_big_endian") throw std::runtime_error(path + ": unknown PLY format '" + fmt + "'");
    auto sizeOf = [&](const std::string &t) -> int {
        if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
        if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
        if (t == "int" || t == "int32" || t == "uint" || t == "uint32" || t == "float" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        throw std::runtime_error(path + ": unknown PLY type '" + t + "'");
    };
    std::istringstream ascii(fmt == "ascii" ? all.substr(off) : std::string());
    const bool big = fmt == "binary_big_endian";
    auto next = [&](const std::string &t) -> double {
        if (fmt == "ascii") { double v; if (!(ascii >> v)) throw std::runtime_error(path + ": truncated PLY data"); return v; }
        const int n = sizeOf(t);
        if (off + n > buf.size()) throw std::runtime_error(path + ": truncated PLY data");
        unsigned char b[8];
        for (int k = 0; k < n; ++k) b[k] = buf[off + (big ? n - 1 - k : k)];
        off += n;
        if (t == "float" || t == "float32") { float v; memcpy(&v, b, 4); return v; }
        if (t == "double" || t == "float64") { double v; memcpy(&v, b, 8); return v; }
        if (t == "char" || t == "int8") return (signed char)b[0];
        if (t == "uchar" || t == "uint8") return b[0];
        if (t == "short" || t == "int16") { int16_t v; memcpy(&v, b, 2); return v; }
        if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, b, 2); return v; }
        if (t == "int" || t == "int32") { int32_t v; memcpy(&v, b, 4); return v; }
        uint32_t v; memcpy(&v, b, 4); return v;
    };
    std::map<std::string, std::vector<float>> cols;
    Mesh m;
    size_t nVerts = 0;
    for (const Elem &e : elems)
        for (size_t r = 0; r < e.count; ++r)
            for (const Prop &p : e.props) {
                if (p.list) {
                    const int n = (int)next(p.t0);
                    uint32_t v[4] = {0, 0, 0, 0};
                    const bool faceList = e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index");
                    if (faceList && n != 3 && n != 4) throw std::runtime_error(path + ": only triangle and quad-based PLY meshes are supported");
                    for (int k = 0; k < n; ++k) { const double d = next(p.t1); if (k < 4) v[k] = (uint32_t)d; }
                    if (faceList) {
                        for (uint32_t id : {v[0], v[1], v[2]}) m.indices.push_back(id);
                        if (n == 4) for (uint32_t id : {v[3], v[0], v[2]}) m.indices.push_back(id);
                    }
                } else {
                    const double d = next(p.t0);
                    if (e.name == "vertex") { cols[p.name].push_back((float)d); if (p.name == "x") ++nVerts; }
                }
            }
    if (m.indices.empty() || nVerts == 0) throw std::runtime_error("Unable to load \"" + path + "\" (no triangles or vertices found)!");
    for (uint32_t id : m.indices) if (id >= nVerts) throw std::runtime_error(path + ": vertex index out of bounds");
    const bool hasN = cols.count("nx") && cols.count("ny") && cols.count("nz");
    double nm[9];
    normalMatrix(toWorld, nm);
    for (size_t i = 0; i < nVerts; ++i) {
        m.positions.push_back(xfPoint(toWorld, V3{cols["x"][i], cols["y"][i], cols["z"][i]}));
        if (hasN) m.normals.push_back(normalized(xfNormal(nm, V3{cols["nx"][i], cols["ny"][i], cols["nz"][i]})));
    }
    std::vector<std::pair<float, float>> vuv;
    const char *uvNames[2][2] = {{"u", "v"}, {"s", "t"}};
    for (auto &nmz : uvNames)
        if (cols.count(nmz[0]) && cols.count(nmz[1])) { vuv.clear(); for (size_t i = 0; i < nVerts; ++i) vuv.emplace_back(cols[nmz[0]][i], cols[nmz[1]][i]); }
    if (maxSmoothAngle >= 0) rebuildTopology(m, vuv.empty() ? nullptr : &vuv, maxSmoothAngle);
    if (faceNormals) {
        m.normals.clear();
        if (flipNormals) for (size_t t = 0; t + 2 < m.indices.size(); t += 3) std::swap(m.indices[t], m.indices[t + 1]);
    } else if (!m.normals.empty()) {
        if (flipNormals) for (V3 &n : m.normals) n = {-n.x, -n.y, -n.z};
    } else {
        computeNormals(m, flipNormals);
    }
    return m;
}

// ------------------------------------------------------------------------------------------------ the scene
struct LoadedScene {
    SceneData scene;
    Properties integrator;          // guided_path properties as given in the XML
    std::vector<std::string> warnings;
};

class SceneXmlLoader {
public:
    SceneXmlLoader(const std::string &path, const std::map<std::string, std::string> &defines, bool strict = true, int width = 0, int height = 0,
                   const std::string &dataDir = "")
        : m_path(path), m_params(defines), m_strict(strict), m_w(width), m_h(height), m_dataDir(dataDir) {}

    LoadedScene load() {
        std::ifstream f(m_path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open '" + m_path + "'");
        std::stringstream ss; ss << f.rdbuf();
        const std::string text = ss.str();
        XmlNode root = XmlParser(text).parseDocument();
        if (root.tag != "scene") throw std::runtime_error(m_path + ": root element is <" + root.tag + ">, expected <scene>");
        size_t slash = m_path.find_last_of('/');
        m_base = slash == std::string::npos ? "." : m_path.substr(0, slash);
        collectDefaults(root);
        LoadedScene out;
        // integrator
        const XmlNode *integ = root.child("integrator");
        if (!integ) throw std::runtime_error("no <integrator>");
        if (integ->get("type") != "guided_path") throw std::runtime_error("integrator type '" + integ->get("type") + "' is not supported: this path implements 'guided_path' only");
        static const char *known[] = {"nee", "sampleCombination", "spatialFilter", "directionalFilter", "bsdfSamplingFractionLoss", "budgetType", "sdTreeMaxMemory",
                                      "sTreeThreshold", "dTreeThreshold", "bsdfSamplingFraction", "sppPerPass", "budget", "dumpSDTree", "rrDepth", "maxDepth",
                                      "strictNormals", "hideEmitters"};
        for (auto &c : integ->children) {
            if (!c.attr("name")) continue;
            const std::string k = c.get("name"), v = sub(c.get("value"));
            if (std::find_if(std::begin(known), std::end(known), [&](const char *n) { return k == n; }) == std::end(known)) {
                out.warnings.push_back("integrator property '" + k + "' is not used by guided_path");
                continue;
            }
            out.integrator.values[k] = v;
        }
        // sensor
        const XmlNode *sensor = root.child("sensor");
        if (!sensor) throw std::runtime_error("no <sensor>");
        if (sensor->get("type") != "perspective") throw std::runtime_error("sensor type '" + sensor->get("type") + "' is not supported (perspective only)");
        auto sp = props(*sensor);
        const XmlNode *film = sensor->child("film");
        std::map<std::string, std::string> fp;
        if (film) {
            fp = props(*film);
            const XmlNode *rf = film->child("rfilter");
            if (rf && rf->get("type") != "box") throw std::runtime_error("rfilter type '" + rf->get("type") + "' is not supported (box only; hdrfilm's default 'gaussian' neither)");
            if (!rf) out.warnings.push_back("no <rfilter>: Mitsuba would default to gaussian; the box filter is used");
        }
        const int W = m_w ? m_w : (fp.count("width") ? std::stoi(fp["width"]) : 768), H = m_h ? m_h : (fp.count("height") ? std::stoi(fp["height"]) : 576);
        if (!sp.count("fov")) throw std::runtime_error("perspective sensor without 'fov' (focalLength is not supported)");
        Mat4 c2w = Mat4::identity();
        if (const XmlNode *tw = sensor->child("transform")) c2w = transform(*tw);
        std::string axis = sp.count("fovAxis") ? sp["fovAxis"] : "x";
        std::transform(axis.begin(), axis.end(), axis.begin(), ::tolower);
        makeCamera(out.scene.camera, c2w, std::stod(sp["fov"]), axis, sp.count("nearClip") ? std::stod(sp["nearClip"]) : 1e-2,
                   sp.count("farClip") ? std::stod(sp["farClip"]) : 1e4, W, H);
        // bsdfs
        for (auto &b : root.children) if (b.tag == "bsdf" && b.attr("id")) m_byId[b.get("id")] = intern(makeBsdf(b, true, out), out);
        {   // an id may also sit on a NESTED bsdf (KITCHEN references the twosided element inside a bumpmap): every element with an id is a named object
            std::function<void(const XmlNode &)> walk = [&](const XmlNode &n) {
                for (auto &c : n.children) {
                    if (c.tag != "bsdf") continue;
                    if (c.attr("id") && !m_byId.count(c.get("id"))) m_byId[c.get("id")] = intern(makeBsdf(c, true, out), out);
                    walk(c);
                }
            };
            for (auto &b : root.children) if (b.tag == "bsdf") walk(b);
        }
        for (auto &e : root.children) {
            if (e.tag != "emitter") continue;
            if (e.get("type") == "constant" && !out.scene.hasEnvironment && !out.scene.hasEnvmap) { out.scene.hasEnvironment = true; colour(e, "radiance", 1.0f, out.scene.environment); continue; }
            if (e.get("type") == "envmap" && !out.scene.hasEnvironment && !out.scene.hasEnvmap) {  // EnvironmentMap::EnvironmentMap, envmap.cpp:100-190
                auto ep = props(e);
                if (!ep.count("filename")) throw std::runtime_error("envmap emitter without filename");
                std::string fn = ep["filename"];
                if (fn.empty() || fn[0] != '/') fn = m_base + "/" + fn;
                HdrImage img = readHdrImage(fn);
                Mat4 m = Mat4::identity();
                if (const XmlNode *tw = e.child("transform")) m = transform(*tw);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        float d = 0;
                        for (int k = 0; k < 3; ++k) d += m.m[4 * a + k] * m.m[4 * b + k];
                        if (std::fabs(d - (a == b ? 1.0f : 0.0f)) > 1e-4f) throw std::runtime_error("envmap: toWorld must be a rotation");
                        out.scene.envmap.to_world[3 * a + b] = m.m[4 * a + b];
                    }
                out.scene.envmap.width = (uint32_t)img.width; out.scene.envmap.height = (uint32_t)img.height;
                out.scene.envmap.scale = ep.count("scale") ? std::stof(ep["scale"]) : 1.0f;
                out.scene.envmapRgb.swap(img.rgb);
                out.scene.hasEnvmap = true;
                continue;
            }
            if (e.get("type") == "sunsky" && !out.scene.hasEnvironment && !out.scene.hasEnvmap) {
                // SunSkyEmitter (sunsky.cpp:100-235) bakes sun + sky into a radiance map and instantiates `envmap` on it: the same bake here
                // (host/sunsky.h); the Hosek-Wilkie / Preetham tables are read from the operator's Mitsuba source tree (the parent of --data-dir,
                // or PPG_MITSUBA_SRC)
                std::string src = getenv("PPG_MITSUBA_SRC") ? getenv("PPG_MITSUBA_SRC") : "";
                if (src.empty() && !m_dataDir.empty()) { std::string dd = m_dataDir; while (dd.size() > 1 && dd.back() == '/') dd.pop_back(); const size_t sl = dd.find_last_of('/'); src = sl == std::string::npos ? "." : dd.substr(0, sl); }
                std::ifstream probe(src + "/src/emitters/sunsky/skymodeldata.h");
                if (src.empty() || !probe) {
                    if (!m_strict) { out.warnings.push_back("emitter 'sunsky' skipped (no Mitsuba source tree for the sky model's tables)"); continue; }
                    throw std::runtime_error("sunsky: the sky model's coefficient tables (src/emitters/sunsky/skymodeldata.h, sunmodel.h) come with Mitsuba's source tree: "
                                             "pass --data-dir <tree>/data or set PPG_MITSUBA_SRC");
                }
                auto ep = props(e);
                for (const char *k : {"sunDirection", "extend"}) if (ep.count(k) && ep[k] != "false") throw std::runtime_error(std::string("sunsky: ") + k + " is not supported");
                sunsky::Params sp;
                auto num = [&](const char *k, double &v) { if (ep.count(k)) v = std::stod(ep[k]); };
                auto inum = [&](const char *k, int &v) { if (ep.count(k)) v = std::stoi(ep[k]); };
                num("latitude", sp.latitude); num("longitude", sp.longitude); num("timezone", sp.timezone); num("hour", sp.hour); num("minute", sp.minute); num("second", sp.second);
                inum("year", sp.year); inum("month", sp.month); inum("day", sp.day); inum("resolution", sp.resolution);
                num("scale", sp.scale); num("sunScale", sp.sunScale); num("skyScale", sp.skyScale); num("sunRadiusScale", sp.sunRadiusScale);
                num("turbidity", sp.turbidity); num("stretch", sp.stretch);
                for (auto &c : e.children)
                    if (c.attr("name") && c.get("name") == "albedo" && (c.tag == "rgb" || c.tag == "srgb" || c.tag == "spectrum")) {
                        float a3[3]; colour(e, "albedo", 0.2f, a3);
                        for (int k = 0; k < 3; ++k) sp.albedo[k] = a3[k];
                    }
                int ew = 0, eh = 0;
                sunsky::bake(sp, src, out.scene.envmapRgb, ew, eh);
                Mat4 m = Mat4::identity();
                if (const XmlNode *tw = e.child("transform")) m = transform(*tw);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        float d = 0;
                        for (int k = 0; k < 3; ++k) d += m.m[4 * a + k] * m.m[4 * b + k];
                        if (std::fabs(d - (a == b ? 1.0f : 0.0f)) > 1e-4f) throw std::runtime_error("sunsky: toWorld must be a rotation");
                        out.scene.envmap.to_world[3 * a + b] = m.m[4 * a + b];
                    }
                out.scene.envmap.width = (uint32_t)ew; out.scene.envmap.height = (uint32_t)eh; out.scene.envmap.scale = 1.0f;
                out.scene.hasEnvmap = true;
                continue;
            }
            if (!m_strict) { out.warnings.push_back("emitter '" + e.get("type") + "' skipped (not supported)"); continue; }
            throw std::runtime_error("emitter type '" + e.get("type") + "' is not supported (area emitters on shapes and one `constant`, `envmap` or `sunsky` environment emitter; SURVEY.md §8 f2)");
        }
        // shapes
        struct Part { Mesh mesh; uint32_t mat; int em; };
        std::vector<Part> parts;
        int defaultMat = -1;
        for (auto &sh : root.children) {
            if (sh.tag != "shape") continue;
            const std::string t = sh.get("type");
            auto pr = props(sh);
            Mat4 m = Mat4::identity();
            if (const XmlNode *tw = sh.child("transform")) m = transform(*tw);
            std::vector<Mesh> meshes;
            ppg_sphere sphere{};
            bool isSphere = false;
            if (t == "obj") {
                if (!pr.count("filename")) throw std::runtime_error("obj shape without filename");
                if (pr.count("shapeIndex")) throw std::runtime_error("obj: shapeIndex is not supported");
                if (pr.count("maxSmoothAngle") && flag(pr, "faceNormals", false)) throw std::runtime_error("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!");
                std::string fn = pr["filename"];
                if (fn.empty() || fn[0] != '/') fn = m_base + "/" + fn;
                if (!m_strict && !std::ifstream(fn)) { out.warnings.push_back("shape skipped: Wavefront OBJ file '" + fn + "' not found"); continue; }
                meshes = loadOBJ(fn, m, flag(pr, "faceNormals", false), flag(pr, "flipNormals", false), flag(pr, "flipTexCoords", true), flag(pr, "collapse", false),
                                 pr.count("maxSmoothAngle") ? std::stof(pr["maxSmoothAngle"]) : -1.0f);
            } else if (t == "ply") {
                if (!pr.count("filename")) throw std::runtime_error("ply shape without filename");
                std::string fn = pr["filename"];
                if (fn.empty() || fn[0] != '/') fn = m_base + "/" + fn;
                if (!std::ifstream(fn)) {
                    if (m_strict) throw std::runtime_error("PLY file '" + fn + "' not found");
                    out.warnings.push_back("shape skipped: PLY file '" + fn + "' not found");
                    continue;
                }
                if (pr.count("maxSmoothAngle") && flag(pr, "faceNormals", false)) throw std::runtime_error("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!");
                meshes.push_back(loadPLY(fn, m, flag(pr, "faceNormals", false), flag(pr, "flipNormals", false), pr.count("maxSmoothAngle") ? std::stof(pr["maxSmoothAngle"]) : -1.0f));
            } else if (t == "serialized") {
                if (!pr.count("filename")) throw std::runtime_error("serialized shape without filename");
                std::string fn = pr["filename"];
                if (fn.empty() || fn[0] != '/') fn = m_base + "/" + fn;
                if (!std::ifstream(fn)) {
                    if (m_strict) throw std::runtime_error("serialized mesh file '" + fn + "' not found");
                    out.warnings.push_back("shape skipped: serialized mesh file '" + fn + "' not found");
                    continue;
                }
                if (pr.count("maxSmoothAngle") && flag(pr, "faceNormals", false)) throw std::runtime_error("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!");
                meshes.push_back(loadSerialized(fn, m, pr.count("shapeIndex") ? std::stoi(pr["shapeIndex"]) : 0, flag(pr, "faceNormals", false), flag(pr, "flipNormals", false),
                                                pr.count("maxSmoothAngle") ? std::stof(pr["maxSmoothAngle"]) : -1.0f));
            } else if (t == "rectangle") {
                meshes.push_back(rectangle(m, flag(pr, "flipNormals", false)));
            } else if (t == "cube") {
                meshes.push_back(cube(m, flag(pr, "flipNormals", false)));
            } else if (t == "sphere") {  // Sphere::Sphere, sphere.cpp:108-131: the scale of toWorld goes into the radius, the rest stays a rotation
                float c[3] = {0, 0, 0};
                for (auto &pc : sh.children)
                    if (pc.tag == "point" && pc.get("name") == "center") {
                        const char *ax[3] = {"x", "y", "z"};
                        for (int a = 0; a < 3; ++a) if (pc.attr(ax[a])) c[a] = std::stof(sub(pc.get(ax[a])));
                    }
                Mat4 o2w = Mat4::identity();
                o2w.m[3] = c[0]; o2w.m[7] = c[1]; o2w.m[11] = c[2];
                sphere = ppg_sphere{};
                sphere.radius = pr.count("radius") ? std::stof(pr["radius"]) : 1.0f;
                if (sh.child("transform")) {
                    const float scale = std::sqrt(m.m[0] * m.m[0] + m.m[4] * m.m[4] + m.m[8] * m.m[8]);  // objectToWorld(Vector(1, 0, 0)).length()
                    Mat4 sc = Mat4::identity();
                    sc.m[0] = sc.m[5] = sc.m[10] = 1 / scale;
                    o2w = (m * sc) * o2w;
                    sphere.radius *= scale;
                }
                if (!(sphere.radius > 0)) throw std::runtime_error("Cannot create spheres of radius <= 0");
                for (int a = 0; a < 3; ++a) { sphere.center[a] = o2w.m[4 * a + 3]; for (int b = 0; b < 3; ++b) sphere.to_world[3 * a + b] = o2w.m[4 * a + b]; }
                sphere.flip_normals = flag(pr, "flipNormals", false) ? 1 : 0;
                isSphere = true;
            } else {
                throw std::runtime_error("shape type '" + t + "' is not supported (obj, ply, serialized, rectangle, cube, sphere)");
            }
            int mat = -1;
            for (auto &c : sh.children) {
                if (c.tag == "bsdf") mat = (int)intern(makeBsdf(c, true, out), out);
                else if (c.tag == "ref") {
                    auto it = m_byId.find(c.get("id"));
                    if (it == m_byId.end()) throw std::runtime_error("<ref id=\"" + c.get("id") + "\">: no such bsdf");
                    mat = (int)it->second;
                }
            }
            if (mat < 0) {  // Shape::configure (shape.cpp:48-72): all-absorbing under an emitter, otherwise a 0.5 Lambertian "for convenience"
                if (sh.child("emitter")) {
                    ppg_material d{}; d.type = PPG_BSDF_DIFFUSE; defaults(d); d.reflectance[0] = d.reflectance[1] = d.reflectance[2] = 0.0f;
                    mat = (int)intern(d, out);
                } else {
                    if (defaultMat < 0) { ppg_material d{}; d.type = PPG_BSDF_DIFFUSE; d.reflectance[0] = d.reflectance[1] = d.reflectance[2] = 0.5f; defaults(d); defaultMat = (int)intern(d, out); }
                    mat = defaultMat;
                }
            }
            int em = -1;
            if (const XmlNode *e = sh.child("emitter")) {
                if (e->get("type") != "area") throw std::runtime_error("emitter type '" + e->get("type") + "' on a shape is not supported (area only)");
                if (meshes.size() > 1) throw std::runtime_error("Cannot attach an emitter to an OBJ file containing multiple objects!");
                em = (int)out.scene.emitters.size();
                ppg_emitter pe{};
                colour(*e, "radiance", 1.0f, pe.radiance);
                out.scene.emitters.push_back(pe);
            }
            for (auto &mm : meshes) parts.push_back(Part{std::move(mm), (uint32_t)mat, em});
            if (isSphere) { sphere.material = (uint32_t)mat; sphere.emitter = em; out.scene.spheres.push_back(sphere); }
        }
        if (parts.empty()) throw std::runtime_error("scene without shapes");
        bool anyNormals = false;
        for (auto &p : parts) anyNormals |= !p.mesh.normals.empty();
        SceneData &S = out.scene;
        for (auto &p : parts) {
            Mesh &m = p.mesh;
            if (anyNormals && m.normals.empty()) {  // un-share the vertices, write the face normals out
                Mesh u;
                for (size_t t = 0; t + 2 < m.indices.size(); t += 3) {
                    V3 a = m.positions[m.indices[t]], b = m.positions[m.indices[t + 1]], c = m.positions[m.indices[t + 2]];
                    V3 n = cross(b - a, c - a);
                    float l = std::sqrt(dot(n, n));
                    if (l != 0) n = {n.x / l, n.y / l, n.z / l};
                    for (V3 q : {a, b, c}) { u.indices.push_back((uint32_t)u.positions.size()); u.positions.push_back(q); u.normals.push_back(n); }
                }
                m = std::move(u);
            }
            const uint32_t base = (uint32_t)(S.positions.size() / 3);
            for (size_t i = 0; i < m.positions.size(); ++i) {
                S.positions.insert(S.positions.end(), {m.positions[i].x, m.positions[i].y, m.positions[i].z});
                if (anyNormals) S.normals.insert(S.normals.end(), {m.normals[i].x, m.normals[i].y, m.normals[i].z});
            }
            for (uint32_t id : m.indices) S.indices.push_back(base + id);
            for (size_t t = 0; t < m.indices.size() / 3; ++t) { S.triMaterial.push_back(p.mat); S.triEmitter.push_back(p.em); }
        }
        out.warnings.insert(out.warnings.end(), m_lenientNotes.begin(), m_lenientNotes.end());
        return out;
    }

private:
    std::string m_path, m_base;
    std::map<std::string, std::string> m_params;
    bool m_strict;
    int m_w, m_h;
    std::string m_dataDir;  // Mitsuba `data` directory (roughplastic)
    mutable std::map<std::string, int> m_rtIndex;
    mutable std::vector<std::string> m_lenientNotes;  // warnings raised inside const helpers  // (distribution, alpha, eta) → slice of out.scene.rtrans
    std::map<std::string, uint32_t> m_byId;
    std::vector<std::string> m_matKeys;

    void collectDefaults(const XmlNode &n) {
        if (n.tag == "default" && n.attr("name") && !m_params.count(n.get("name"))) m_params[n.get("name")] = n.get("value");
        for (auto &c : n.children) collectDefaults(c);
    }
    std::string sub(const std::string &t) const {
        if (t.find('$') == std::string::npos) return t;
        std::string o;
        for (size_t i = 0; i < t.size();) {
            if (t[i] != '$') { o += t[i++]; continue; }
            size_t j = i + 1;
            while (j < t.size() && (std::isalnum((unsigned char)t[j]) || t[j] == '_')) ++j;
            const std::string k = t.substr(i + 1, j - i - 1);
            auto it = m_params.find(k);
            if (it == m_params.end()) throw std::runtime_error("undefined parameter $" + k + " (pass it with -D " + k + "=...)");
            o += it->second;
            i = j;
        }
        return o;
    }
    std::map<std::string, std::string> props(const XmlNode &e) const {
        std::map<std::string, std::string> o;
        for (auto &c : e.children)
            if ((c.tag == "boolean" || c.tag == "integer" || c.tag == "float" || c.tag == "string") && c.attr("name")) o[c.get("name")] = sub(c.get("value"));
        return o;
    }
    static bool flag(const std::map<std::string, std::string> &p, const std::string &k, bool d) {
        auto it = p.find(k);
        if (it == p.end()) return d;
        std::string v = it->second;
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        return v == "true" || v == "1";
    }
    void colour(const XmlNode &e, const std::string &name, float dflt, float rgb[3]) const {
        for (auto &c : e.children) {
            if (c.get("name") != name) continue;
            if (c.tag == "rgb" || c.tag == "srgb" || c.tag == "spectrum") {
                if (c.attr("filename")) {  // InterpolatedSpectrum(path) (spectrum.cpp:575-600): "wavelength value" lines, # comments
                    std::string fn = sub(c.get("filename"));
                    if (fn.empty() || fn[0] != '/') fn = m_base + "/" + fn;
                    if (c.tag != "spectrum" || c.attr("value")) throw std::runtime_error("<" + c.tag + " filename>: please provide one of 'value' or 'filename'");
                    std::vector<double> lam, val;
                    if (!readSPD(fn, lam, val)) throw std::runtime_error("<spectrum filename=\"" + c.get("filename") + "\">: file not found");
                    if (lam.size() < 2) throw std::runtime_error("<spectrum filename>: fewer than two samples");
                    interpolatedToRGB(lam, val, rgb);
                    return;
                }
                parseColour(c.tag, sub(c.get("value")), rgb);
                return;
            }
            if (c.tag == "blackbody") {
                std::string t = sub(c.get("temperature"));
                while (!t.empty() && isspace((unsigned char)t.back())) t.pop_back();
                if (!t.empty() && (t.back() == 'K' || t.back() == 'k')) t.pop_back();
                blackbodyToRGB(std::stod(t), c.attr("scale") ? std::stod(sub(c.get("scale"))) : 1.0, rgb);
                return;
            }
            if (c.tag == "texture" || c.tag == "ref") {
                if (m_strict) throw std::runtime_error("textured '" + name + "' is not supported (SURVEY.md §8 f1)");
                m_lenientNotes.push_back("texture on '" + name + "' ignored: the plug-in's default value is used");
                break;
            }
        }
        rgb[0] = rgb[1] = rgb[2] = dflt;
    }
    bool hasChildNamed(const XmlNode &e, const std::string &name) const {
        for (auto &c : e.children) if (c.get("name") == name) return true;
        return false;
    }
    Mat4 transform(const XmlNode &e) const {
        Mat4 m = Mat4::identity();
        for (auto &c : e.children) {
            auto g = [&](const char *k, float d) { auto *a = c.attr(k); return a ? std::stof(sub(*a)) : d; };
            Mat4 t = Mat4::identity();
            if (c.tag == "translate") { t.m[3] = g("x", 0); t.m[7] = g("y", 0); t.m[11] = g("z", 0); }
            else if (c.tag == "scale") {
                if (c.attr("value")) { t.m[0] = t.m[5] = t.m[10] = g("value", 1); }
                else { t.m[0] = g("x", 1); t.m[5] = g("y", 1); t.m[10] = g("z", 1); }
            } else if (c.tag == "rotate") {  // Transform::rotate, transform.cpp:65-97
                V3 a = normalized(V3{g("x", 0), g("y", 0), g("z", 0)});
                const float th = g("angle", 0) * (float)(3.14159265358979323846 / 180.0), s = std::sin(th), co = std::cos(th);
                t.m[0] = a.x * a.x + (1 - a.x * a.x) * co; t.m[1] = a.x * a.y * (1 - co) - a.z * s; t.m[2] = a.x * a.z * (1 - co) + a.y * s;
                t.m[4] = a.x * a.y * (1 - co) + a.z * s; t.m[5] = a.y * a.y + (1 - a.y * a.y) * co; t.m[6] = a.y * a.z * (1 - co) - a.x * s;
                t.m[8] = a.x * a.z * (1 - co) - a.y * s; t.m[9] = a.y * a.z * (1 - co) + a.x * s; t.m[10] = a.z * a.z + (1 - a.z * a.z) * co;
            } else if (c.tag == "lookAt" || c.tag == "lookat") {  // Transform::lookAt, transform.cpp:191-214
                auto o = parseFloats(sub(c.get("origin"))), tg = parseFloats(sub(c.get("target")));
                std::vector<double> up = c.attr("up") ? parseFloats(sub(c.get("up"))) : std::vector<double>();
                if (o.size() != 3 || tg.size() != 3) throw std::runtime_error("<lookAt> needs origin and target");
                V3 p{(float)o[0], (float)o[1], (float)o[2]}, d = normalized(V3{(float)tg[0], (float)tg[1], (float)tg[2]} - p);
                V3 u = up.size() == 3 ? V3{(float)up[0], (float)up[1], (float)up[2]} : cross(d, std::fabs(d.x) < std::fabs(d.y) ? V3{1, 0, 0} : V3{0, 1, 0});
                V3 left = normalized(cross(u, d)), nu = cross(d, left);
                t.m[0] = left.x; t.m[4] = left.y; t.m[8] = left.z; t.m[1] = nu.x; t.m[5] = nu.y; t.m[9] = nu.z;
                t.m[2] = d.x; t.m[6] = d.y; t.m[10] = d.z; t.m[3] = p.x; t.m[7] = p.y; t.m[11] = p.z;
            } else if (c.tag == "matrix") {
                auto v = parseFloats(sub(c.get("value")));
                if (v.size() != 16) throw std::runtime_error("<matrix> needs 16 values");
                for (int i = 0; i < 16; ++i) t.m[i] = (float)v[i];
            } else {
                throw std::runtime_error("unsupported transform element <" + c.tag + ">");
            }
            m = t * m;  // later elements are applied after earlier ones
        }
        return m;
    }
    static Mesh rectangle(const Mat4 &toWorld, bool flip) {  // shapes/rectangle.cpp:170-203
        Mat4 m = toWorld;
        if (flip) { Mat4 s = Mat4::identity(); s.m[10] = -1; m = m * s; }
        Mesh r;
        const V3 c[4] = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
        double nm[9];
        normalMatrix(m, nm);
        V3 n = normalized(xfNormal(nm, V3{0, 0, 1}));
        for (const V3 &p : c) { r.positions.push_back(xfPoint(m, p)); r.normals.push_back(n); }
        r.indices = {0, 1, 2, 2, 3, 0};
        return r;
    }
    // shapes/cube.cpp:24-30, 73-103: [-1, 1]^3 as 6 faces x 4 vertices with their own normals; face order -y, +y, +x, +z, -x, -z, corners
    // from the start corner counter-clockwise about the normal, triangles (0, 1, 2), (3, 0, 2) per face
    static Mesh cube(const Mat4 &toWorld, bool flip) {
        static const int faces[6][6] = {{0, -1, 0, 1, -1, -1}, {0, 1, 0, 1, 1, -1}, {1, 0, 0, 1, -1, -1}, {0, 0, 1, 1, -1, 1}, {-1, 0, 0, -1, -1, 1}, {0, 0, -1, 1, 1, -1}};
        Mesh r;
        double nm[9];
        normalMatrix(toWorld, nm);
        for (int f = 0; f < 6; ++f) {
            const V3 n{(float)faces[f][0], (float)faces[f][1], (float)faces[f][2]};
            V3 p{(float)faces[f][3], (float)faces[f][4], (float)faces[f][5]};
            V3 nw = normalized(xfNormal(nm, n));
            if (flip) nw = {-nw.x, -nw.y, -nw.z};
            for (int k = 0; k < 4; ++k) {
                r.positions.push_back(xfPoint(toWorld, p)); r.normals.push_back(nw);
                const V3 c = cross(n, p);
                const float d = dot(n, p);
                p = {c.x + d * n.x, c.y + d * n.y, c.z + d * n.z};  // a quarter turn about the normal
            }
            const uint32_t b = 4 * (uint32_t)f;
            for (uint32_t id : {b, b + 1, b + 2, b + 3, b, b + 2}) r.indices.push_back(id);
        }
        return r;
    }
    static void makeCamera(ppg_camera &cam, const Mat4 &c2w, double fov, const std::string &axisIn, double nearC, double farC, int W, int H) {
        // sensor.cpp:239-264, 301-305; perspective.cpp:150-164 (m_sampleToCamera), inverted in closed form
        const double aspect = (double)W / H, PI = 3.14159265358979323846;
        std::string axis = axisIn;
        if (axis == "smaller") axis = aspect > 1 ? "y" : "x";
        else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
        double xfov;
        if (axis == "x") xfov = fov;
        else if (axis == "y") xfov = 2 * std::atan(std::tan(0.5 * fov * PI / 180) * aspect) * 180 / PI;
        else if (axis == "diagonal") { double diag = 2 * std::tan(0.5 * fov * PI / 180), w = diag / std::sqrt(1 + 1 / (aspect * aspect)); xfov = 2 * std::atan(w * 0.5) * 180 / PI; }
        else throw std::runtime_error("fovAxis '" + axisIn + "' not supported");
        const double cot = 1 / std::tan(xfov / 2 * PI / 180);
        double inv[16] = {0};
        inv[0] = -2 / cot; inv[3] = 1 / cot;
        inv[5] = -2 / (aspect * cot); inv[7] = 1 / (aspect * cot);
        inv[11] = 1;
        inv[14] = -(farC - nearC) / (nearC * farC); inv[15] = 1 / nearC;
        for (int i = 0; i < 16; ++i) { cam.sample_to_camera[i] = (float)inv[i]; cam.camera_to_world[i] = c2w.m[i]; }
        cam.near_clip = (float)nearC; cam.far_clip = (float)farC; cam.width = W; cam.height = H;
    }
    static void defaults(ppg_material &m) {  // what ppg_host.bindings.Material.from_dict fills in
        for (int k = 0; k < 3; ++k) { m.specular[k] = 1.0f; m.k[k] = 1.0f; }
        m.alpha = 0.1f;
        const bool cond = m.type == PPG_BSDF_MIRROR || m.type == PPG_BSDF_CONDUCTOR || m.type == PPG_BSDF_ROUGHCONDUCTOR;
        for (int k = 0; k < 3; ++k) m.eta[k] = cond ? 0.0f : 1.5046f;
    }
    // MicrofacetDistribution(props), microfacet.h:99-145: distribution name (flag set on m), isotropic alpha
    std::string microfacet(std::map<std::string, std::string> &p, const std::string &t, ppg_material &m) const {
        std::string distr = p.count("distribution") ? p["distribution"] : "beckmann";
        std::transform(distr.begin(), distr.end(), distr.begin(), ::tolower);
        if (distr != "ggx" && distr != "beckmann") throw std::runtime_error(t + ": distribution '" + distr + "' is not supported (ggx, beckmann)");
        if (distr == "beckmann") m.flags |= PPG_MAT_BECKMANN;  // Mitsuba's default (microfacet.h:99)
        if (p.count("alphaU") || p.count("alphaV")) {
            if (!p.count("alphaU") || !p.count("alphaV") || std::stof(p["alphaU"]) != std::stof(p["alphaV"])) throw std::runtime_error(t + ": anisotropic roughness is not supported");
            m.alpha = std::stof(p["alphaU"]);
        } else m.alpha = p.count("alpha") ? std::stof(p["alpha"]) : 0.1f;
        if (p.count("sampleVisible") && !flag(p, "sampleVisible", true)) throw std::runtime_error(t + ": sampleVisible=false is not supported");
        return distr;
    }
    double lookupIOR(const std::map<std::string, std::string> &p, const std::string &name, const std::string &dflt) const {  // ior.h:95-111
        static const std::map<std::string, double> ior = {{"vacuum", 1.0}, {"air", 1.000277}, {"water", 1.3330}, {"polypropylene", 1.49}, {"bk7", 1.5046}, {"diamond", 2.419}};
        std::string v = p.count(name) ? p.at(name) : dflt;
        char *end = nullptr;
        double d = std::strtod(v.c_str(), &end);
        if (end && *end == '\0' && end != v.c_str()) return d;
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        auto it = ior.find(v);
        if (it == ior.end()) throw std::runtime_error("IOR name '" + v + "' is not in the built-in list; give a number");
        return it->second;
    }
    void conductorIOR(const XmlNode &e, const std::map<std::string, std::string> &p, const std::string &what, ppg_material &m) const {
        const float ext = (float)lookupIOR(p, "extEta", "air");
        std::string mat = p.count("material") ? p.at("material") : "Cu";
        std::transform(mat.begin(), mat.end(), mat.begin(), ::tolower);
        float eta[3] = {0, 0, 0}, k[3] = {1, 1, 1};
        if (mat != "none") {
            const std::string name = p.count("material") ? p.at("material") : "Cu";
            if (!hasChildNamed(e, "eta") || !hasChildNamed(e, "k")) {  // the measured spectra ship with Mitsuba (data/ior/<name>.{eta,k}.spd)
                std::string dir = m_dataDir;
                if (dir.empty()) { const char *ev = std::getenv("PPG_MITSUBA_DATA"); if (ev) dir = ev; }
                std::vector<double> le, ve, lk, vk;
                if (dir.empty() || !readSPD(dir + "/ior/" + name + ".eta.spd", le, ve) || !readSPD(dir + "/ior/" + name + ".k.spd", lk, vk) || le.size() < 2 || lk.size() < 2)
                    throw std::runtime_error(what + "(material=" + name + "): the measured IOR spectra data/ior/" + name + ".{eta,k}.spd come with Mitsuba: pass --data-dir / PPG_MITSUBA_DATA, or give eta and k");
                interpolatedToRGB(le, ve, eta, false, false);
                interpolatedToRGB(lk, vk, k, false, false);
            }
            if (hasChildNamed(e, "eta")) colour(e, "eta", 0.0f, eta);
            if (hasChildNamed(e, "k")) colour(e, "k", 1.0f, k);
        }
        for (int c = 0; c < 3; ++c) { m.eta[c] = eta[c] / ext; m.k[c] = k[c] / ext; }
    }
    ppg_material makeBsdf(const XmlNode &e, bool allowWrap, LoadedScene &out) const {
        std::string t = e.get("type");
        auto p = props(e);
        ppg_material m{};
        auto inner = [&]() { std::vector<const XmlNode *> v; for (auto &c : e.children) if (c.tag == "bsdf") v.push_back(&c); return v; };
        if (t == "diffuse") { m.type = PPG_BSDF_DIFFUSE; defaults(m); colour(e, "reflectance", 0.5f, m.reflectance); return m; }
        if (t == "twosided" && allowWrap) {
            auto in = inner();
            if (in.size() == 1) {
                m = makeBsdf(*in[0], false, out);
                if (m.type == PPG_BSDF_DIELECTRIC || m.type == PPG_BSDF_THINDIELECTRIC || m.type == PPG_BSDF_ROUGHDIELECTRIC) throw std::runtime_error("twosided(dielectric): only BRDFs can be two-sided (twosided.cpp:84-88)");
                if (m.type == PPG_BSDF_DIFFUSE && !(m.flags & PPG_MAT_MASK)) { m.type = PPG_BSDF_TWOSIDED_DIFFUSE; return m; }
                m.flags |= PPG_MAT_TWOSIDED;
                return m;
            }
            t = "twosided(...)";
        } else if (t == "mask" && allowWrap) {
            auto in = inner();
            if (in.size() == 1) {
                m = makeBsdf(*in[0], true, out);
                if (m.flags & PPG_MAT_MASK) throw std::runtime_error("mask(mask(...)) is not supported");
                if (m.type == PPG_BSDF_TWOSIDED_DIFFUSE) { m.type = PPG_BSDF_DIFFUSE; m.flags |= PPG_MAT_TWOSIDED; }
                m.flags |= PPG_MAT_MASK;
                colour(e, "opacity", 0.5f, m.opacity);
                return m;
            }
            t = "mask(...)";
        } else if (t == "conductor") {
            std::string mat = p.count("material") ? p["material"] : "Cu";
            std::transform(mat.begin(), mat.end(), mat.begin(), ::tolower);
            m.type = mat == "none" ? PPG_BSDF_MIRROR : PPG_BSDF_CONDUCTOR;
            defaults(m);
            colour(e, "specularReflectance", 1.0f, m.reflectance);
            if (mat != "none") conductorIOR(e, p, t, m);
            return m;
        } else if (t == "roughconductor" || t == "roughdielectric") {
            if (t == "roughconductor") {
                m.type = PPG_BSDF_ROUGHCONDUCTOR; defaults(m);
                colour(e, "specularReflectance", 1.0f, m.reflectance);
                conductorIOR(e, p, t, m);
            } else {  // roughdielectric.cpp:183-211
                m.type = PPG_BSDF_ROUGHDIELECTRIC; defaults(m);
                colour(e, "specularReflectance", 1.0f, m.reflectance); colour(e, "specularTransmittance", 1.0f, m.specular);
                const double intIOR = lookupIOR(p, "intIOR", "bk7"), extIOR = lookupIOR(p, "extIOR", "air");
                if (intIOR < 0 || extIOR < 0 || intIOR == extIOR) throw std::runtime_error(t + ": the interior and exterior indices of refraction must be positive and differ");
                m.eta[0] = m.eta[1] = m.eta[2] = (float)(intIOR / extIOR);
            }
            microfacet(p, t, m);
            return m;
        } else if (t == "roughplastic") {  // roughplastic.cpp:197-227, 285-305
            m.type = PPG_BSDF_ROUGHPLASTIC; defaults(m);
            colour(e, "diffuseReflectance", 0.5f, m.reflectance); colour(e, "specularReflectance", 1.0f, m.specular);
            const double intIOR = lookupIOR(p, "intIOR", "polypropylene"), extIOR = lookupIOR(p, "extIOR", "air");
            if (intIOR < 0 || extIOR < 0 || intIOR == extIOR) throw std::runtime_error(t + ": the interior and exterior indices of refraction must be positive and differ");
            m.eta[0] = m.eta[1] = m.eta[2] = (float)(intIOR / extIOR);
            if (flag(p, "nonlinear", false)) m.flags |= PPG_MAT_NONLINEAR;
            const std::string distr = microfacet(p, t, m);
            char key[96];
            snprintf(key, sizeof key, "%s/%.9g/%.9g", distr.c_str(), m.alpha, m.eta[0]);
            auto it = m_rtIndex.find(key);
            if (it == m_rtIndex.end()) {
                std::vector<float> slice;
                try { slice = roughplasticSlice(distr, m.alpha, m.eta[0], m_dataDir); }
                catch (const std::exception &ex) { throw std::runtime_error(t + ": " + ex.what()); }
                out.scene.rtransSamples = (uint32_t)slice.size() - 1;
                it = m_rtIndex.emplace(key, (int)(out.scene.rtrans.size() / slice.size())).first;
                out.scene.rtrans.insert(out.scene.rtrans.end(), slice.begin(), slice.end());
            }
            m.rtrans = it->second;
            return m;
        } else if (t == "plastic" || t == "dielectric" || t == "thindielectric") {
            m.type = t == "plastic" ? PPG_BSDF_PLASTIC : (t == "dielectric" ? PPG_BSDF_DIELECTRIC : PPG_BSDF_THINDIELECTRIC);
            defaults(m);
            const float eta = (float)(lookupIOR(p, "intIOR", t == "plastic" ? "polypropylene" : "bk7") / lookupIOR(p, "extIOR", "air"));
            m.eta[0] = m.eta[1] = m.eta[2] = eta;
            if (t == "plastic") {
                colour(e, "diffuseReflectance", 0.5f, m.reflectance); colour(e, "specularReflectance", 1.0f, m.specular);
                if (flag(p, "nonlinear", false)) m.flags |= PPG_MAT_NONLINEAR;
            } else {
                colour(e, "specularReflectance", 1.0f, m.reflectance); colour(e, "specularTransmittance", 1.0f, m.specular);
            }
            return m;
        }
        if (!m_strict && (t == "bumpmap" || t == "coating" || t == "roughcoating" || t == "normalmap")) {  // adapters around one nested BSDF: render the nested one
            auto in = inner();
            if (in.size() == 1) { out.warnings.push_back("bsdf '" + t + "' dropped around its nested bsdf"); return makeBsdf(*in[0], allowWrap, out); }
        }
        if (m_strict) throw std::runtime_error("bsdf type '" + t + "' is not supported yet (diffuse, conductor, roughconductor, plastic, roughplastic, dielectric, thindielectric, roughdielectric, mask, twosided; SURVEY.md §8 f1)");
        out.warnings.push_back("bsdf '" + t + "' replaced by diffuse(0.5)");
        m = ppg_material{}; m.type = PPG_BSDF_DIFFUSE; defaults(m);
        m.reflectance[0] = m.reflectance[1] = m.reflectance[2] = 0.5f;
        return m;
    }
public:
    // The <bsdf> element carrying `id` (at any nesting depth: every element with an id is a named object in Mitsuba) as a ppg_material;
    // rough-transmittance slices it needs are appended to out.scene.rtrans.  Used by the Mitsuba plug-in shim
    // (mitsuba_plugin/guided_path_hip.cpp), which cannot reach the nested BSDF of an adapter through Mitsuba's API.
    bool bsdfById(const std::string &id, LoadedScene &out, ppg_material &m) {
        std::ifstream f(m_path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open '" + m_path + "'");
        std::stringstream ss; ss << f.rdbuf();
        const std::string text = ss.str();
        XmlNode root = XmlParser(text).parseDocument();
        size_t slash = m_path.find_last_of('/');
        m_base = slash == std::string::npos ? "." : m_path.substr(0, slash);
        collectDefaults(root);
        const XmlNode *found = nullptr;
        std::function<void(const XmlNode &)> walk = [&](const XmlNode &n) {
            for (auto &c : n.children) {
                if (!found && c.tag == "bsdf" && c.attr("id") && c.get("id") == id) found = &c;
                if (!found) walk(c);
            }
        };
        walk(root);
        if (!found) return false;
        m = makeBsdf(*found, true, out);
        return true;
    }
    // The ppg_material of a <bsdf> element that is not in a file: the plug-in shim builds one from a flat plug-in's Properties (a BSDF
    // without an id cannot be looked up in the scene file) and gets exactly what the scene loader makes of the same parameters.
    ppg_material bsdfFromElement(const XmlNode &e, LoadedScene &out) {
        m_base = ".";
        return makeBsdf(e, true, out);
    }
private:
    uint32_t intern(const ppg_material &m, LoadedScene &out) {
        std::string key((const char *)&m, sizeof m);
        for (size_t i = 0; i < m_matKeys.size(); ++i) if (m_matKeys[i] == key) return (uint32_t)i;
        m_matKeys.push_back(key);
        out.scene.materials.push_back(m);
        return (uint32_t)m_matKeys.size() - 1;
    }
};

namespace xml {
// one <bsdf id="..."> of a scene file as a ppg_material appended to `data` (its rough-transmittance slice, if any, to data.rtrans)
// (name, tag, value) triples of a flat BSDF plug-in's parameters → its ppg_material, through the scene loader's own <bsdf> handling:
// tag = float | integer | boolean | string | rgb, value as it would stand in the scene file ("0.1, 0.2, 0.3" for rgb)
struct BsdfParam { std::string name, tag, value; };
inline bool bsdfFromProperties(const std::string &plugin, const std::vector<BsdfParam> &params, const std::string &dataDir, SceneData &data, ppg_material &m, std::string &why);

inline bool bsdfById(const std::string &scenePath, const std::string &id, const std::string &dataDir, SceneData &data, ppg_material &m, std::string &why) {
    try {
        SceneXmlLoader loader(scenePath, {}, true, 0, 0, dataDir);
        LoadedScene tmp;
        if (!loader.bsdfById(id, tmp, m)) return false;
        if (!tmp.scene.rtrans.empty()) {
            if (data.rtransSamples && data.rtransSamples != tmp.scene.rtransSamples) { why = "rough-transmittance tables of different resolutions"; return false; }
            const uint32_t have = data.rtransSamples ? (uint32_t)(data.rtrans.size() / (data.rtransSamples + 1)) : 0u;
            data.rtransSamples = tmp.scene.rtransSamples;
            data.rtrans.insert(data.rtrans.end(), tmp.scene.rtrans.begin(), tmp.scene.rtrans.end());
            if (m.type == PPG_BSDF_ROUGHPLASTIC) m.rtrans += (int32_t)have;
        }
        return true;
    } catch (const std::exception &e) {
        why = e.what();
        return false;
    }
}

inline bool bsdfFromProperties(const std::string &plugin, const std::vector<BsdfParam> &params, const std::string &dataDir, SceneData &data, ppg_material &m, std::string &why) {
    try {
        SceneXmlLoader loader("", {}, true, 0, 0, dataDir);
        LoadedScene tmp;
        XmlNode e;
        e.tag = "bsdf"; e.attrs.push_back({"type", plugin});
        for (const BsdfParam &p : params) {
            XmlNode c;
            c.tag = p.tag; c.attrs.push_back({"name", p.name}); c.attrs.push_back({"value", p.value});
            e.children.push_back(c);
        }
        m = loader.bsdfFromElement(e, tmp);
        if (!tmp.scene.rtrans.empty()) {
            if (data.rtransSamples && data.rtransSamples != tmp.scene.rtransSamples) { why = "rough-transmittance tables of different resolutions"; return false; }
            const uint32_t have = data.rtransSamples ? (uint32_t)(data.rtrans.size() / (data.rtransSamples + 1)) : 0u;
            data.rtransSamples = tmp.scene.rtransSamples;
            data.rtrans.insert(data.rtrans.end(), tmp.scene.rtrans.begin(), tmp.scene.rtrans.end());
            if (m.type == PPG_BSDF_ROUGHPLASTIC) m.rtrans += (int32_t)have;
        }
        return true;
    } catch (const std::exception &e) {
        why = e.what();
        return false;
    }
}
}  // namespace xml

}  // namespace ppg
