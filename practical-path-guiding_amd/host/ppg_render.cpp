/*
 * ppg_render — stand-alone C++ driver of the guided path tracer (the role `mitsuba scene.xml` plays for
 * plugins/guided_path.so; flags follow mitsuba.cpp:154-250 where they apply).
 *
 *   ppg_render [-D key=value]... [-o out.pfm] [-q] scene.xml      (Mitsuba scene XML, supported subset: host/scene_xml.h)
 *   ppg_render [-D key=value]... [-o out.pfm] [-q] scene.ppgs
 *   ppg_render --cbox WIDTHxHEIGHT [-D key=value]... [-o out.pfm]
 * A Mitsuba scene XML is converted first: `python -m ppg_host scene.xml --ppgs scene.ppgs` (also writes scene.ppgs.props,
 * the XML's integrator properties, which are picked up here).
 *
 * scene.ppgs is the flat binary scene written by ppg_host.scenes.save_scene() (header "PPGS", then the arrays of
 * include/ppg.h's ppg_scene).  --cbox builds scenes/cbox/cbox.xml procedurally like ppg_host.scenes.cbox_scene.
 * -D sets integrator properties (reference names: budget, budgetType, sppPerPass, sTreeThreshold, ...).
 * Output: PFM (little endian RGB float), the log lines of the reference on stdout.
 */
#include <cstdarg>
#include <cstring>
#include <fstream>
#include <memory>
#include <iostream>

#include "guided_path_hip.h"
#include "plugin_core.h"
#include "rccl_reducer.h"
#include "scene_xml.h"

using namespace ppg;

static bool loadSceneChecked(const char *path, SceneData &s);
// a corrupt or truncated file is "cannot load scene", never an allocation of whatever size its header claims
static bool loadScene(const char *path, SceneData &s) {
    try { return loadSceneChecked(path, s); } catch (const std::exception &) { return false; }
}
static bool loadSceneChecked(const char *path, SceneData &s) {
    std::ifstream f(path, std::ios::binary);
    f.seekg(0, std::ios::end);
    const uint64_t fileSize = f ? (uint64_t)f.tellg() : 0;
    f.seekg(0);
    // every count read from the file is checked against what is left of it before anything is sized by it
    auto fits = [&](uint64_t bytes) { const std::streamoff p = f.tellg(); return p >= 0 && bytes <= fileSize - (uint64_t)p; };
    char magic[4];
    uint32_t hdr[6];
    if (!f.read(magic, 4) || memcmp(magic, "PPGS", 4) != 0 || !f.read((char *)hdr, sizeof hdr)) return false;
    const uint32_t nv = hdr[0], nt = hdr[1], nm = hdr[2], ne = hdr[3], hasN = hdr[4];
    if (!fits((uint64_t)nv * 12 * (hasN ? 2 : 1) + (uint64_t)nt * 20 + (uint64_t)nm * sizeof(ppg_material) + (uint64_t)ne * sizeof(ppg_emitter) + sizeof(ppg_camera))) return false;
    s.positions.resize(3 * (size_t)nv); f.read((char *)s.positions.data(), s.positions.size() * 4);
    if (hasN) { s.normals.resize(3 * (size_t)nv); f.read((char *)s.normals.data(), s.normals.size() * 4); }
    s.indices.resize(3 * (size_t)nt); f.read((char *)s.indices.data(), s.indices.size() * 4);
    s.triMaterial.resize(nt); f.read((char *)s.triMaterial.data(), (size_t)nt * 4);
    s.triEmitter.resize(nt); f.read((char *)s.triEmitter.data(), (size_t)nt * 4);
    s.materials.resize(nm); f.read((char *)s.materials.data(), (size_t)nm * sizeof(ppg_material));
    s.emitters.resize(ne); f.read((char *)s.emitters.data(), (size_t)ne * sizeof(ppg_emitter));
    f.read((char *)&s.camera, sizeof(ppg_camera));
    s.hasEnvironment = (hdr[5] & 1) != 0;  // hdr[5]: optional blocks, bit 0 environment, bit 1 rtrans, bit 2 spheres, bit 3 envmap
    if (s.hasEnvironment) f.read((char *)s.environment, 12);
    if (hdr[5] & 2) {
        uint32_t rt[2];
        f.read((char *)rt, 8);
        if (!f || rt[1] < 2 || !fits((uint64_t)rt[0] * ((uint64_t)rt[1] + 1) * 4)) return false;
        s.rtransSamples = rt[1];
        s.rtrans.resize((size_t)rt[0] * (rt[1] + 1)); f.read((char *)s.rtrans.data(), s.rtrans.size() * 4);
    }
    if (hdr[5] & 4) {
        uint32_t n = 0;
        f.read((char *)&n, 4);
        if (!f || !fits((uint64_t)n * sizeof(ppg_sphere))) return false;
        s.spheres.resize(n); f.read((char *)s.spheres.data(), (size_t)n * sizeof(ppg_sphere));
    }
    if (hdr[5] & 8) {
        uint32_t wh[2];
        f.read((char *)wh, 8); f.read((char *)&s.envmap.scale, 4); f.read((char *)s.envmap.to_world, 36);
        if (!f || wh[0] == 0 || wh[1] == 0 || wh[0] > 0x7fffu || wh[1] > 0x7fffu || !fits((uint64_t)wh[0] * wh[1] * 12)) return false;
        s.envmap.width = wh[0]; s.envmap.height = wh[1]; s.hasEnvmap = true;
        s.envmapRgb.resize((size_t)wh[0] * wh[1] * 3); f.read((char *)s.envmapRgb.data(), s.envmapRgb.size() * 4);
    }
    if ((hdr[5] & 16) && !fits((uint64_t)nv * 8)) return false;
    if (hdr[5] & 16) { s.texcoords.resize(2 * (size_t)nv); f.read((char *)s.texcoords.data(), s.texcoords.size() * 4); }  // bit 4: texture coordinates
    if (hdr[5] & 32) {  // bit 5: bitmap textures (pixels as float32 RGB, or as the 8-bit sRGB source decoded through the conversion's 256-entry table)
        uint32_t n = 0;
        f.read((char *)&n, 4);
        if (!f) return false;
        float srgb[256];
        for (int i = 0; i < 256; ++i) { const double v = i / 255.0; srgb[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4)); }
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t wh[2], storage = 0; float sc[4]; int32_t wr[3];
            f.read((char *)wh, 8); f.read((char *)sc, 16); f.read((char *)wr, 12); f.read((char *)&storage, 4);
            // (ppg_set_scene accepts textures below 32768 x 32768)
            if (!f || wh[0] == 0 || wh[1] == 0 || wh[0] > 0x7fffu || wh[1] > 0x7fffu || !fits((uint64_t)wh[0] * wh[1] * 3 * (storage == 1 ? 1 : 4))) return false;
            ppg_texture t{};
            t.width = wh[0]; t.height = wh[1]; t.uv_scale[0] = sc[0]; t.uv_scale[1] = sc[1]; t.uv_offset[0] = sc[2]; t.uv_offset[1] = sc[3];
            t.wrap_u = wr[0]; t.wrap_v = wr[1]; t.nearest = wr[2];
            std::vector<float> px((size_t)wh[0] * wh[1] * 3);
            if (storage == 1) {
                std::vector<unsigned char> raw(px.size());
                f.read((char *)raw.data(), raw.size());
                for (size_t i = 0; i < px.size(); ++i) px[i] = srgb[raw[i]];
            } else f.read((char *)px.data(), px.size() * 4);
            s.textures.push_back(t); s.texturePixels.push_back(std::move(px));
        }
    }
    return (bool)f;
}

struct M4 { double m[16]; };  // row major

static void cboxScene(int w, int h, SceneData &s) {
    // geometry / colours: see ppg_host/scenes.py:cbox_scene (scenes/cbox/cbox.xml + meshes/*.obj of the reference)
    struct Q { float v[4][3]; int mat, em; };
    std::vector<Q> quads;
    auto quad = [&](std::initializer_list<std::initializer_list<float>> vs, int mat, int em) {
        Q q; int i = 0; for (auto &v : vs) { int j = 0; for (float c : v) q.v[i][j++] = c; ++i; } q.mat = mat; q.em = em; quads.push_back(q);
    };
    const float ly = 1020.0f - 548.79999f, z0 = 550.0f - 227.0f, z1 = 550.0f - 332.0f;
    quad({{343, ly, z0}, {343, ly, z1}, {213, ly, z1}, {213, ly, z0}}, 4, 0);
    quad({{552.79999f, 0, 0}, {0, 0, 0}, {0, 0, 559.20001f}, {549.59998f, 0, 559.20001f}}, 1, -1);
    quad({{556, 548.79999f, 0}, {556, 548.79999f, 559.20001f}, {0, 548.79999f, 559.20001f}, {0, 548.79999f, 0}}, 1, -1);
    quad({{549.59998f, 0, 559.20001f}, {0, 0, 559.20001f}, {0, 548.79999f, 559.20001f}, {556, 548.79999f, 559.20001f}}, 1, -1);
    quad({{0, 0, 559.20001f}, {0, 0, 0}, {0, 548.79999f, 0}, {0, 548.79999f, 559.20001f}}, 3, -1);
    quad({{552.79999f, 0, 0}, {549.59998f, 0, 559.20001f}, {556, 548.79999f, 559.20001f}, {556, 548.79999f, 0}}, 2, -1);
    auto box = [&](const float fp[4][2], float hh) {
        const float *a = fp[0], *b = fp[1], *c = fp[2], *d = fp[3];
        quad({{a[0], hh, a[1]}, {b[0], hh, b[1]}, {c[0], hh, c[1]}, {d[0], hh, d[1]}}, 0, -1);
        const float *ring[4] = {a, d, c, b};
        for (int i = 0; i < 4; ++i) { const float *p = ring[i], *q = ring[(i + 1) % 4]; quad({{p[0], 0, p[1]}, {p[0], hh, p[1]}, {q[0], hh, q[1]}, {q[0], 0, q[1]}}, 0, -1); }
        quad({{d[0], 0, d[1]}, {c[0], 0, c[1]}, {b[0], 0, b[1]}, {a[0], 0, a[1]}}, 0, -1);
    };
    const float sm[4][2] = {{130, 65}, {82, 225}, {240, 272}, {290, 114}}, lg[4][2] = {{423, 247}, {265, 296}, {314, 456}, {472, 406}};
    box(sm, 165); box(lg, 330);
    for (auto &q : quads) {
        uint32_t base = (uint32_t)(s.positions.size() / 3);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) s.positions.push_back(q.v[i][j]);
        const uint32_t idx[6] = {base, base + 1, base + 2, base, base + 2, base + 3};
        s.indices.insert(s.indices.end(), idx, idx + 6);
        for (int t = 0; t < 2; ++t) { s.triMaterial.push_back((uint32_t)q.mat); s.triEmitter.push_back(q.em); }
    }
    const float rgb[5][3] = {{0.88579154f, 0.698900044f, 0.666440606f}, {0.88579154f, 0.698900044f, 0.666440606f}, {0.570083559f, 0.0430142134f, 0.0443643667f},
                             {0.10540954f, 0.377989352f, 0.076434195f}, {0.936401188f, 0.740475953f, 0.705280542f}};
    for (auto &c : rgb) { ppg_material m{}; m.type = PPG_BSDF_DIFFUSE; memcpy(m.reflectance, c, 12); s.materials.push_back(m); }
    ppg_emitter e{}; e.radiance[0] = 36.7738686f; e.radiance[1] = 21.976778f; e.radiance[2] = 5.50738001f; s.emitters.push_back(e);
    // perspective camera: sensor.cpp:239-264, perspective.cpp:150-164, transform.cpp:99-123, 191-214
    const double aspect = (double)w / h, fov = 39.3077, nearC = 10, farC = 2800;
    const double xfov = aspect > 1 ? 2 * std::atan(std::tan(0.5 * fov * M_PI / 180) * aspect) * 180 / M_PI : fov;  // fovAxis "smaller"
    const double cot = 1 / std::tan(xfov / 2 * M_PI / 180), recip = 1 / (farC - nearC);
    // cameraToSample = scale(-0.5, -0.5*aspect, 1) * translate(-1, -1/aspect, 0) * perspective  → invert analytically
    // x_s = -0.5 (cot x / z - 1), y_s = -0.5 aspect (cot y / z - 1 / aspect), z_s = far recip (1 - near / z)
    M4 inv{};  // sample → camera, homogeneous: (x, y, z, w) = (-(2 xs - 1)/cot, -(2 ys / aspect - 1/aspect)/cot, 1, (far - zs (far - near)) / (near far))
    inv.m[0] = -2 / cot; inv.m[3] = 1 / cot;
    inv.m[5] = -2 / (aspect * cot); inv.m[7] = 1 / (aspect * cot);
    inv.m[11] = 1;
    inv.m[14] = -(farC - nearC) / (nearC * farC); inv.m[15] = 1 / nearC;
    (void)recip;
    for (int i = 0; i < 16; ++i) s.camera.sample_to_camera[i] = (float)inv.m[i];
    // lookAt(origin (278, 273, -800), target = origin + z, up = y): dir = z, left = up x dir = x, newUp = y
    const double c2w[16] = {1, 0, 0, 278, 0, 1, 0, 273, 0, 0, 1, -800, 0, 0, 0, 1};
    for (int i = 0; i < 16; ++i) s.camera.camera_to_world[i] = (float)c2w[i];
    s.camera.near_clip = (float)nearC; s.camera.far_clip = (float)farC; s.camera.width = w; s.camera.height = h;
}

static void writePFM(const char *path, const std::vector<float> &rgb, int w, int h) {
    std::ofstream f(path, std::ios::binary);
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (int y = h - 1; y >= 0; --y) f.write((const char *)&rgb[(size_t)y * w * 3], (size_t)w * 12);  // PFM is bottom-up
}

// the flat scene file of ppg_host.save_scene (for tests and for handing a loaded XML scene to other tools)
static bool saveScene(const char *path, const SceneData &s) {
    std::ofstream f(path, std::ios::binary);
    const uint32_t hdr[6] = {(uint32_t)(s.positions.size() / 3), (uint32_t)(s.indices.size() / 3), (uint32_t)s.materials.size(), (uint32_t)s.emitters.size(),
                             s.normals.empty() ? 0u : 1u, (s.hasEnvironment ? 1u : 0u) | (s.rtrans.empty() ? 0u : 2u) | (s.spheres.empty() ? 0u : 4u) | (s.hasEnvmap ? 8u : 0u) |
                                 (s.texcoords.empty() ? 0u : 16u) | (s.textures.empty() ? 0u : 32u)};
    f.write("PPGS", 4); f.write((const char *)hdr, sizeof hdr);
    f.write((const char *)s.positions.data(), s.positions.size() * 4);
    if (!s.normals.empty()) f.write((const char *)s.normals.data(), s.normals.size() * 4);
    f.write((const char *)s.indices.data(), s.indices.size() * 4);
    f.write((const char *)s.triMaterial.data(), s.triMaterial.size() * 4);
    f.write((const char *)s.triEmitter.data(), s.triEmitter.size() * 4);
    f.write((const char *)s.materials.data(), s.materials.size() * sizeof(ppg_material));
    f.write((const char *)s.emitters.data(), s.emitters.size() * sizeof(ppg_emitter));
    f.write((const char *)&s.camera, sizeof(ppg_camera));
    if (s.hasEnvironment) f.write((const char *)s.environment, 12);
    if (!s.rtrans.empty()) {
        const uint32_t rt[2] = {(uint32_t)(s.rtrans.size() / (s.rtransSamples + 1)), s.rtransSamples};
        f.write((const char *)rt, 8); f.write((const char *)s.rtrans.data(), s.rtrans.size() * 4);
    }
    if (!s.spheres.empty()) {
        const uint32_t n = (uint32_t)s.spheres.size();
        f.write((const char *)&n, 4); f.write((const char *)s.spheres.data(), (size_t)n * sizeof(ppg_sphere));
    }
    if (s.hasEnvmap) {
        const uint32_t wh[2] = {s.envmap.width, s.envmap.height};
        f.write((const char *)wh, 8); f.write((const char *)&s.envmap.scale, 4); f.write((const char *)s.envmap.to_world, 36);
        f.write((const char *)s.envmapRgb.data(), s.envmapRgb.size() * 4);
    }
    if (!s.texcoords.empty()) f.write((const char *)s.texcoords.data(), s.texcoords.size() * 4);
    if (!s.textures.empty()) {
        const uint32_t n = (uint32_t)s.textures.size();
        f.write((const char *)&n, 4);
        for (uint32_t k = 0; k < n; ++k) {
            const ppg_texture &t = s.textures[k];
            const uint32_t wh[2] = {t.width, t.height}, storage = 0;
            const float sc[4] = {t.uv_scale[0], t.uv_scale[1], t.uv_offset[0], t.uv_offset[1]};
            const int32_t wr[3] = {t.wrap_u, t.wrap_v, t.nearest};
            f.write((const char *)wh, 8); f.write((const char *)sc, 16); f.write((const char *)wr, 12); f.write((const char *)&storage, 4);
            f.write((const char *)s.texturePixels[k].data(), s.texturePixels[k].size() * 4);
        }
    }
    return (bool)f;
}

int main(int argc, char **argv) {
    Properties props;
    std::string out = "out.pfm", scenePath, dumpScene, bsdfId, bsdfPlugin, ncclIdFile, runTag;
    std::vector<xml::BsdfParam> bsdfParams;
    int rank = 0, world = 1;
    bool quiet = false, lenient = false, pluginCore = false;
    int cancelAfterMs = -1;
    std::string dataDir;  // `data` directory of a Mitsuba tree (roughplastic: data/microfacet/*.dat); default $PPG_MITSUBA_DATA
    int cw = 0, ch = 0, sw = 0, sh = 0;
    std::map<std::string, std::string> defines;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-D" && i + 1 < argc) {
            std::string kv = argv[++i];
            size_t eq = kv.find('=');
            if (eq == std::string::npos) { std::cerr << "-D expects key=value\n"; return 2; }
            props.values[kv.substr(0, eq)] = kv.substr(eq + 1);
            defines[kv.substr(0, eq)] = kv.substr(eq + 1);  // also a $name for a scene XML, like mitsuba -D (mitsuba.cpp:58-87)
        } else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "--ppgs" && i + 1 < argc) dumpScene = argv[++i];
        else if (a == "--bsdf-plugin" && i + 1 < argc) bsdfPlugin = argv[++i];  // ... of a flat plug-in given by --bsdf-param name:tag:value (what the shim makes of a BSDF without an id)
        else if (a == "--bsdf-param" && i + 1 < argc) {
            const std::string v = argv[++i];
            const size_t c1 = v.find(':'), c2 = v.find(':', c1 + 1);
            if (c1 == std::string::npos || c2 == std::string::npos) { std::cerr << "--bsdf-param name:tag:value" << std::endl; return 2; }
            bsdfParams.push_back({v.substr(0, c1), v.substr(c1 + 1, c2 - c1 - 1), v.substr(c2 + 1)});
        }
        else if (a == "--bsdf-id" && i + 1 < argc) bsdfId = argv[++i];  // print the ppg_material of <bsdf id=...> (what the Mitsuba plug-in shim asks for) and exit
        else if (a == "--rank" && i + 1 < argc) rank = atoi(argv[++i]);       // multi-GPU: one process per GPU, image tiles sharded over `--world` ranks,
        else if (a == "--world" && i + 1 < argc) world = atoi(argv[++i]);     // RCCL communicator bootstrapped through the file `--nccl-id` (rank 0 writes it);
        else if (a == "--nccl-id" && i + 1 < argc) ncclIdFile = argv[++i];    // host/rccl_reducer.h
        else if (a == "--run-tag" && i + 1 < argc) runTag = argv[++i];        // the same string on all ranks of one run: a stale id file of an earlier run is ignored
        else if (a == "--lenient") lenient = true;
        else if (a == "--plugin-core") pluginCore = true;   // render the way the Mitsuba plug-in does: ppg::PluginCore (host/plugin_core.h), ONE ppg_render() call,
                                                            // SD-tree dumps (-D dumpSDTree=true) to "<out without extension>-NN.sdt" like scene->getDestinationFile()
        else if (a == "--cancel-after-ms" && i + 1 < argc) cancelAfterMs = atoi(argv[++i]);  // with --plugin-core: Integrator::cancel() from another thread (0 = before render())
        else if (a == "--data-dir" && i + 1 < argc) dataDir = argv[++i];
        else if (a == "--size" && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &sw, &sh) != 2) { std::cerr << "--size WxH\n"; return 2; } }
        else if (a == "-q") quiet = true;
        else if (a == "--cbox" && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &cw, &ch) != 2) { std::cerr << "--cbox WxH\n"; return 2; } }
        else if (a == "-h" || a == "--help") { std::cout << "usage: ppg_render [-D key=value]... [-o out.pfm] [-q] [--size WxH] [--lenient] [--data-dir mitsuba/data] [--ppgs flat-scene-out] (scene.xml | scene.ppgs | --cbox WxH)\n"; return 0; }
        else scenePath = a;
    }
    // scene.ppgs.props (written next to the flat scene by `python -m ppg_host scene.xml --ppgs scene.ppgs`): the XML's
    // integrator properties, one key=value per line; -D on the command line wins
    if (!scenePath.empty()) {
        std::ifstream pf(scenePath + ".props");
        std::string line;
        while (std::getline(pf, line)) {
            size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            std::string k = line.substr(0, eq), v = line.substr(eq + 1);
            if (v == "True") v = "true";
            if (v == "False") v = "false";
            if (!props.values.count(k)) props.values[k] = v;
        }
    }
    SceneData scene;
    const bool isXml = scenePath.size() > 4 && scenePath.compare(scenePath.size() - 4, 4, ".xml") == 0;
    if (!bsdfId.empty() || !bsdfPlugin.empty()) {
        ppg_material m{};
        std::string why;
        if (!bsdfPlugin.empty()) {
            if (!xml::bsdfFromProperties(bsdfPlugin, bsdfParams, dataDir, scene, m, why)) { std::cerr << "bsdf plug-in '" << bsdfPlugin << "': " << why << std::endl; return 2; }
        } else if (!isXml || !xml::bsdfById(scenePath, bsdfId, dataDir, scene, m, why)) { std::cerr << "no bsdf '" << bsdfId << "': " << why << std::endl; return 2; }
        std::cout << "{\"type\": " << m.type << ", \"flags\": " << m.flags << ", \"reflectance\": [" << m.reflectance[0] << ", " << m.reflectance[1] << ", " << m.reflectance[2]
                  << "], \"alpha\": " << m.alpha << ", \"eta\": " << m.eta[0] << ", \"rtrans\": " << m.rtrans << ", \"rtrans_slices\": "
                  << (scene.rtransSamples ? scene.rtrans.size() / (scene.rtransSamples + 1) : 0) << "}" << std::endl;
        return 0;
    }
    if (cw > 0) cboxScene(cw, ch, scene);
    else if (isXml) {
        try {
            LoadedScene ls = SceneXmlLoader(scenePath, defines, !lenient, sw, sh, dataDir).load();
            for (auto &w : ls.warnings) if (!quiet) std::cerr << "warning: " << w << std::endl;
            scene = std::move(ls.scene);
            for (auto &kv : ls.integrator.values) if (!props.values.count(kv.first)) props.values[kv.first] = kv.second;  // -D on the command line wins
        } catch (const std::exception &e) {
            std::cerr << "cannot load scene '" << scenePath << "': " << e.what() << std::endl;
            return 2;
        }
    } else if (scenePath.empty() || !loadScene(scenePath.c_str(), scene)) { std::cerr << "cannot load scene '" << scenePath << "'\n"; return 2; }
    if (!dumpScene.empty()) {  // convert only (no GPU needed)
        if (!saveScene(dumpScene.c_str(), scene)) { std::cerr << "cannot write '" << dumpScene << "'\n"; return 2; }
        std::ofstream pf(dumpScene + ".props");
        for (auto &kv : props.values) pf << kv.first << "=" << kv.second << "\n";
        return 0;
    }
    if (pluginCore) {
        PluginCore core;
        core.configure(props);
        core.setSeed((uint64_t)std::stoull(props.getString("seed", "0")));
        std::string dest = out;
        const size_t dot = dest.find_last_of('.');
        if (dot != std::string::npos && dest.find('/', dot) == std::string::npos) dest.erase(dot);
        std::thread canceller;
        if (cancelAfterMs == 0) core.cancel();
        else if (cancelAfterMs > 0) canceller = std::thread([&] { std::this_thread::sleep_for(std::chrono::milliseconds(cancelAfterMs)); core.cancel(); });
        std::string err;
        const ppg_scene sv = scene.view();
        const int rc = core.render(sv, dest, err);
        if (canceller.joinable()) canceller.join();
        if (rc != PPG_OK && rc != PPG_ERR_CANCELLED) { std::cerr << "error: " << err << std::endl; return 3; }
        if (rc == PPG_ERR_CANCELLED && !quiet) std::cout << "(cancelled)" << std::endl;
        if (core.hasFilm()) {  // (as the plug-in does: no picture after a cancel() that came before the context had a scene)
            std::vector<float> rgb((size_t)scene.camera.width * scene.camera.height * 3);
            if (core.readFilm(rgb.data(), err) != PPG_OK) { std::cerr << "error: " << err << std::endl; return 3; }
            writePFM(out.c_str(), rgb, scene.camera.width, scene.camera.height);
        }
        return rc == PPG_OK ? 0 : 1;
    }
    try {
        std::unique_ptr<RcclReducer> reducer;
        if (!ncclIdFile.empty()) {
            if (world < 1 || rank < 0 || rank >= world) { std::cerr << "--rank / --world out of range\n"; return 2; }
            if (!props.values.count("device")) props.values["device"] = std::to_string(rank);  // one GPU per rank of the node
            const auto t0 = std::chrono::steady_clock::now();
            // the run tag keeps a rank from accepting the id file a crashed earlier run left at the same path: any string all ranks of THIS run share
            for (const char *v : {"PPG_RUN_TAG", "SLURM_JOB_ID", "TORCHELASTIC_RUN_ID"}) if (runTag.empty() && getenv(v)) runTag = getenv(v);
            if (runTag.empty() && world > 1) { std::cerr << "--world > 1 needs --run-tag (or PPG_RUN_TAG / SLURM_JOB_ID): a string shared by the ranks of this run\n"; return 2; }
            reducer.reset(new RcclReducer(rank, world, std::stoi(props.values["device"]), ncclIdFile, runTag));
            if (!quiet) std::cout << "RCCL communicator: rank " << rank << " of " << world << " ready after "
                                  << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s" << std::endl;
        }
        GuidedPathTracerHIP gpt(props);
        const auto t0 = std::chrono::steady_clock::now();
        const bool talk = !quiet && rank == 0;
        bool ok = gpt.render(scene, talk ? [](const std::string &s) { std::cout << s << std::endl; } : GuidedPathTracerHIP::LogFn(), reducer.get());
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (talk) std::cout << "Render time: " << sec << "s" << (ok ? "" : " (cancelled)") << std::endl;
        if (reducer && talk) std::cout << "RCCL: " << reducer->collectives() << " collectives, " << reducer->bytes() / 1e6 << " MB staged" << std::endl;
        if (rank == 0) writePFM(out.c_str(), gpt.film(), scene.camera.width, scene.camera.height);
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 3;
    }
}
