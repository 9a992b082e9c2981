/*
 * guided_path_hip.cpp — Mitsuba 0.5 integrator plug-in that routes GuidedPathTracer::render() through libppg_hip.so (MI355X).
 *
 * SOURCE ONLY.  It is written against the Mitsuba tree of the reference (mitsuba/include/mitsuba/...) and is meant to be dropped into
 * mitsuba/src/integrators/path/ next to guided_path.cpp and built like any other plug-in, linked with -lppg_hip:
 *
 *     plugins += env.SharedLibrary('guided_path_hip', ['path/guided_path_hip.cpp'], LIBS = env['LIBS'] + ['ppg_hip'],
 *                                  CPPPATH = env['CPPPATH'] + ['#/../practical-path-guiding_amd/host', '#/../include'])
 *
 * It cannot be compiled in this repository's environment (mitsuba.h:24 needs boost, SURVEY.md §7 "hard parts") and is therefore not
 * exercised by the tests; everything below the Mitsuba types — the C-ABI, the scene description it fills, the BSDF parser it reuses —
 * is (tests/test_abi.py, tests/test_cpp_host.py).  Selecting <integrator type="guided_path_hip"> (or installing the library as
 * plugins/guided_path.so) renders through the GPU with the reference's property names (GP = guided_path.cpp).
 *
 * What it does in render():
 *   geometry    walks scene->getShapes(): every TriMesh (obj, ply, serialized, rectangle, cube, ... all become TriMesh objects) is appended
 *               to one vertex / index array in WORLD space exactly as Mitsuba holds it — positions, vertex normals, texture coordinates —
 *               so OBJ semantics (vertex merging, maxSmoothAngle, flipTexCoords, computeNormals) are Mitsuba's own; `sphere` shapes become
 *               ppg_sphere records (centre, radius, rotation, flipNormals from the shape's Properties, sphere.cpp:108-131)
 *   BSDFs       Mitsuba exposes no accessor for the nested BSDF of twosided / mask / bumpmap or for a plug-in's parameters after
 *               configure(); the Properties every ConfigurableObject keeps (cobject.h:77) hold the scene file's values for flat BSDFs, and for
 *               adapters the <bsdf> element is re-read from the scene file (scene->getSourceFile(), matched by id, scene.h:1107) with the
 *               parser of this repository's stand-alone driver (host/scene_xml.h — the same code path tests/test_cpp_host.py pins against
 *               the Python loader).  Unsupported plug-ins end the render with an error that names them.
 *   emitters    `area` emitters by shape (radiance from the emitter's Properties), `constant`, `envmap` (level 0 of the bitmap it was given)
 *               and `sunsky` / `sky` / `sun` through the environment map Mitsuba itself rasterised (getEnvironmentEmitter() → its bitmap)
 *   sensor      PerspectiveCamera: m_sampleToCamera rebuilt from getXFov / clip planes / crop (perspective.cpp:150-164), world transform
 *   result      film->setBitmap(weight-normalised RGB), false when cancelled (GP:1584)
 */
#include <mitsuba/render/scene.h>
#include <mitsuba/render/trimesh.h>
#include <mitsuba/render/sensor.h>
#include <mitsuba/render/film.h>
#include <mitsuba/render/emitter.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/fresolver.h>   /* Thread::getFileResolver()->resolve(): the running Mitsuba's data directory */
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/statistics.h>

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ppg.h"
#include "guided_path_hip.h"   /* ppg::SceneData (host side of the C-ABI) */
#include "plugin_core.h"       /* ppg::PluginCore: property mapping + create / scene / render / cancel, shared with the stand-alone driver */
#include "scene_xml.h"         /* ppg::xml: the <bsdf> parser of the stand-alone driver */

MTS_NAMESPACE_BEGIN

class GuidedPathTracerHIP : public Integrator {
public:
    /* same names and defaults as GuidedPathTracer(const Properties &), GP:1015-1084: ppg::PluginCore::configure (host/plugin_core.h), which
       the stand-alone driver runs too.  The context is created in render(), where the destination of the SD-tree dumps is known. */
    GuidedPathTracerHIP(const Properties &props) : Integrator(props) { m_core.configure(props); }

    virtual ~GuidedPathTracerHIP() { }

    bool preprocess(const Scene *, RenderQueue *, const RenderJob *, int, int, int) { return true; }

    /* Integrator::render, integrator.h:74-75 (GP:1516-1585) */
    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int sceneResID, int sensorResID, int samplerResID) {
        ref<Scheduler> sched = Scheduler::getInstance();
        ref<Sensor> sensor = static_cast<Sensor *>(sched->getResource(sensorResID));
        ref<Film> film = sensor->getFilm();

        ppg::SceneData data;
        std::string why;
        if (!flatten(scene, sensor.get(), film.get(), data, why)) {
            Log(EError, "guided_path_hip: %s", why.c_str());
            return false;                 /* (EError normally throws; if it is configured not to, stop here) */
        }
        const ppg_scene desc = data.view();
        Log(EInfo, "Starting render job (%ix%i, MI355X, %i triangles, %i analytic spheres) ..", film->getCropSize().x, film->getCropSize().y,
            (int) (data.indices.size() / 3), (int) data.spheres.size());
        /* create → scene → render; SD-tree dumps to "<dest>-NN.sdt" (GP:1192-1195) */
        const int rc = m_core.render(desc, scene->getDestinationFile().string(), why);
        if (rc != PPG_OK && rc != PPG_ERR_CANCELLED) {
            Log(EError, "guided_path_hip: %s", why.c_str());   /* unknown enum strings: where GP:1023.. Assert(false) */
            return false;
        }
        if (!m_core.hasFilm()) return false;   /* cancel() before the context had a scene: nothing to develop, GP:1584 */
        ref<Bitmap> out = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, film->getCropSize());
        if (m_core.readFilm(out->getFloat32Data(), why) != PPG_OK) {
            Log(EError, "guided_path_hip: %s", why.c_str());
            return false;
        }
        film->setBitmap(out);
        return rc == PPG_OK;              /* false when cancelled, GP:1584 */
    }

    void cancel() { m_core.cancel(); }   /* Integrator::cancel, integrator.h:84 / GP:1643-1648; thread-safe, also before render() */

    std::string toString() const { return "GuidedPathTracerHIP[libppg_hip.so]"; }

    MTS_DECLARE_CLASS()

private:
    /* ------------------------------------------------------------------------------------------------------------------------------ */
    static void copy4x4(const Matrix4x4 &m, float *dst) {
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) dst[4 * r + c] = (float) m(r, c);
    }

    /* BSDF of a shape → index into data.materials.  Flat plug-ins from their own Properties; adapters from the scene file (see header). */
    int materialOf(const Shape *shape, const Scene *scene, ppg::SceneData &data, std::map<const BSDF *, int> &cache, std::string &why) {
        const BSDF *bsdf = shape->getBSDF();
        if (!bsdf) {                      /* Shape::configure (shape.cpp:48-72) always installs one; be defensive */
            ppg_material m; memset(&m, 0, sizeof m); m.type = PPG_BSDF_DIFFUSE;
            data.materials.push_back(m); return (int) data.materials.size() - 1;
        }
        std::map<const BSDF *, int>::const_iterator it = cache.find(bsdf);
        if (it != cache.end()) return it->second;
        ppg_material m; memset(&m, 0, sizeof m);
        const Properties &p = bsdf->getProperties();
        const std::string plugin = p.getPluginName();
        bool ok = false;
        if (plugin == "twosided" || plugin == "mask" || plugin == "bumpmap" || p.getID() != "unnamed") {
            /* the nested BSDF is not reachable through Mitsuba's API: read the element with this id from the scene file.  ppg::xml::SceneLoader
               is the loader behind `ppg_render scene.xml`; bsdfById() returns the ppg_material it makes of that <bsdf> element (incl. the
               rough-transmittance slice of a roughplastic, appended to data.rtrans, and its textures, appended to data.textures) */
            ok = ppg::xml::bsdfById(scene->getSourceFile().string(), p.getID(), m_dataDir(), data, m, why);
        }
        if (!ok) {
            /* A flat plug-in without an id: its Properties are the parameters the scene file gave it.  They are handed to the scene loader's own
               <bsdf> handling as the element they came from (ppg::xml::bsdfFromProperties) — one code path for every plug-in the HIP path
               knows (diffuse, conductor, roughconductor, plastic, roughplastic, dielectric, thindielectric, roughdielectric), named materials,
               IOR names and rough-transmittance slices included.  Spectra arrive as linear RGB (this is an RGB build of Mitsuba). */
            std::vector<ppg::xml::BsdfParam> params;
            std::vector<std::string> names;
            p.putPropertyNames(names);
            for (size_t k = 0; k < names.size(); ++k) {
                const std::string &n = names[k];
                ppg::xml::BsdfParam q;
                q.name = n;
                switch (p.getType(n)) {
                    case Properties::EBoolean: q.tag = "boolean"; q.value = p.getBoolean(n) ? "true" : "false"; break;
                    case Properties::EInteger: q.tag = "integer"; q.value = formatString("%i", p.getInteger(n)); break;
                    case Properties::EFloat: q.tag = "float"; q.value = formatString("%.9g", (double) p.getFloat(n)); break;
                    case Properties::EString: q.tag = "string"; q.value = p.getString(n); break;
                    case Properties::ESpectrum: {
                        Float r, g, b;
                        p.getSpectrum(n).toLinearRGB(r, g, b);
                        q.tag = "rgb"; q.value = formatString("%.9g, %.9g, %.9g", (double) r, (double) g, (double) b);
                        break;
                    }
                    default: continue;   /* transforms, points: no BSDF parameter of the supported plug-ins */
                }
                params.push_back(q);
            }
            std::string why2;
            ok = ppg::xml::bsdfFromProperties(plugin, params, m_dataDir(), data, m, why2);
            if (!ok && why.empty()) why = why2;
        }
        if (!ok) {
            if (why.empty()) why = "BSDF plug-in '" + plugin + "' (id '" + p.getID() + "') is not supported by the HIP path";
            return -1;
        }
        data.materials.push_back(m);
        cache[bsdf] = (int) data.materials.size() - 1;
        return cache[bsdf];
    }

    static std::string m_dataDir() {    /* data/microfacet/*.dat, data/ior/*.spd of the running Mitsuba */
        return Thread::getThread()->getFileResolver()->resolve("data/microfacet/beckmann.dat").parent_path().parent_path().string();
    }

    /* Scene → ppg::SceneData */
    bool flatten(const Scene *scene, const Sensor *sensor, const Film *film, ppg::SceneData &data, std::string &why) {
        std::map<const BSDF *, int> matCache;
        bool anyNormals = false, anyUVs = false;
        const ref_vector<Shape> &shapes = scene->getShapes();
        for (size_t i = 0; i < shapes.size(); ++i)
            if (shapes[i]->getClass()->derivesFrom(MTS_CLASS(TriMesh))) {
                const TriMesh *mesh = static_cast<const TriMesh *>(shapes[i].get());
                anyNormals |= mesh->getVertexNormals() != NULL;
                anyUVs |= mesh->getVertexTexcoords() != NULL;
            }
        const float nan = std::numeric_limits<float>::quiet_NaN();
        for (size_t i = 0; i < shapes.size(); ++i) {
            const Shape *shape = shapes[i].get();
            const int mat = materialOf(shape, scene, data, matCache, why);
            if (mat < 0) return false;
            int em = -1;
            if (shape->isEmitter()) {   /* AreaLight (area.cpp:62-75): constant radiance */
                const Properties &ep = shape->getEmitter()->getProperties();
                if (ep.getPluginName() != "area") { why = "emitter plug-in '" + ep.getPluginName() + "' on a shape is not supported"; return false; }
                Float r, g, b;
                ep.getSpectrum("radiance", Spectrum(1.0f)).toLinearRGB(r, g, b);
                ppg_emitter e; e.radiance[0] = r; e.radiance[1] = g; e.radiance[2] = b; e._pad = 0;
                data.emitters.push_back(e);
                em = (int) data.emitters.size() - 1;
            }
            if (shape->getClass()->derivesFrom(MTS_CLASS(TriMesh))) {
                const TriMesh *mesh = static_cast<const TriMesh *>(shape);
                const uint32_t base = (uint32_t) (data.positions.size() / 3);
                const Point *P = mesh->getVertexPositions();
                const Normal *N = mesh->getVertexNormals();
                const Point2 *UV = mesh->getVertexTexcoords();
                const Triangle *T = mesh->getTriangles();
                if (anyNormals && !N) {
                    /* one normal array for the whole scene: a mesh without vertex normals gets its vertices un-shared and its face normals
                       written out (the shading frame fillIntersectionRecord derives without normals, skdtree.h:388-401) */
                    for (size_t t = 0; t < mesh->getTriangleCount(); ++t) {
                        const Point &a = P[T[t].idx[0]], &b = P[T[t].idx[1]], &c = P[T[t].idx[2]];
                        Normal fn(cross(b - a, c - a));
                        const Float len = fn.length();
                        if (len != 0) fn /= len;
                        for (int v = 0; v < 3; ++v) {
                            const Point &q = P[T[t].idx[v]];
                            data.positions.push_back(q.x); data.positions.push_back(q.y); data.positions.push_back(q.z);
                            data.normals.push_back(fn.x); data.normals.push_back(fn.y); data.normals.push_back(fn.z);
                            if (anyUVs) { data.texcoords.push_back(UV ? UV[T[t].idx[v]].x : nan); data.texcoords.push_back(UV ? UV[T[t].idx[v]].y : nan); }
                            data.indices.push_back(base + (uint32_t) (3 * t + v));
                        }
                        data.triMaterial.push_back((uint32_t) mat); data.triEmitter.push_back(em);
                    }
                    continue;
                }
                for (size_t v = 0; v < mesh->getVertexCount(); ++v) {
                    data.positions.push_back(P[v].x); data.positions.push_back(P[v].y); data.positions.push_back(P[v].z);
                    if (anyNormals) { data.normals.push_back(N[v].x); data.normals.push_back(N[v].y); data.normals.push_back(N[v].z); }
                    if (anyUVs) { data.texcoords.push_back(UV ? UV[v].x : nan); data.texcoords.push_back(UV ? UV[v].y : nan); }
                }
                for (size_t t = 0; t < mesh->getTriangleCount(); ++t) {
                    for (int v = 0; v < 3; ++v) data.indices.push_back(base + T[t].idx[v]);
                    data.triMaterial.push_back((uint32_t) mat); data.triEmitter.push_back(em);
                }
            } else if (shape->getProperties().getPluginName() == "sphere") {
                /* Sphere::Sphere (sphere.cpp:108-131): the scale of toWorld goes into the radius, the rest stays a rotation */
                const Properties &sp = shape->getProperties();
                Transform o2w = sp.getTransform("toWorld", Transform()) * Transform::translate(Vector(sp.getPoint("center", Point(0.0f))));
                Float radius = sp.getFloat("radius", 1.0f);
                if (sp.hasProperty("toWorld")) {
                    const Float scale = o2w(Vector(1, 0, 0)).length();
                    o2w = o2w * Transform::scale(Vector(1 / scale));
                    radius *= scale;
                }
                ppg_sphere s; memset(&s, 0, sizeof s);
                const Point c = o2w(Point(0.0f));
                s.center[0] = c.x; s.center[1] = c.y; s.center[2] = c.z; s.radius = radius;
                for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) s.to_world[3 * r + k] = o2w.getMatrix()(r, k);
                s.material = (uint32_t) mat; s.emitter = em; s.flip_normals = sp.getBoolean("flipNormals", false) ? 1 : 0;
                data.spheres.push_back(s);
            } else {
                why = "shape plug-in '" + shape->getProperties().getPluginName() + "' is not supported (triangle meshes and spheres are)";
                return false;
            }
        }

        /* environment emitter: constant, or whatever map Mitsuba holds (envmap, and sunsky / sky / sun after their own rasterisation) */
        if (const Emitter *env = scene->getEnvironmentEmitter()) {
            const Properties &ep = env->getProperties();
            if (ep.getPluginName() == "constant") {
                Float r, g, b;
                ep.getSpectrum("radiance", Spectrum(1.0f)).toLinearRGB(r, g, b);
                data.hasEnvironment = true; data.environment[0] = r; data.environment[1] = g; data.environment[2] = b;
            } else {
                ref<Bitmap> map = const_cast<Emitter *>(env)->getBitmap(Vector2i(-1, -1));   /* EnvironmentMap::getBitmap: the full-resolution level */
                if (!map) { why = "environment emitter '" + ep.getPluginName() + "' exposes no bitmap"; return false; }
                ref<Bitmap> rgb = map->convert(Bitmap::ERGB, Bitmap::EFloat32);
                data.hasEnvmap = true;
                data.envmap.width = (uint32_t) rgb->getWidth(); data.envmap.height = (uint32_t) rgb->getHeight();
                data.envmap.scale = 1.0f;      /* getBitmap() returns the map with m_scale applied */
                data.envmapRgb.assign(rgb->getFloat32Data(), rgb->getFloat32Data() + (size_t) rgb->getWidth() * rgb->getHeight() * 3);
                const Matrix4x4 &w = env->getWorldTransform()->eval(0).getMatrix();
                for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) data.envmap.to_world[3 * r + k] = w(r, k);
            }
        }

        /* PerspectiveCamera (perspective.cpp:150-164): m_sampleToCamera for the film's crop window; world transform at time 0 */
        if (sensor->getProperties().getPluginName() != "perspective") { why = "sensor plug-in '" + sensor->getProperties().getPluginName() + "' is not supported"; return false; }
        const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
        const Vector2i size = film->getSize(), crop = film->getCropSize();
        const Point2i off = film->getCropOffset();
        const Float aspect = (Float) size.x / (Float) size.y;
        const Vector2 relSize((Float) crop.x / size.x, (Float) crop.y / size.y);
        const Point2 relOffset((Float) off.x / size.x, (Float) off.y / size.y);
        const Transform cameraToSample = Transform::scale(Vector(1.0f / relSize.x, 1.0f / relSize.y, 1.0f))
            * Transform::translate(Vector(-relOffset.x, -relOffset.y, 0.0f))
            * Transform::scale(Vector(-0.5f, -0.5f * aspect, 1.0f))
            * Transform::translate(Vector(-1.0f, -1.0f / aspect, 0.0f))
            * Transform::perspective(cam->getXFov(), cam->getNearClip(), cam->getFarClip());
        copy4x4(cameraToSample.inverse().getMatrix(), data.camera.sample_to_camera);
        copy4x4(cam->getWorldTransform(0).getMatrix(), data.camera.camera_to_world);
        data.camera.near_clip = cam->getNearClip(); data.camera.far_clip = cam->getFarClip();
        data.camera.width = crop.x; data.camera.height = crop.y;
        if (film->getReconstructionFilter()->getProperties().getPluginName() != "box")
            Log(EWarn, "guided_path_hip: the HIP path accumulates with the box filter of the reference's scenes (film uses '%s')",
                film->getReconstructionFilter()->getProperties().getPluginName().c_str());
        return true;
    }

    ppg::PluginCore m_core;
};

/* not serialisable, like the reference's integrator (guided_path.cpp:2421: MTS_IMPLEMENT_CLASS, no stream constructor) */
MTS_IMPLEMENT_CLASS(GuidedPathTracerHIP, false, Integrator)
MTS_EXPORT_PLUGIN(GuidedPathTracerHIP, "Guided path tracer (MI355X, libppg_hip.so)");
MTS_NAMESPACE_END
