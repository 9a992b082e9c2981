/*
 * sunsky.h — Mitsuba's `sunsky` emitter as what it is: a bake.  SunSkyEmitter (emitters/sunsky.cpp:100-235) rasterises the Hosek-Wilkie
 * sky (emitters/sky.cpp:386-447, sunsky/skymodel.cpp) into a resolution x resolution/2 latitude-longitude bitmap, splats Preetham's sun
 * disc (sunsky/sunmodel.h:255-365) on top with (0,2)-sequence samples (sunsky.cpp:170-205) and instantiates an `envmap` emitter on it.
 * The scene loader does the same (the C++ twin of ppg_host/sunsky.py, which documents every step), so the kernels see an ordinary
 * ppg_envmap and `ppg_render kitchen-improved.xml` renders the headline scene without the Python converter.
 *
 * The model's coefficient tables (datasetRGB1..3, datasetRGBRad1..3 of sunsky/skymodeldata.h; k_o / k_g / k_wa / solar tables of
 * sunsky/sunmodel.h) are PARSED from the operator's Mitsuba source tree at load time, never stored in this repository.
 */
#ifndef PPG_HOST_SUNSKY_H
#define PPG_HOST_SUNSKY_H

#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ppg {

// (included by scene_xml.h after its spectrum helpers: detail::evalInterp, interpolatedToRGB)

namespace sunsky {

// `double name[] = { ... };` / `Float name[N] = { ... };` initialisers of a C source file
inline std::map<std::string, std::vector<double>> parseCArrays(const std::string &path, const std::vector<std::string> &names) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("sunsky: cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str(), clean;
    clean.reserve(text.size());
    for (size_t i = 0; i < text.size();) {  // strip comments
        if (text.compare(i, 2, "/*") == 0) { const size_t e = text.find("*/", i + 2); i = e == std::string::npos ? text.size() : e + 2; }
        else if (text.compare(i, 2, "//") == 0) { const size_t e = text.find('\n', i); i = e == std::string::npos ? text.size() : e; }
        else clean.push_back(text[i++]);
    }
    std::map<std::string, std::vector<double>> out;
    for (const std::string &name : names) {
        size_t pos = 0;
        bool found = false;
        while ((pos = clean.find(name, pos)) != std::string::npos) {
            const bool wordStart = pos == 0 || !(isalnum((unsigned char)clean[pos - 1]) || clean[pos - 1] == '_');
            size_t q = pos + name.size();
            while (q < clean.size() && isspace((unsigned char)clean[q])) ++q;
            if (wordStart && q < clean.size() && clean[q] == '[') {
                const size_t eq = clean.find('=', q), open = clean.find('{', q), semi = clean.find(';', q);
                if (eq != std::string::npos && open != std::string::npos && eq < open && open < semi) {
                    const size_t close = clean.find('}', open);
                    std::string body = clean.substr(open + 1, close - open - 1);
                    for (char &c : body) if (c == ',' || c == '\n' || c == '\r' || c == '\t') c = ' ';
                    std::istringstream is(body);
                    std::vector<double> v;
                    std::string tok;
                    while (is >> tok) v.push_back(std::stod(tok));  // (stod stops at a trailing `f` suffix)
                    out[name] = v;
                    found = true;
                    break;
                }
            }
            pos += name.size();
        }
        if (!found) throw std::runtime_error("sunsky: array " + name + " not found in " + path);
    }
    return out;
}

struct Params {
    double latitude = 35.6894, longitude = 139.6917, timezone = 9, hour = 15, minute = 0, second = 0;
    int year = 2010, month = 7, day = 10, resolution = 512;
    double scale = 1, sunScale = -1, skyScale = -1, sunRadiusScale = 1, turbidity = 3, stretch = 1;
    double albedo[3] = {0.2, 0.2, 0.2};
};

// computeSunCoordinates, sunmodel.h:217-251 (PSA algorithm, :103-205): (zenith angle, azimuth), float-rounded like the reference's Float
inline void sunCoordinates(const Params &p, double &zenith, double &azimuth) {
    const double lat = (float)p.latitude, lon = (float)p.longitude, tz = (float)p.timezone;
    const double hour = (float)p.hour, minute = (float)p.minute, second = (float)p.second;
    const double decHours = hour - tz + (minute + second / 60.0) / 60.0;
    const long aux1 = (p.month - 14) / 12;  // C integer division truncates towards zero
    const long aux2 = (1461 * (p.year + 4800 + aux1)) / 4 + (367 * (p.month - 2 - 12 * aux1)) / 12 - (3 * ((p.year + 4900 + aux1) / 100)) / 4 + p.day - 32075;
    const double elapsed = (double)aux2 - 0.5 + decHours / 24.0 - 2451545.0;
    const double omega = 2.1429 - 0.0010394594 * elapsed;
    const double meanLongitude = 4.8950630 + 0.017202791698 * elapsed;
    const double anomaly = 6.2400600 + 0.0172019699 * elapsed;
    const double eclLong = meanLongitude + 0.03341607 * std::sin(anomaly) + 0.00034894 * std::sin(2 * anomaly) - 0.0001134 - 0.0000203 * std::sin(omega);
    const double eclObl = 0.4090928 - 6.2140e-9 * elapsed + 0.0000396 * std::cos(omega);
    const double sinEl = std::sin(eclLong);
    double ra = std::atan2(std::cos(eclObl) * sinEl, std::cos(eclLong));
    if (ra < 0) ra += 2 * M_PI;
    const double decl = std::asin(std::sin(eclObl) * sinEl);
    const double gmst = 6.6974243242 + 0.0657098283 * elapsed + decHours;
    const double lmst = (float)((double)(float)(gmst * 15 + lon) * (M_PI / 180.0));
    const double latR = (float)(lat * (M_PI / 180.0));
    const double hourAngle = lmst - ra;
    double elevation = std::acos(std::cos(latR) * std::cos(hourAngle) * std::cos(decl) + std::sin(decl) * std::sin(latR));
    double az = std::atan2(-std::sin(hourAngle), std::tan(decl) * std::cos(latR) - std::sin(latR) * std::cos(hourAngle));
    if (az < 0) az += 2 * M_PI;
    elevation += (6371.01 / 149597890) * std::sin(elevation);  // parallax
    zenith = (float)elevation; azimuth = (float)az;
}

inline double bezier5(double t, const double *m, int stride) {  // skymodel.cpp:106-113
    const double s = 1.0 - t;
    return std::pow(s, 5) * m[0] + 5 * std::pow(s, 4) * t * m[stride] + 10 * std::pow(s, 3) * t * t * m[2 * stride] + 10 * s * s * std::pow(t, 3) * m[3 * stride] +
           5 * s * std::pow(t, 4) * m[4 * stride] + std::pow(t, 5) * m[5 * stride];
}

// arhosek_rgb_skymodelstate_alloc_init for one channel (skymodel.cpp:346-374, 80-224)
inline void hosekState(const std::vector<double> &ds, const std::vector<double> &rad, double turbidity, double albedo, double solarElevation, double config[9], double &radiance) {
    const int it = (int)turbidity;
    if (it < 1 || it > 10) throw std::runtime_error("sunsky: turbidity must be in [1, 10]");
    const double rem = turbidity - it, t = std::pow(solarElevation / (M_PI / 2.0), 1.0 / 3.0);
    for (int i = 0; i < 9; ++i) config[i] = 0;
    radiance = 0;
    const double wAlb[2] = {1.0 - albedo, albedo};
    for (int alb = 0; alb < 2; ++alb) {
        const int turbs[2] = {it - 1, it};
        const double wT[2] = {1.0 - rem, rem};
        for (int k = 0; k < 2; ++k) {
            if (turbs[k] > 9) continue;  // int_turbidity == 10: the function returns before the high-turbidity terms
            const double *m = ds.data() + 9 * 6 * 10 * alb + 9 * 6 * turbs[k];
            for (int i = 0; i < 9; ++i) config[i] += wAlb[alb] * wT[k] * bezier5(t, m + i, 9);
            radiance += wAlb[alb] * wT[k] * bezier5(t, rad.data() + 6 * 10 * alb + 6 * turbs[k], 1);
        }
    }
}

inline double hosekRadiance(const double c[9], double radiance, double theta, double gamma) {  // skymodel.cpp:226-239, 383-397
    const double cg = std::cos(gamma), ct = std::cos(theta);
    const double expM = std::exp(c[4] * gamma), rayM = cg * cg;
    const double mieM = (1.0 + cg * cg) / std::pow(1.0 + c[8] * c[8] - 2.0 * c[8] * cg, 1.5);
    const double zenith = std::sqrt(ct);
    return (1.0 + c[0] * std::exp(c[1] / (ct + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith) * radiance;
}

// computeSunRadiance (sunmodel.h:317-365) through Spectrum::fromContinuousSpectrum of the RGB build
inline void sunRadianceRGB(std::map<std::string, std::vector<double>> &T, double theta, double turbidity, double rgb[3]) {
    std::vector<double> lam, data;
    std::vector<double> kO(T["k_oAmplitudes"].begin(), T["k_oAmplitudes"].begin() + 64);
    const double beta = 0.04608365822050 * turbidity - 0.04586025928522;
    const double m = 1.0 / (std::cos(theta) + 0.15 * std::pow(93.885 - theta / M_PI * 180.0, -1.253));
    for (double l = 350.0; l <= 800.0; l += 5.0) {
        const double ko = detail::evalInterp(T["k_oWavelengths"], kO, l), kg = detail::evalInterp(T["k_gWavelengths"], T["k_gAmplitudes"], l);
        const double kwa = detail::evalInterp(T["k_waWavelengths"], T["k_waAmplitudes"], l), sol = detail::evalInterp(T["solWavelengths"], T["solAmplitudes"], l);
        const double tauR = std::exp(-m * 0.008735 * std::pow(l / 1000.0, -4.08)), tauA = std::exp(-m * beta * std::pow(l / 1000.0, -1.3));
        const double tauO = std::exp(-m * ko * 0.35), tauG = std::exp(-1.41 * kg * m / std::pow(1 + 118.93 * kg * m, 0.45));
        const double tauWA = std::exp(-0.2385 * kwa * 2.0 * m / std::pow(1 + 20.07 * kwa * 2.0 * m, 0.45));
        lam.push_back(l); data.push_back(sol * tauR * tauA * tauO * tauG * tauWA);
    }
    float out[3];
    interpolatedToRGB(lam, data, out, false, true);
    for (int c = 0; c < 3; ++c) rgb[c] = out[c];
}

// (0,2)-sequence point i (qmc.h:43-59, 82-87, 115-120)
inline void sample02(uint32_t i, float &u, float &v) {
    uint32_t x = 0;
    for (int b = 0; b < 32; ++b) x |= ((i >> b) & 1u) << (31 - b);
    u = (float)(x >> 8) / (float)(1 << 24);
    uint32_t y = 0, vv = 1u << 31;
    for (uint32_t k = i; k; k >>= 1, vv ^= vv >> 1) if (k & 1u) y ^= vv;
    v = (float)((double)y / 4294967296.0);
}

// SunSkyEmitter(props) → float RGB [resolution / 2][resolution][3] in the envmap plug-in's latitude-longitude layout
inline void bake(const Params &p, const std::string &mitsubaSrc, std::vector<float> &rgb, int &width, int &height) {
    const std::string d = mitsubaSrc + "/src/emitters/sunsky/";
    auto T = parseCArrays(d + "skymodeldata.h", {"datasetRGB1", "datasetRGB2", "datasetRGB3", "datasetRGBRad1", "datasetRGBRad2", "datasetRGBRad3"});
    auto T2 = parseCArrays(d + "sunmodel.h", {"k_oWavelengths", "k_oAmplitudes", "k_gWavelengths", "k_gAmplitudes", "k_waWavelengths", "k_waAmplitudes", "solWavelengths", "solAmplitudes"});
    T.insert(T2.begin(), T2.end());
    const double sunScale = p.sunScale < 0 ? p.scale : p.sunScale, skyScale = p.skyScale < 0 ? p.scale : p.skyScale;
    const int w = p.resolution, h = p.resolution / 2;
    double sunZenith, sunAzimuth;
    sunCoordinates(p, sunZenith, sunAzimuth);
    const double sunElevation = 0.5 * M_PI - sunZenith;
    if (sunElevation < 0) throw std::runtime_error("sunsky: the sun is below the horizon (sky.cpp:239-240)");
    if (p.sunRadiusScale == 0) throw std::runtime_error("sunsky: sunRadiusScale = 0 (directional sun) is not supported");
    std::vector<double> img((size_t)w * h * 3, 0.0);
    // the sky, sky.cpp:409-447 (one model state per channel, each with its own albedo)
    for (int ch = 0; ch < 3; ++ch) {
        double cfg[9], radiance;
        hosekState(T["datasetRGB" + std::to_string(ch + 1)], T["datasetRGBRad" + std::to_string(ch + 1)], p.turbidity, p.albedo[ch], sunElevation, cfg, radiance);
        for (int y = 0; y < h; ++y) {
            const double theta = ((y + 0.5) * (M_PI / h)) / p.stretch;
            if (!(std::cos(theta) > 0)) continue;
            for (int x = 0; x < w; ++x) {
                const double phi = (x + 0.5) * (2 * M_PI / w);
                double cosGamma = std::cos(theta) * std::cos(sunZenith) + std::sin(theta) * std::sin(sunZenith) * std::cos(phi - sunAzimuth);
                cosGamma = std::min(1.0, std::max(-1.0, cosGamma));
                const double v = hosekRadiance(cfg, radiance, theta, std::acos(cosGamma)) / 106.856980;
                img[((size_t)y * w + x) * 3 + ch] = (double)(float)(std::max(v, 0.0) * skyScale);
            }
        }
    }
    // the sun, sunsky.cpp:165-205
    double sunRad[3];
    sunRadianceRGB(T, sunZenith, p.turbidity, sunRad);
    for (double &c : sunRad) c *= sunScale;
    const double sz = sunZenith * p.stretch;
    const double n[3] = {std::sin(sunAzimuth) * std::sin(sz), std::cos(sz), -std::cos(sunAzimuth) * std::sin(sz)};  // toSphere
    double c[3];  // coordinateSystem, util.cpp:592-601
    if (std::fabs(n[0]) > std::fabs(n[1])) { const double inv = 1.0 / std::sqrt(n[0] * n[0] + n[2] * n[2]); c[0] = n[2] * inv; c[1] = 0; c[2] = -n[0] * inv; }
    else { const double inv = 1.0 / std::sqrt(n[1] * n[1] + n[2] * n[2]); c[0] = 0; c[1] = n[2] * inv; c[2] = -n[1] * inv; }
    const double b[3] = {c[1] * n[2] - c[2] * n[1], c[2] * n[0] - c[0] * n[2], c[0] * n[1] - c[1] * n[0]};
    const double th0 = (0.5358 * 0.5) * (M_PI / 180.0);  // SUN_APP_RADIUS, sunsky.cpp:34
    const double cosCut = std::cos(th0 * p.sunRadiusScale);
    const long nSamples = (long)std::max(100.0, (double)((long)p.resolution * p.resolution / 2) * (0.5 * (1 - cosCut)) * 1000);
    double value[3];
    for (int k = 0; k < 3; ++k) value[k] = sunRad[k] * (2 * M_PI * (1 - std::cos(th0))) * (double)((long)w * h) / (2 * M_PI * M_PI * nSamples);
    for (long i = 0; i < nSamples; ++i) {
        float uf, vf;
        sample02((uint32_t)i, uf, vf);
        const double u = uf, v = vf;
        const double ct = (1 - u) + u * cosCut, st = std::sqrt(std::max(0.0, 1.0 - ct * ct)), ph = 2.0 * M_PI * v;  // squareToUniformCone, warp.cpp:54-63
        const double l[3] = {std::cos(ph) * st, std::sin(ph) * st, ct};
        double dir[3];
        for (int k = 0; k < 3; ++k) dir[k] = l[0] * b[k] + l[1] * c[k] + l[2] * n[k];  // Frame(n): s = b, t = c
        const double sinTheta = std::sqrt(std::max(0.0, 1.0 - dir[1] * dir[1]));
        double az = std::atan2(dir[0], -dir[2]);
        if (az < 0) az += 2 * M_PI;
        const double el = std::acos(std::min(1.0, std::max(-1.0, dir[1])));
        const long px = std::min<long>(w - 1, std::max<long>(0, (long)(az * (w / (2 * M_PI)))));
        const long py = std::min<long>(h - 1, std::max<long>(0, (long)(el * (h / M_PI))));
        for (int k = 0; k < 3; ++k) img[((size_t)py * w + px) * 3 + k] += value[k] / std::max(1e-3, sinTheta);
    }
    rgb.resize(img.size());
    for (size_t i = 0; i < img.size(); ++i) rgb[i] = (float)img[i];
    width = w; height = h;
}

}  // namespace sunsky
}  // namespace ppg
#endif
