"""The Mitsuba plug-in (mitsuba_plugin/guided_path_hip.cpp) cannot be BUILT here — Mitsuba needs boost, xerces, OpenEXR, Eigen — but it can be
PARSED: g++ -fsyntax-only against the reference's real headers in /root/reference/mitsuba/include, with throw-away stand-ins for the four boost
headers those headers pull in (a version number, BOOST_STATIC_ASSERT, scoped_ptr, filesystem::path).  This is a check of OUR file against the
reference's interfaces (Integrator, Scene, TriMesh, Film, Properties, the Log macro ...), not an oracle and not a build of the reference:
nothing is linked, nothing runs.  It found two bugs the name check of test_plugin_names.py could not: the host header's `Log` typedef
collided with Mitsuba's Log macro, and fresolver.h was not included.  Development container only (skipped where /root/reference is absent)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "practical-path-guiding_amd")
MITSUBA_INC = "/root/reference/mitsuba/include"

STUBS = {
    "boost/version.hpp": "#pragma once\n#define BOOST_VERSION 105400\n",
    "boost/static_assert.hpp": "#pragma once\n#define BOOST_STATIC_ASSERT(x) static_assert(x, #x)\n",
    "boost/scoped_ptr.hpp": "#pragma once\nnamespace boost { template <class T> class scoped_ptr { T *p; public: explicit scoped_ptr(T *q = 0) : p(q) {} "
                            "T *get() const { return p; } T *operator->() const { return p; } T &operator*() const { return *p; } void reset(T *q = 0) { p = q; } }; }\n",
    "boost/filesystem.hpp": "#pragma once\n#include <string>\nnamespace boost { namespace filesystem { class path { std::string s; public: path() {} "
                            "path(const std::string &x) : s(x) {} path(const char *x) : s(x) {} std::string string() const { return s; } path parent_path() const { return *this; } "
                            "path filename() const { return *this; } path extension() const { return *this; } path operator/(const path &o) const { return path(s + \"/\" + o.s); } "
                            "bool empty() const { return s.empty(); } }; } }\n",
    "boost/filesystem/fstream.hpp": "#pragma once\n#include <fstream>\nnamespace boost { namespace filesystem { typedef std::ifstream ifstream; typedef std::ofstream ofstream; typedef std::fstream fstream; } }\n",
}


@pytest.mark.skipif(not os.path.isdir(MITSUBA_INC) or shutil.which("g++") is None, reason="needs the reference's headers (development container) and g++")
def test_plugin_source_parses_against_the_references_headers(tmp_path):
    for name, text in STUBS.items():
        f = tmp_path / "stubs" / name
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    # the flags of the reference's own build: config-linux-gcc.py (-DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3, gnu++11: constants.h uses hex floats)
    cmd = ["g++", "-std=gnu++11", "-fsyntax-only", "-DSINGLE_PRECISION", "-DSPECTRUM_SAMPLES=3", "-DMTS_SSE", "-I", str(tmp_path / "stubs"), "-I", MITSUBA_INC,
           "-I", os.path.join(PKG, "host"), "-I", os.path.join(ROOT, "include"), os.path.join(PKG, "mitsuba_plugin", "guided_path_hip.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    errors = [line for line in r.stderr.splitlines() if "error" in line]
    assert r.returncode == 0 and not errors, "\n".join(errors[:20])
