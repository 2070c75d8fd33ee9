// tirt_obj.hip -- native Wavefront OBJ/MTL ingest (host code only; no device work).
//
// The reference loads meshes through the third-party PyWavefront 1.3.3 package
// (Scene.py:66-127: `pywavefront.Wavefront(filename)`, then one interleaved float list per
// material).  This file is the C++ side of `ti_raytrace_amd.ObjLoader`: the same grouping rules
// (see the module docstring of ObjLoader.py, whose pure-Python parser is kept as the checker in
// tests/test_host.py), with strtod for the numbers so that every coordinate is the same double
// Python's float() produces.  Teapot.obj (25k triangles): 0.13 s in Python, 0.03 s here.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "tirt_internal.h"

namespace {

struct ObjMat {
    std::string name;
    int is_default = 0;
    double diffuse[4] = {0.8, 0.8, 0.8, 1.0}, ambient[4] = {0.2, 0.2, 0.2, 1.0}, specular[4] = {0.0, 0.0, 0.0, 1.0},
           emissive[4] = {0.0, 0.0, 0.0, 1.0};
    double transparency = 1.0, optical_density = 1.0, shininess = 0.0;
    int format = 0;                 // 0 unset, else bit0: T2F, bit1: N3F  (+4 once decided): 4 V3F, 5 T2F_V3F, 6 N3F_V3F, 7 T2F_N3F_V3F
    std::vector<double> flat;
};

}  // namespace

struct tirt_obj {
    std::vector<ObjMat> mats;       // in first-appearance order (MTL file order, then on-the-spot defaults)
    std::map<std::string, int> by_name;
};

namespace {

bool read_file(const std::string &path, std::string &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t k;
    out.clear();
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, k);
    fclose(f);
    return true;
}

// Python str.split() on the part of a line before '#'
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r' || c == '\n'; }
void tokenize(const char *b, const char *e, std::vector<std::string> &tok)
{
    tok.clear();
    const char *p = b;
    while (p < e) {
        while (p < e && is_space(*p)) p++;
        if (p >= e || *p == '#') break;
        const char *q = p;
        while (q < e && !is_space(*q) && *q != '#') q++;
        tok.emplace_back(p, q);
        if (q < e && *q == '#') break;
        p = q;
    }
}
std::string join_from(const std::vector<std::string> &tok, size_t first)
{
    std::string s;
    for (size_t k = first; k < tok.size(); k++) { if (k > first) s += ' '; s += tok[k]; }
    return s;
}
bool to_double(const std::string &s, double &v)
{
    if (s.empty()) return false;
    char *end = nullptr;
    errno = 0;
    v = strtod(s.c_str(), &end);
    return end == s.c_str() + s.size();
}
bool to_long(const char *b, const char *e, long &v)
{
    if (b >= e) return false;
    std::string s(b, e);
    char *end = nullptr;
    v = strtol(s.c_str(), &end, 10);
    return end == s.c_str() + s.size();
}
#define OBJ_FAIL(msg) do { tirt::set_error(std::string("tirt_obj_load: ") + (msg) + " (" + path + ":" + std::to_string(line) + ")"); return TIRT_ERR_ARG; } while (0)

template <class F> int for_each_line(const std::string &text, F &&fn)
{
    const char *p = text.data(), *end = p + text.size();
    int line = 0;
    while (p < end) {
        const char *q = p;
        while (q < end && *q != '\n' && *q != '\r') q++;
        line++;
        int rc = fn(p, q, line);
        if (rc) return rc;
        if (q < end && *q == '\r' && q + 1 < end && q[1] == '\n') q++;
        p = q + 1;
    }
    return 0;
}

int parse_mtl(const std::string &path, tirt_obj *o)
{
    std::string text;
    if (!read_file(path, text)) return 0;            // ObjLoader.py: a missing MTL file is skipped
    std::vector<std::string> tok;
    int cur = -1;
    return for_each_line(text, [&](const char *b, const char *e, int line) -> int {
        tokenize(b, e, tok);
        if (tok.empty()) return 0;
        const std::string &key = tok[0];
        if (key == "newmtl") {
            ObjMat m; m.name = join_from(tok, 1);
            auto it = o->by_name.find(m.name);
            if (it != o->by_name.end()) { o->mats[it->second] = m; cur = it->second; }     // dict assignment: same slot, new object
            else { o->by_name[m.name] = (int)o->mats.size(); o->mats.push_back(m); cur = (int)o->mats.size() - 1; }
            return 0;
        }
        if (cur < 0) return 0;
        ObjMat &m = o->mats[cur];
        auto rgb = [&](double *dst) -> bool {
            if (tok.size() < 4) return false;
            double a, b2, c;
            if (!to_double(tok[1], a) || !to_double(tok[2], b2) || !to_double(tok[3], c)) return false;
            dst[0] = a; dst[1] = b2; dst[2] = c; dst[3] = 1.0; return true;
        };
        auto scalar = [&](double &dst) -> bool { return tok.size() >= 2 && to_double(tok[1], dst); };
        if (key == "Kd") { if (!rgb(m.diffuse)) OBJ_FAIL("bad Kd"); }
        else if (key == "Ka") { if (!rgb(m.ambient)) OBJ_FAIL("bad Ka"); }
        else if (key == "Ks") { if (!rgb(m.specular)) OBJ_FAIL("bad Ks"); }
        else if (key == "Ke") { if (!rgb(m.emissive)) OBJ_FAIL("bad Ke"); }
        else if (key == "d") { if (!scalar(m.transparency)) OBJ_FAIL("bad d"); }
        else if (key == "Tr") { double t; if (!scalar(t)) OBJ_FAIL("bad Tr"); m.transparency = 1.0 - t; }
        else if (key == "Ni") { if (!scalar(m.optical_density)) OBJ_FAIL("bad Ni"); }
        else if (key == "Ns") { if (!scalar(m.shininess)) OBJ_FAIL("bad Ns"); }
        return 0;
    });
}

struct Corner { long v, t, n; };

}  // namespace

extern "C" {

int tirt_obj_load(const char *path_c, tirt_obj **out)
{
    if (!path_c || !out) { tirt::set_error("tirt_obj_load: null argument"); return TIRT_ERR_ARG; }
    const std::string path(path_c);
    std::string text;
    if (!read_file(path, text)) { tirt::set_error("tirt_obj_load: cannot open " + path); return TIRT_ERR_ARG; }
    tirt_obj *o = new tirt_obj();
    std::string base;
    { size_t s = path.find_last_of('/'); if (s != std::string::npos) base = path.substr(0, s); }
    std::vector<double> pos, nor, tex;                // xyz, xyz, uv
    std::vector<std::string> tok;
    std::vector<Corner> corners;
    int current = -1;
    int rc = for_each_line(text, [&](const char *b, const char *e, int line) -> int {
        tokenize(b, e, tok);
        if (tok.empty()) return 0;
        const std::string &key = tok[0];
        if (key == "v" || key == "vn") {
            double x, y, z;
            if (tok.size() < 4 || !to_double(tok[1], x) || !to_double(tok[2], y) || !to_double(tok[3], z)) OBJ_FAIL("bad " + key);
            std::vector<double> &dst = (key == "v") ? pos : nor;
            dst.push_back(x); dst.push_back(y); dst.push_back(z);
        } else if (key == "vt") {
            double u, v = 0.0;
            if (tok.size() < 2 || !to_double(tok[1], u)) OBJ_FAIL("bad vt");
            if (tok.size() > 2 && !to_double(tok[2], v)) OBJ_FAIL("bad vt");
            tex.push_back(u); tex.push_back(v);
        } else if (key == "mtllib") {
            const std::string name = join_from(tok, 1);
            const std::string mtl = (!name.empty() && name[0] == '/') ? name : (base.empty() ? name : base + "/" + name);
            int r = parse_mtl(mtl, o);
            if (r) return r;
        } else if (key == "usemtl") {
            const std::string name = join_from(tok, 1);
            auto it = o->by_name.find(name);
            if (it == o->by_name.end()) {
                ObjMat m; m.name = name; m.is_default = 1;
                o->by_name[name] = (int)o->mats.size(); o->mats.push_back(m); current = (int)o->mats.size() - 1;
            } else current = it->second;
        } else if (key == "f") {
            if (current < 0) {
                ObjMat m; m.name = "default" + std::to_string(o->mats.size()); m.is_default = 1;
                o->by_name[m.name] = (int)o->mats.size(); o->mats.push_back(m); current = (int)o->mats.size() - 1;
            }
            corners.clear();
            bool has_vt = false, has_vn = false;
            for (size_t k = 1; k < tok.size(); k++) {
                const std::string &t = tok[k];
                const char *p0 = t.data(), *pe = p0 + t.size();
                const char *s1 = (const char *)memchr(p0, '/', t.size());
                const char *s2 = s1 ? (const char *)memchr(s1 + 1, '/', (size_t)(pe - s1 - 1)) : nullptr;
                Corner c = {0, 0, 0};
                if (!to_long(p0, s1 ? s1 : pe, c.v)) OBJ_FAIL("bad face index");
                if (s1 && (s2 ? s2 : pe) > s1 + 1 && !to_long(s1 + 1, s2 ? s2 : pe, c.t)) OBJ_FAIL("bad face index");
                if (s2) {
                    const char *s3 = (const char *)memchr(s2 + 1, '/', (size_t)(pe - s2 - 1));
                    if ((s3 ? s3 : pe) > s2 + 1 && !to_long(s2 + 1, s3 ? s3 : pe, c.n)) OBJ_FAIL("bad face index");
                }
                has_vt |= c.t != 0; has_vn |= c.n != 0;
                corners.push_back(c);
            }
            ObjMat &m = o->mats[current];
            if (m.format == 0) m.format = 4 | (has_vt ? 1 : 0) | (has_vn ? 2 : 0);
            const bool want_vt = (m.format & 1) != 0, want_vn = (m.format & 2) != 0;
            // Python indexing: 1-based positives, negatives count from the end, 0 only for "absent"
            auto fetch = [&](const std::vector<double> &src, int width, long idx, double *dst) -> bool {
                const long count = (long)(src.size() / (size_t)width);
                long k = idx > 0 ? idx - 1 : count + idx;
                if (k < 0 || k >= count) return false;
                for (int j = 0; j < width; j++) dst[j] = src[(size_t)k * width + j];
                return true;
            };
            auto emit = [&](const Corner &c) -> bool {
                double tmp[3];
                if (want_vt) { if (c.t) { if (!fetch(tex, 2, c.t, tmp)) return false; } else { tmp[0] = tmp[1] = 0.0; } m.flat.push_back(tmp[0]); m.flat.push_back(tmp[1]); }
                if (want_vn) { if (c.n) { if (!fetch(nor, 3, c.n, tmp)) return false; } else { tmp[0] = tmp[1] = tmp[2] = 0.0; } m.flat.push_back(tmp[0]); m.flat.push_back(tmp[1]); m.flat.push_back(tmp[2]); }
                if (c.v == 0 || !fetch(pos, 3, c.v, tmp)) { if (c.v == 0 && !pos.empty()) { tmp[0] = pos[0]; tmp[1] = pos[1]; tmp[2] = pos[2]; } else return false; }
                m.flat.push_back(tmp[0]); m.flat.push_back(tmp[1]); m.flat.push_back(tmp[2]);
                return true;
            };
            for (size_t k = 2; k < corners.size(); k++)
                if (!emit(corners[0]) || !emit(corners[k - 1]) || !emit(corners[k])) OBJ_FAIL("face index out of range");
        }
        return 0;
    });
    if (rc) { delete o; return rc; }
    *out = o;
    return TIRT_OK;
}

void tirt_obj_free(tirt_obj *o) { delete o; }

int tirt_obj_material_count(const tirt_obj *o) { return o ? (int)o->mats.size() : 0; }

/* params[19] = diffuse rgba, ambient rgba, specular rgba, emissive rgba, transparency, optical_density, shininess;
 * vertex_format: 4 V3F, 5 T2F_V3F, 6 N3F_V3F, 7 T2F_N3F_V3F, 0 = the material owns no face */
int tirt_obj_material_info(const tirt_obj *o, int i, char *name, int name_cap, double *params, int *vertex_format, int *is_default,
                           long long *n_floats)
{
    if (!o || i < 0 || i >= (int)o->mats.size()) { tirt::set_error("tirt_obj_material_info: bad index"); return TIRT_ERR_ARG; }
    const ObjMat &m = o->mats[i];
    if (name && name_cap > 0) { strncpy(name, m.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (params) {
        for (int k = 0; k < 4; k++) { params[k] = m.diffuse[k]; params[4 + k] = m.ambient[k]; params[8 + k] = m.specular[k]; params[12 + k] = m.emissive[k]; }
        params[16] = m.transparency; params[17] = m.optical_density; params[18] = m.shininess;
    }
    if (vertex_format) *vertex_format = m.format;
    if (is_default) *is_default = m.is_default;
    if (n_floats) *n_floats = (long long)m.flat.size();
    return TIRT_OK;
}

int tirt_obj_material_vertices(const tirt_obj *o, int i, double *out, long long n)
{
    if (!o || i < 0 || i >= (int)o->mats.size() || !out || n != (long long)o->mats[i].flat.size()) {
        tirt::set_error("tirt_obj_material_vertices: bad index or size"); return TIRT_ERR_ARG;
    }
    if (n) memcpy(out, o->mats[i].flat.data(), sizeof(double) * (size_t)n);
    return TIRT_OK;
}

}  // extern "C"
