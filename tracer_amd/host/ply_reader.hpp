// ply_reader.hpp -- the PLY files `Shape "plymesh"` names (pbrt-v3 scenes keep their big meshes in them).
//
// The reference vendors minipbrt for its pbrt to-do (RT_Metal/Tracer/minipbrt.h:1140-1150 PLYMesh, :71-73 "PLY files are
// not automatically loaded"); its PLYMesh::triangle_mesh() (minipbrt.cpp:4380-4450) defines what a renderer gets out of
// such a file: the `vertex` element's x y z (+ nx ny nz, + u v | s t | texture_u texture_v | texture_s texture_t) and the
// `face` element's vertex_indices / vertex_index list, triangles as they are, quads split (0 1 3) (2 3 1).  This reader
// returns exactly that (tests/test_pbrt_reader.py compares it with minipbrt compiled in place); polygons with more than
// four corners are fanned from their first corner (minipbrt clips ears: the one declared difference).
// Formats: ascii, binary_little_endian, binary_big_endian; any scalar property type; list properties other than the
// face indices are skipped.
#pragma once

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace trc {

struct PlyMesh {
    std::vector<float> P, N, UV;        // 3 / 3 / 2 floats per vertex (N, UV empty when the file has none)
    std::vector<uint32_t> indices;      // 3 per triangle
};

namespace ply_detail {

enum Type { I8, U8, I16, U16, I32, U32, F32, F64, Bad };
inline Type type_of(const std::string& s) {
    if (s == "char" || s == "int8") return I8;
    if (s == "uchar" || s == "uint8") return U8;
    if (s == "short" || s == "int16") return I16;
    if (s == "ushort" || s == "uint16") return U16;
    if (s == "int" || s == "int32") return I32;
    if (s == "uint" || s == "uint32") return U32;
    if (s == "float" || s == "float32") return F32;
    if (s == "double" || s == "float64") return F64;
    return Bad;
}
inline size_t size_of(Type t) { return t == I8 || t == U8 ? 1 : t == I16 || t == U16 ? 2 : t == F64 ? 8 : 4; }

struct Property { std::string name; bool list = false; Type count_type = U8, type = F32; };
struct Element { std::string name; size_t count = 0; std::vector<Property> props; };

struct Cursor {
    const std::string& d;
    size_t i;
    bool ascii, big;
    bool ok = true;
    // one scalar of type t as a double (ascii: the next whitespace-separated token)
    double scalar(Type t) {
        if (ascii) {
            while (i < d.size() && std::isspace((unsigned char)d[i])) ++i;
            if (i >= d.size()) { ok = false; return 0; }
            char* end = nullptr;
            const double v = std::strtod(d.c_str() + i, &end);
            if (end == d.c_str() + i) { ok = false; return 0; }
            i = (size_t)(end - d.c_str());
            return v;
        }
        const size_t n = size_of(t);
        if (i + n > d.size()) { ok = false; return 0; }
        unsigned char b[8];
        for (size_t k = 0; k < n; ++k) b[k] = (unsigned char)d[i + (big ? n - 1 - k : k)];
        i += n;
        switch (t) {
            case I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
            case U8: return b[0];
            case I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
            case U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
            case I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
            case U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
            case F32: { float v; std::memcpy(&v, b, 4); return v; }
            default: { double v; std::memcpy(&v, b, 8); return v; }
        }
    }
};

}  // namespace ply_detail

inline bool read_ply(const std::string& path, PlyMesh& out) {
    using namespace ply_detail;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string data;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) data.append(chunk, got);
    std::fclose(f);

    // header: lines up to "end_header"
    size_t pos = 0;
    auto line = [&](std::string& l) {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        l = data.substr(pos, e - pos);
        if (!l.empty() && l.back() == '\r') l.pop_back();
        pos = e + 1;
        return true;
    };
    auto words = [](const std::string& l) {
        std::vector<std::string> w;
        size_t i = 0;
        while (i < l.size()) {
            while (i < l.size() && std::isspace((unsigned char)l[i])) ++i;
            size_t e = i;
            while (e < l.size() && !std::isspace((unsigned char)l[e])) ++e;
            if (e > i) w.push_back(l.substr(i, e - i));
            i = e;
        }
        return w;
    };
    std::string l;
    if (!line(l) || l != "ply") return false;
    bool ascii = false, big = false, have_format = false, ended = false;
    std::vector<Element> elems;
    while (line(l)) {
        const std::vector<std::string> w = words(l);
        if (w.empty() || w[0] == "comment" || w[0] == "obj_info") continue;
        if (w[0] == "end_header") { ended = true; break; }
        if (w[0] == "format" && w.size() >= 2) {
            ascii = w[1] == "ascii"; big = w[1] == "binary_big_endian";
            if (!ascii && !big && w[1] != "binary_little_endian") return false;
            have_format = true;
        } else if (w[0] == "element" && w.size() >= 3) {
            Element e; e.name = w[1];
            char* end = nullptr;
            const unsigned long long n = std::strtoull(w[2].c_str(), &end, 10);
            if (end == w[2].c_str() || n > (1ull << 31)) return false;
            e.count = (size_t)n;
            elems.push_back(e);
        } else if (w[0] == "property" && !elems.empty()) {
            Property p;
            if (w.size() >= 5 && w[1] == "list") { p.list = true; p.count_type = type_of(w[2]); p.type = type_of(w[3]); p.name = w[4]; }
            else if (w.size() >= 3) { p.type = type_of(w[1]); p.name = w[2]; }
            else return false;
            if (p.type == Bad || p.count_type == Bad || p.count_type == F32 || p.count_type == F64) return false;
            elems.back().props.push_back(p);
        } else return false;
    }
    if (!ended || !have_format) return false;

    Cursor c{data, pos, ascii, big};
    bool got_vertices = false, got_faces = false;
    size_t n_vertices = 0;
    for (const Element& e : elems) {
        if (e.name == "vertex" && !got_vertices) {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t k = 0; k < e.props.size(); ++k) {
                const std::string& n = e.props[k].name;
                if (e.props[k].list) continue;
                if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
                else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
                else if (iu < 0 && (n == "u" || n == "s" || n == "texture_u" || n == "texture_s")) iu = (int)k;
                else if (iv < 0 && (n == "v" || n == "t" || n == "texture_v" || n == "texture_t")) iv = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) return false;
            const bool has_n = inx >= 0 && iny >= 0 && inz >= 0, has_uv = iu >= 0 && iv >= 0;
            // a vertex costs at least one byte per property in any format: a count the file cannot hold is a lie
            if (e.count > data.size()) return false;
            out.P.resize(e.count * 3);
            if (has_n) out.N.resize(e.count * 3);
            if (has_uv) out.UV.resize(e.count * 2);
            std::vector<double> row(e.props.size());
            for (size_t r = 0; r < e.count; ++r) {
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const Property& p = e.props[k];
                    if (p.list) { const double n = c.scalar(p.count_type); if (!c.ok || !(n >= 0 && n <= 1e6)) return false; for (double j = 0; j < n && c.ok; ++j) c.scalar(p.type); row[k] = 0; }
                    else row[k] = c.scalar(p.type);
                }
                if (!c.ok) return false;
                out.P[3 * r] = (float)row[ix]; out.P[3 * r + 1] = (float)row[iy]; out.P[3 * r + 2] = (float)row[iz];
                if (has_n) { out.N[3 * r] = (float)row[inx]; out.N[3 * r + 1] = (float)row[iny]; out.N[3 * r + 2] = (float)row[inz]; }
                if (has_uv) { out.UV[2 * r] = (float)row[iu]; out.UV[2 * r + 1] = (float)row[iv]; }
            }
            n_vertices = e.count;
            got_vertices = true;
        } else if (e.name == "face" && !got_faces) {
            int il = -1;
            for (size_t k = 0; k < e.props.size(); ++k)
                if (e.props[k].list && (e.props[k].name == "vertex_indices" || e.props[k].name == "vertex_index")) { il = (int)k; break; }
            if (il < 0 || !got_vertices) return false;
            if (e.count > data.size()) return false;
            std::vector<int64_t> poly;
            for (size_t r = 0; r < e.count; ++r) {
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const Property& p = e.props[k];
                    if (!p.list) { c.scalar(p.type); continue; }
                    const double n = c.scalar(p.count_type);
                    if (!c.ok || !(n >= 0 && n <= 1e6)) return false;
                    if ((int)k != il) { for (double j = 0; j < n && c.ok; ++j) c.scalar(p.type); continue; }
                    poly.clear();
                    // the index is range-checked as a double BEFORE the cast: a float-typed list (or ascii "nan" / "1e300")
                    // converted out of range is undefined behaviour, not merely a bad index
                    for (double j = 0; j < n && c.ok; ++j) {
                        const double v = c.scalar(p.type);
                        if (!(v >= 0 && v < (double)n_vertices)) return false;
                        poly.push_back((int64_t)v);
                    }
                }
                if (!c.ok) return false;
                for (int64_t v : poly) if (v < 0 || (size_t)v >= n_vertices) return false;
                auto tri = [&](size_t a, size_t b, size_t cc) { out.indices.push_back((uint32_t)poly[a]); out.indices.push_back((uint32_t)poly[b]); out.indices.push_back((uint32_t)poly[cc]); };
                if (poly.size() == 3) tri(0, 1, 2);
                else if (poly.size() == 4) { tri(0, 1, 3); tri(2, 3, 1); }
                else for (size_t k = 1; k + 1 < poly.size(); ++k) tri(0, k, k + 1);
            }
            got_faces = true;
        } else {                                               // an element nobody asked for: step over its rows
            if (e.count > data.size()) return false;
            for (size_t r = 0; r < e.count && c.ok; ++r)
                for (const Property& p : e.props) {
                    if (p.list) { const double n = c.scalar(p.count_type); if (!c.ok || !(n >= 0 && n <= 1e6)) return false; for (double j = 0; j < n && c.ok; ++j) c.scalar(p.type); }
                    else c.scalar(p.type);
                }
            if (!c.ok) return false;
        }
        if (got_vertices && got_faces) break;
    }
    return got_vertices && got_faces && !out.indices.empty();
}

}  // namespace trc
