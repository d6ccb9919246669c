// mesh.cpp -- triangle mesh inputs for the mesh scenes (BASELINE configs 3 and 4).
//
// The reference loads its mesh through Apple's ModelIO with 32-byte vertices
// {float3 position, float3 normal, float2 uv} and adds smooth normals
// (RT_Metal/Tracer/AAPLRenderer.mm:474-511; vertex layout Common.hh:32-36).  ModelIO does not
// exist here, so this file provides (a) a minimal Wavefront OBJ reader producing the same
// 32-byte vertex + u32 index arrays, and (b) procedural meshes used on the GPU box, where the
// reference's asset files do not travel.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "host_math.hpp"
#include "pbrt_text.hpp"
#include "ply_reader.hpp"

using namespace trc;

struct trc_host_mesh {
    std::vector<trc_TriangleVertex> vertices;
    std::vector<uint32_t> indices;
};

namespace {

// area-weighted smooth normals from the triangles (used when a file carries none)
void generate_normals(trc_host_mesh& m) {
    std::vector<trc_float3> acc(m.vertices.size(), f3(0.0f));
    for (size_t t = 0; t + 2 < m.indices.size(); t += 3) {
        const uint32_t ia = m.indices[t], ib = m.indices[t + 1], ic = m.indices[t + 2];
        const trc_float3 a = f3(m.vertices[ia].v[0], m.vertices[ia].v[1], m.vertices[ia].v[2]);
        const trc_float3 b = f3(m.vertices[ib].v[0], m.vertices[ib].v[1], m.vertices[ib].v[2]);
        const trc_float3 c = f3(m.vertices[ic].v[0], m.vertices[ic].v[1], m.vertices[ic].v[2]);
        const trc_float3 fn = cross(b - a, c - a);
        acc[ia] = acc[ia] + fn; acc[ib] = acc[ib] + fn; acc[ic] = acc[ic] + fn;
    }
    for (size_t i = 0; i < m.vertices.size(); ++i) {
        float len = length(acc[i]);
        trc_float3 n = len > 0.0f ? acc[i] / len : f3(0, 1, 0);
        m.vertices[i].n[0] = n.x; m.vertices[i].n[1] = n.y; m.vertices[i].n[2] = n.z;
    }
}

int resolve(long idx, size_t count) {   // OBJ indices are 1-based, negative = relative to the end
    if (idx > 0) return (int)idx - 1;
    if (idx < 0) return (int)((long)count + idx);
    return -1;
}

}  // namespace

extern "C" {

trc_status trc_host_mesh_load_obj(const char* path, trc_host_mesh** out) {
    if (!path || !out) return TRC_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "rb");
    if (!f) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* m = new (std::nothrow) trc_host_mesh();
    if (!m) { std::fclose(f); return TRC_ERR_OOM; }

    std::vector<trc_float3> pos, nor;
    std::vector<trc_float2> tex;
    std::map<std::tuple<int, int, int>, uint32_t> remap;   // (v, vt, vn) -> unified vertex
    bool any_normal = false;
    std::vector<uint32_t> face;
    char line[4096];
    while (std::fgets(line, sizeof line, f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float x = 0, y = 0, z = 0; std::sscanf(p + 2, "%f %f %f", &x, &y, &z); pos.push_back(f3(x, y, z));
        } else if (p[0] == 'v' && p[1] == 'n') {
            float x = 0, y = 0, z = 0; std::sscanf(p + 3, "%f %f %f", &x, &y, &z); nor.push_back(f3(x, y, z));
        } else if (p[0] == 'v' && p[1] == 't') {
            trc_float2 t; t.x = 0; t.y = 0; std::sscanf(p + 3, "%f %f", &t.x, &t.y); tex.push_back(t);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            face.clear();
            p += 2;
            while (*p) {
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                long vi = std::strtol(p, &p, 10), ti = 0, ni = 0;
                if (*p == '/') { ++p; if (*p != '/') ti = std::strtol(p, &p, 10); if (*p == '/') { ++p; ni = std::strtol(p, &p, 10); } }
                const int v = resolve(vi, pos.size()), t = resolve(ti, tex.size()), n = resolve(ni, nor.size());
                if (v < 0 || v >= (int)pos.size()) { std::fclose(f); delete m; return TRC_ERR_INVALID_ARG; }
                auto key = std::make_tuple(v, t, n);
                auto it = remap.find(key);
                if (it == remap.end()) {
                    trc_TriangleVertex e;
                    std::memset(&e, 0, sizeof e);
                    e.v[0] = pos[v].x; e.v[1] = pos[v].y; e.v[2] = pos[v].z;
                    if (n >= 0 && n < (int)nor.size()) { e.n[0] = nor[n].x; e.n[1] = nor[n].y; e.n[2] = nor[n].z; any_normal = true; }
                    if (t >= 0 && t < (int)tex.size()) { e.uv[0] = tex[t].x; e.uv[1] = tex[t].y; }
                    it = remap.emplace(key, (uint32_t)m->vertices.size()).first;
                    m->vertices.push_back(e);
                }
                face.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {   // fan triangulation
                m->indices.push_back(face[0]); m->indices.push_back(face[k]); m->indices.push_back(face[k + 1]);
            }
        }
    }
    std::fclose(f);
    if (m->indices.empty()) { delete m; return TRC_ERR_INVALID_ARG; }
    if (!any_normal) generate_normals(*m);
    *out = m;
    return TRC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- pbrt-v3 triangle meshes
// "Support pbrt-v3 file format" is the reference's unchecked to-do (RT_Metal/README.md:57); it vendors minipbrt for
// it and so far reads one medium through it (AAPLRenderer.mm:629-636).  This reader takes what the mesh slot of the
// scene needs from a pbrt-v3 file: every `Shape "trianglemesh"` outside object definitions, its points through the
// current transformation matrix (Identity / Translate / Scale / Rotate / LookAt / Transform / ConcatTransform /
// CoordinateSystem / CoordSysTransform, Attribute and Transform stacks, WorldBegin, Include), normals through the
// inverse transpose, uv (`uv` or `st`), merged into one 32-byte-vertex mesh; shapes without normals get smooth
// ones.  tests/test_pbrt_reader.py checks points, indices and matrices against the reference's minipbrt.
extern "C" trc_status trc_host_mesh_load_pbrt(const char* path, trc_host_mesh** out) {
    if (!path || !out) return TRC_ERR_INVALID_ARG;
    *out = nullptr;
    std::string text;
    if (!read_pbrt_text(path, 0, text)) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* mesh = new (std::nothrow) trc_host_mesh();
    if (!mesh) return TRC_ERR_OOM;
    PbrtLexer lx(text);
    M4 ctm = m4_identity();
    std::vector<M4> stack;
    std::map<std::string, M4> named;
    int object_depth = 0;
    std::vector<float> a;
    std::vector<PbrtParam> params;
    auto fail = [&]() { delete mesh; return TRC_ERR_INVALID_ARG; };
    for (;;) {
        PbrtToken t = lx.next();
        if (t.kind == PbrtToken::End) break;
        if (t.kind != PbrtToken::Word) continue;                   // stray value of a directive we do not interpret
        const std::string& d = t.text;
        if (d == "Identity") ctm = m4_identity();
        else if (d == "Translate") {
            if (!read_numbers(lx, 3, a)) return fail();
            M4 m = m4_identity(); m.m[0][3] = a[0]; m.m[1][3] = a[1]; m.m[2][3] = a[2];
            ctm = m4_mul(ctm, m);
        } else if (d == "Scale") {
            if (!read_numbers(lx, 3, a)) return fail();
            M4 m = m4_identity(); m.m[0][0] = a[0]; m.m[1][1] = a[1]; m.m[2][2] = a[2];
            ctm = m4_mul(ctm, m);
        } else if (d == "Rotate") {                                 // pbrt-v3 Rotate(theta, axis)
            if (!read_numbers(lx, 4, a)) return fail();
            ctm = m4_mul(ctm, m4_rotate(a[0], a[1], a[2], a[3]));
        } else if (d == "LookAt") {                                 // pbrt-v3 LookAt: the world-to-camera matrix
            if (!read_numbers(lx, 9, a)) return fail();
            M4 w2c;
            if (!m4_look_at(a.data(), w2c)) return fail();
            ctm = m4_mul(ctm, w2c);
        } else if (d == "Transform" || d == "ConcatTransform") {    // 16 numbers, column-major
            if (!read_numbers(lx, 16, a)) return fail();
            M4 m;
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m.m[r][c] = a[(size_t)c * 4 + r];
            ctm = d == "Transform" ? m : m4_mul(ctm, m);
        } else if (d == "CoordinateSystem" || d == "CoordSysTransform") {
            PbrtToken n = lx.next();
            if (n.kind != PbrtToken::String) return fail();
            if (d == "CoordinateSystem") named[n.text] = ctm;
            else { auto it = named.find(n.text); if (it != named.end()) ctm = it->second; }
        } else if (d == "Camera") {
            // the reference's parser names the CURRENT matrix "camera" (minipbrt.cpp:7056-7057); pbrt-v3 itself stores
            // its inverse (api.cpp pbrtCamera).  The reference's reading is kept: a file means here what it means there.
            lx.next();
            if (!read_params(lx, params)) return fail();
            named["camera"] = ctm;
        } else if (d == "WorldBegin") { ctm = m4_identity(); named["world"] = ctm; }
        else if (d == "AttributeBegin" || d == "TransformBegin") stack.push_back(ctm);
        else if (d == "AttributeEnd" || d == "TransformEnd") { if (stack.empty()) return fail(); ctm = stack.back(); stack.pop_back(); }
        else if (d == "ObjectBegin") { lx.next(); ++object_depth; stack.push_back(ctm); }
        else if (d == "ObjectEnd") { if (object_depth == 0 || stack.empty()) return fail(); --object_depth; ctm = stack.back(); stack.pop_back(); }
        else if (d == "Shape") {
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params(lx, params)) return fail();
            if (kind.text != "trianglemesh" || object_depth > 0) continue;
            const PbrtParam *P = nullptr, *N = nullptr, *UV = nullptr, *I = nullptr;
            for (const PbrtParam& q : params) {
                if (q.name == "P") P = &q;
                else if (q.name == "N") N = &q;
                else if (q.name == "uv" || q.name == "st") UV = &q;
                else if (q.name == "indices") I = &q;
            }
            if (!P || P->numbers.size() % 3 != 0 || P->numbers.empty()) return fail();
            const size_t nv = P->numbers.size() / 3;
            std::vector<double> seq;                                // a lone triangle may omit its indices
            if (!I) { if (nv != 3) return fail(); seq = {0, 1, 2}; }
            const std::vector<double>& idx = I ? I->numbers : seq;
            if (idx.size() % 3 != 0) return fail();
            if (N && N->numbers.size() != nv * 3) N = nullptr;
            if (UV && UV->numbers.size() != nv * 2) UV = nullptr;
            M4 inv;
            const bool has_inv = m4_inverse(ctm, inv);
            trc_host_mesh part;
            part.vertices.resize(nv);
            for (size_t v = 0; v < nv; ++v) {
                const float x = (float)P->numbers[3 * v], y = (float)P->numbers[3 * v + 1], z = (float)P->numbers[3 * v + 2];
                float q[4];
                for (int r = 0; r < 4; ++r) q[r] = ctm.m[r][0] * x + ctm.m[r][1] * y + ctm.m[r][2] * z + ctm.m[r][3];
                trc_TriangleVertex& tv = part.vertices[v];
                const float w = q[3];
                tv.v[0] = w == 1 ? q[0] : q[0] / w; tv.v[1] = w == 1 ? q[1] : q[1] / w; tv.v[2] = w == 1 ? q[2] : q[2] / w;
                tv.n[0] = tv.n[1] = tv.n[2] = 0; tv.uv[0] = tv.uv[1] = 0;
                if (N && has_inv) {                                 // (M^-1)^T n
                    const float nx = (float)N->numbers[3 * v], ny = (float)N->numbers[3 * v + 1], nz = (float)N->numbers[3 * v + 2];
                    for (int r = 0; r < 3; ++r) tv.n[r] = inv.m[0][r] * nx + inv.m[1][r] * ny + inv.m[2][r] * nz;
                }
                if (UV) { tv.uv[0] = (float)UV->numbers[2 * v]; tv.uv[1] = (float)UV->numbers[2 * v + 1]; }
            }
            for (double k : idx) {
                if (!(k >= 0 && k < (double)nv)) return fail();   // also rejects NaN
                part.indices.push_back((uint32_t)k);
            }
            if (!N || !has_inv) generate_normals(part);
            const uint32_t base = (uint32_t)mesh->vertices.size();
            mesh->vertices.insert(mesh->vertices.end(), part.vertices.begin(), part.vertices.end());
            for (uint32_t k : part.indices) mesh->indices.push_back(base + k);
        } else {
            // any other directive: skip its leading strings (type, names) / numbers and its parameter list
            for (;;) {
                const PbrtToken nx = lx.peek();
                if (nx.kind == PbrtToken::Number || (nx.kind == PbrtToken::String && nx.text.find_first_of(" \t") == std::string::npos)) lx.next();
                else break;
            }
            if (!read_params(lx, params)) return fail();
        }
    }
    if (mesh->indices.empty()) return fail();
    *out = mesh;
    return TRC_OK;
}

extern "C" {

trc_status trc_host_mesh_make_ball(uint32_t n_lat, uint32_t n_lon, float bump, trc_host_mesh** out) {
    if (!out || n_lat < 2 || n_lon < 3) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* m = new (std::nothrow) trc_host_mesh();
    if (!m) return TRC_ERR_OOM;
    const double pi = 3.14159265358979323846;
    for (uint32_t i = 0; i <= n_lat; ++i) {
        const double theta = pi * i / n_lat;
        for (uint32_t j = 0; j <= n_lon; ++j) {
            const double phi = 2 * pi * j / n_lon;
            const double r = 1.0 + bump * std::sin(7 * theta) * std::sin(5 * phi);
            trc_TriangleVertex e;
            std::memset(&e, 0, sizeof e);
            e.v[0] = (float)(r * std::sin(theta) * std::cos(phi));
            e.v[1] = (float)(r * std::cos(theta));
            e.v[2] = (float)(r * std::sin(theta) * std::sin(phi));
            e.uv[0] = (float)j / n_lon; e.uv[1] = (float)i / n_lat;
            m->vertices.push_back(e);
        }
    }
    const uint32_t stride = n_lon + 1;
    for (uint32_t i = 0; i < n_lat; ++i)
        for (uint32_t j = 0; j < n_lon; ++j) {
            const uint32_t a = i * stride + j, b = a + 1, c = a + stride, d = c + 1;
            m->indices.insert(m->indices.end(), {a, b, c});
            m->indices.insert(m->indices.end(), {b, d, c});
        }
    generate_normals(*m);
    *out = m;
    return TRC_OK;
}

// the triangles of a PLY file (ply_reader.hpp) as a mesh handle: what minipbrt's PLYMesh::triangle_mesh() hands a renderer
// (minipbrt.cpp:4380-4450); smooth normals when the file carries none
trc_status trc_host_mesh_load_ply(const char* path, trc_host_mesh** out) {
    if (!path || !out) return TRC_ERR_INVALID_ARG;
    *out = nullptr;
    PlyMesh pm;
    if (!read_ply(path, pm)) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* m = new (std::nothrow) trc_host_mesh();
    if (!m) return TRC_ERR_OOM;
    const size_t nv = pm.P.size() / 3;
    m->vertices.resize(nv);
    for (size_t v = 0; v < nv; ++v) {
        trc_TriangleVertex& t = m->vertices[v];
        std::memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; ++k) t.v[k] = pm.P[3 * v + k];
        if (!pm.N.empty()) for (int k = 0; k < 3; ++k) t.n[k] = pm.N[3 * v + k];
        if (!pm.UV.empty()) { t.uv[0] = pm.UV[2 * v]; t.uv[1] = pm.UV[2 * v + 1]; }
    }
    m->indices = pm.indices;
    if (pm.N.empty()) generate_normals(*m);
    *out = m;
    return TRC_OK;
}

trc_status trc_host_mesh_from_arrays(const trc_TriangleVertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                                     uint32_t n_indices, trc_host_mesh** out) {
    if (!vertices || !indices || !out || n_vertices == 0 || n_indices < 3 || n_indices % 3) return TRC_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_indices; ++i)
        if (indices[i] >= n_vertices) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* m = new (std::nothrow) trc_host_mesh();
    if (!m) return TRC_ERR_OOM;
    m->vertices.assign(vertices, vertices + n_vertices);
    m->indices.assign(indices, indices + n_indices);
    *out = m;
    return TRC_OK;
}

trc_status trc_host_mesh_replicate(const trc_host_mesh* src, uint32_t k, float spacing, trc_host_mesh** out) {
    if (!src || !out || k == 0) return TRC_ERR_INVALID_ARG;
    trc_host_mesh* m = new (std::nothrow) trc_host_mesh();
    if (!m) return TRC_ERR_OOM;
    const size_t nv = src->vertices.size();
    m->vertices.reserve(nv * k * k);
    m->indices.reserve(src->indices.size() * k * k);
    for (uint32_t gy = 0; gy < k; ++gy)
        for (uint32_t gx = 0; gx < k; ++gx) {
            const uint32_t base = (uint32_t)m->vertices.size();
            for (const auto& v : src->vertices) {
                trc_TriangleVertex e = v;
                e.v[0] += spacing * gx;
                e.v[1] += spacing * gy;
                m->vertices.push_back(e);
            }
            for (uint32_t i : src->indices) m->indices.push_back(base + i);
        }
    *out = m;
    return TRC_OK;
}

void trc_host_mesh_view(const trc_host_mesh* m, const trc_TriangleVertex** vertices, uint32_t* n_vertices,
                        const uint32_t** indices, uint32_t* n_indices) {
    if (vertices) *vertices = m->vertices.data();
    if (n_vertices) *n_vertices = (uint32_t)m->vertices.size();
    if (indices) *indices = m->indices.data();
    if (n_indices) *n_indices = (uint32_t)m->indices.size();
}

void trc_host_mesh_destroy(trc_host_mesh* m) { delete m; }

}  // extern "C"
