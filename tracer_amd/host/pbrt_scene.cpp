// pbrt_scene.cpp -- a whole scene from a pbrt-v3 file: camera, film, area lights, materials, spheres, triangle meshes.
//
// "Support pbrt-v3 file format" is the reference's unchecked to-do (RT_Metal/README.md:57); it vendors minipbrt for it
// (RT_Metal/Tracer/minipbrt.h:1528-1546) and so far reads one medium through it (AAPLRenderer.mm:626-651).  This file
// finishes that intent for what the hot path can render: what the file says is parsed like minipbrt parses it
// (tests/test_pbrt_scene.py compares camera, film, every shape's type / shapeToWorld / radius / material / area light
// field for field with the reference's own minipbrt.cpp compiled in place), then mapped onto the reference's
// primitives with the reference's own constructors (Tracer.mm:127-172):
//
//   Camera "perspective" + LookAt    -> trc_Camera through MakeCamera (Tracer.mm:87-125): eye, look, up as written, fov,
//                                       lensradius -> aperture = 2 lensradius, focaldistance (finite) -> focus_dist
//   Film "image" x/yresolution       -> frame size
//   Shape "sphere"                   -> trc_Sphere (MakeSphere), centre = shapeToWorld * 0, radius * the uniform scale
//   Shape "trianglemesh"             -> an axis-aligned rectangle (4 points, 2 triangles) becomes a trc_Square (MakeSquare):
//                                       that is how pbrt files spell the walls and lights of a Cornell box, and the
//                                       reference samples its lights as squareList[5] / [6] (Render.metal:320-324);
//                                       anything else joins the triangle list (material 19, Triangle.hh:82)
//   Material matte / plastic / metal / glass (+ mirror as metal), MakeNamedMaterial / NamedMaterial
//                                    -> trc_Material Lambert / Plastic / Metal / Glass with albedo Kd / Kd / 1 / Kt
//                                       (the lobes' other parameters are hard-coded in the reference, Material.hh note)
//   AreaLightSource "diffuse" L      -> trc_Material Diffuse (the reference's emitter type) with albedo L
//   Texture "name" .. "checkerboard" -> a material whose Kd names it gets TextureInfo{Checker, albedo = tex1}: the
//                                       reference's only procedural pattern (Texture.hh:17-43: albedo x {1, 0.5} in a fixed
//                                       8 x 4 grid of the surface's uv), so tex2 and u / vscale are reported, not rendered
//   Shape "plymesh"                  -> the triangles of the PLY file (ply_reader.hpp), like a trianglemesh
//   Shape "disk" / "cylinder" / "cone" / "paraboloid" / "hyperboloid" -> tessellated (kQuadricSegments steps of phi) into triangles with the quadric's own
//                                       normals and (phi / phimax, radial | axial) as uv, like a trianglemesh
//
// Squares are ordered so that emitters sit at indices 5 and 6 when the file has any (padding with unreferenced
// degenerate squares, duplicating a lone emitter); `mis_ready` says whether traceMIS / traceVolume may be used.
#include <algorithm>
#include <cmath>
#include <map>
#include <new>
#include <string>

#include "host_scene.hpp"
#include "pbrt_text.hpp"
#include "ply_reader.hpp"

using namespace trc;

namespace {

struct GfxState {                       // what AttributeBegin / End save (pbrt-v3 GraphicsState, the part used here)
    int32_t material = TRC_PBRT_MATTE;  // pbrt's default material is matte, Kd 0.5
    float color[3] = {0.5f, 0.5f, 0.5f};
    int32_t texture = TRC_PBRT_TEX_NONE;   // what the material's colour parameter names: nothing, a checkerboard, another texture
    float tex2[3] = {0, 0, 0};
    bool emitter = false;
    float L[3] = {1, 1, 1};
};
struct NamedTexture { int32_t kind; float tex1[3], tex2[3]; };
constexpr uint32_t kQuadricSegments = 64;     // steps of phi over a full turn
constexpr size_t kMaxInstancedTriangles = size_t(1) << 24;   // ObjectInstance copies stop here (16.7 M triangles: 1.9 GB of device scene)

const PbrtParam* find(const std::vector<PbrtParam>& ps, const char* name) {
    for (const PbrtParam& p : ps) if (p.name == name) return &p;
    return nullptr;
}
void rgb_of(const std::vector<PbrtParam>& ps, const char* name, float out[3]) {
    const PbrtParam* p = find(ps, name);
    if (!p || p->numbers.empty() || p->type == "texture") return;
    if (p->numbers.size() >= 3) for (int k = 0; k < 3; ++k) out[k] = (float)p->numbers[k];
    else out[0] = out[1] = out[2] = (float)p->numbers[0];
}
float float_of(const std::vector<PbrtParam>& ps, const char* name, float dflt) {
    const PbrtParam* p = find(ps, name);
    return (p && !p->numbers.empty()) ? (float)p->numbers[0] : dflt;
}
void set_material_constant(GfxState& g, const std::string& type, const std::vector<PbrtParam>& ps);
void set_material(GfxState& g, const std::string& type, const std::vector<PbrtParam>& ps, const std::map<std::string, std::string>& strs,
                  const std::map<std::string, NamedTexture>& textures) {
    g.texture = TRC_PBRT_TEX_NONE; g.tex2[0] = g.tex2[1] = g.tex2[2] = 0.0f;
    set_material_constant(g, type, ps);
    // "texture Kd" "name" (Kr for mirror, Kt for glass): the colour comes from a named texture
    const char* slot = g.material == TRC_PBRT_MIRROR ? "Kr" : g.material == TRC_PBRT_GLASS ? "Kt" : "Kd";
    const PbrtParam* p = find(ps, slot);
    auto ref = strs.find(slot);
    if (p && p->type == "texture" && ref != strs.end()) {
        auto it = textures.find(ref->second);
        if (it == textures.end()) return;                          // pbrt reports an undefined texture and keeps the default
        g.texture = it->second.kind;
        if (it->second.kind == TRC_PBRT_TEX_CHECKERBOARD) { std::memcpy(g.color, it->second.tex1, sizeof g.color); std::memcpy(g.tex2, it->second.tex2, sizeof g.tex2); }
    }
}
void set_material_constant(GfxState& g, const std::string& type, const std::vector<PbrtParam>& ps) {
    g.color[0] = g.color[1] = g.color[2] = 1.0f;
    if (type == "matte") { g.material = TRC_PBRT_MATTE; g.color[0] = g.color[1] = g.color[2] = 0.5f; rgb_of(ps, "Kd", g.color); }
    else if (type == "plastic") { g.material = TRC_PBRT_PLASTIC; g.color[0] = g.color[1] = g.color[2] = 0.25f; rgb_of(ps, "Kd", g.color); }
    else if (type == "metal") { g.material = TRC_PBRT_METAL; }
    else if (type == "mirror") { g.material = TRC_PBRT_MIRROR; g.color[0] = g.color[1] = g.color[2] = 0.9f; rgb_of(ps, "Kr", g.color); }
    else if (type == "glass") { g.material = TRC_PBRT_GLASS; rgb_of(ps, "Kt", g.color); }
    else { g.material = TRC_PBRT_OTHER; g.color[0] = g.color[1] = g.color[2] = 0.5f; rgb_of(ps, "Kd", g.color); }
}

int32_t material_type_of(int32_t pbrt_material) {
    switch (pbrt_material) {
        case TRC_PBRT_PLASTIC: return TRC_MAT_PLASTIC;
        case TRC_PBRT_METAL: case TRC_PBRT_MIRROR: return TRC_MAT_METAL;
        case TRC_PBRT_GLASS: return TRC_MAT_GLASS;
        default: return TRC_MAT_LAMBERT;
    }
}

// parameter list that also keeps string values ("string type" "matte"): name -> value
std::string dir_of(const char* path) {
    const std::string p(path);
    const size_t slash = p.find_last_of('/');
    return slash == std::string::npos ? std::string() : p.substr(0, slash + 1);
}

bool read_params_with_strings(PbrtLexer& lx, std::vector<PbrtParam>& nums, std::map<std::string, std::string>& strs) {
    nums.clear(); strs.clear();
    while (lx.peek().kind == PbrtToken::String) {
        if (lx.peek().text.find_first_of(" \t") == std::string::npos) break;
        PbrtToken decl = lx.next();
        PbrtParam p;
        const size_t sp = decl.text.find_first_of(" \t");
        p.type = decl.text.substr(0, sp);
        const size_t b = decl.text.find_first_not_of(" \t", sp);
        p.name = b == std::string::npos ? std::string() : decl.text.substr(b);
        PbrtToken v = lx.next();
        if (v.kind == PbrtToken::Open) {
            for (;;) {
                PbrtToken e = lx.next();
                if (e.kind == PbrtToken::Close) break;
                if (e.kind == PbrtToken::End) return false;
                if (e.kind == PbrtToken::Number) p.numbers.push_back(e.value);
                else if (e.kind == PbrtToken::String) strs[p.name] = e.text;
            }
        } else if (v.kind == PbrtToken::Number) p.numbers.push_back(v.value);
        else if (v.kind == PbrtToken::String) strs[p.name] = v.text;
        else if (v.kind != PbrtToken::Word) return false;
        nums.push_back(std::move(p));
    }
    return true;
}

struct Quad { int axis_k; float k, i0, i1, j0, j1; };

// four points + two triangles forming an axis-aligned rectangle (in world space)?
bool as_axis_aligned_rectangle(const std::vector<trc_float3>& P, const std::vector<uint32_t>& idx, Quad& q) {
    if (P.size() != 4 || idx.size() != 6) return false;
    for (int ak = 0; ak < 3; ++ak) {
        const float k = get(P[0], ak);
        if (!(get(P[1], ak) == k && get(P[2], ak) == k && get(P[3], ak) == k)) continue;
        const int ai = ak == 0 ? 1 : 0, aj = ak == 2 ? 1 : 2;       // the reference's (i, j) pairs: (1,2) (0,2) (0,1)
        float lo_i = FLT_MAX, hi_i = -FLT_MAX, lo_j = FLT_MAX, hi_j = -FLT_MAX;
        for (const trc_float3& p : P) {
            lo_i = std::min(lo_i, get(p, ai)); hi_i = std::max(hi_i, get(p, ai));
            lo_j = std::min(lo_j, get(p, aj)); hi_j = std::max(hi_j, get(p, aj));
        }
        if (!(lo_i < hi_i && lo_j < hi_j)) return false;
        int corners = 0;                                           // every point is a distinct corner of that rectangle
        for (const trc_float3& p : P) {
            const bool ci = get(p, ai) == lo_i || get(p, ai) == hi_i, cj = get(p, aj) == lo_j || get(p, aj) == hi_j;
            if (!ci || !cj) return false;
            corners |= 1 << ((get(p, ai) == hi_i ? 1 : 0) | (get(p, aj) == hi_j ? 2 : 0));
        }
        if (corners != 15) return false;
        // the two triangles must cover it: each uses three distinct corners and together they use all four
        uint32_t used = 0;
        for (int t = 0; t < 2; ++t) {
            const uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
            if (a == b || b == c || a == c) return false;
            used |= (1u << a) | (1u << b) | (1u << c);
        }
        if (used != 15u) return false;
        q.axis_k = ak; q.k = k; q.i0 = lo_i; q.i1 = hi_i; q.j0 = lo_j; q.j1 = hi_j;
        return true;
    }
    return false;
}

}  // namespace

extern "C" trc_status trc_host_scene_load_pbrt(const char* path, trc_host_scene** out_scene, trc_Camera* out_camera,
                                              trc_pbrt_info* info, trc_pbrt_shape* shapes, uint32_t capacity) {
    if (!path || !out_scene) return TRC_ERR_INVALID_ARG;
    *out_scene = nullptr;
    std::string text;
    if (!read_pbrt_text(path, 0, text)) return TRC_ERR_INVALID_ARG;
    trc_host_scene* s = new (std::nothrow) trc_host_scene();
    if (!s) return TRC_ERR_OOM;
    trc_pbrt_info inf;
    std::memset(&inf, 0, sizeof inf);
    inf.fov = 90.0f; inf.xres = 640; inf.yres = 480; inf.focaldistance = 1e30f;
    { const M4 id = m4_identity(); std::memcpy(inf.camera_to_world, id.m, sizeof id.m); }
    float eye[3] = {0, 0, 0}, look[3] = {0, 0, 1}, up[3] = {0, 1, 0};
    bool have_lookat = false;

    PbrtLexer lx(text);
    M4 ctm = m4_identity();
    std::vector<M4> tstack;
    std::vector<GfxState> gstack;
    GfxState g;
    std::map<std::string, M4> named;
    std::map<std::string, GfxState> named_materials;
    int object_depth = 0;
    // object instancing (pbrt-v3 api.cpp pbrtObjectBegin / pbrtObjectInstance): the body of an ObjectBegin .. ObjectEnd block is
    // a template; every ObjectInstance places a copy of its shapes, world = InstanceToWorld (the CTM at the ObjectInstance) x
    // the shape's own ObjectToWorld (the CTM at its Shape directive).  The instances are expanded after the world's own
    // shapes, in file order, by reading the template's body again with that prefix -- flat triangles / spheres, since the
    // reference's scene arrays have no instance level (Render.hh:135-252 walks ONE tree of primitives).
    struct ObjectDef { size_t body; M4 ctm; GfxState g; };
    struct InstanceRef { std::string name; M4 to_world; };
    std::map<std::string, ObjectDef> objects;
    std::vector<InstanceRef> instances;
    size_t next_instance = 0;
    bool replaying = false;
    M4 inst_prefix = m4_identity();
    std::vector<float> a;
    std::vector<PbrtParam> params;
    std::map<std::string, std::string> strs;
    std::vector<trc_pbrt_shape> descs;

    struct PendingSquare { Quad q; uint32_t material; bool emitter; size_t desc; };
    std::vector<PendingSquare> squares;
    std::map<std::string, uint32_t> material_index;                 // one trc_Material per distinct (type, colour)
    std::map<std::string, NamedTexture> named_textures;
    auto intern_material = [&](int32_t type, const float c[3], bool checker = false) -> uint32_t {
        char key[112];
        std::snprintf(key, sizeof key, "%d/%a/%a/%a/%d", type, c[0], c[1], c[2], checker ? 1 : 0);
        auto it = material_index.find(key);
        if (it != material_index.end()) return it->second;
        uint32_t idx = (uint32_t)s->materials.size();
        if (idx == 19) { s->materials.push_back(make_material(TRC_MAT_LAMBERT)); idx = 20; }   // 19 is the triangles' slot
        trc_Material m = make_material(type);
        m.textureInfo.albedo = f3(c[0], c[1], c[2]);
        if (checker) m.textureInfo.type = TRC_TEX_CHECKER;           // TextureInfo::value, Texture.hh:24-28
        if (type == TRC_MAT_METAL || type == TRC_MAT_GLASS) m.specular = 1;
        if (type == TRC_MAT_GLASS) m.eta = 1.5f;
        s->materials.push_back(m);
        material_index[key] = idx;
        return idx;
    };
    bool have_tri_material = false;
    trc_Material tri_material = make_material(TRC_MAT_LAMBERT);
    tri_material.textureInfo.albedo = f3(0.5f);

    auto fail = [&](trc_status st = TRC_ERR_INVALID_ARG) { delete s; return st; };
    for (;;) {
        PbrtToken t = lx.next();
        if (t.kind == PbrtToken::End) {
            // the file is read: now the instances, one template body each
            bool again = false;
            while (!replaying && next_instance < instances.size()) {
                const InstanceRef& in = instances[next_instance++];
                auto it = objects.find(in.name);
                if (it == objects.end()) { inf.n_unsupported_shapes++; continue; }       // an instance of nothing
                // every instance is a flat copy of its template (the reference's scene has no instancing either): bounded, so that
                // a file with 10^5 instances of a mesh ends as "unsupported" and not as an out-of-memory kill
                if (s->indices.size() / 3 >= kMaxInstancedTriangles) { inf.n_unsupported_shapes++; continue; }
                lx.i = it->second.body; ctm = it->second.ctm; g = it->second.g;
                tstack.clear(); gstack.clear(); object_depth = 0;
                inst_prefix = in.to_world; replaying = true; again = true;
                break;
            }
            if (again) continue;
            break;
        }
        if (t.kind != PbrtToken::Word) continue;
        const std::string& d = t.text;
        if (replaying && d == "ObjectEnd") { replaying = false; lx.i = text.size(); continue; }      // this copy is placed
        if (replaying && (d == "ObjectBegin" || d == "ObjectInstance" || d == "WorldEnd")) return fail();
        if (d == "Identity") ctm = m4_identity();
        else if (d == "Translate") {
            if (!read_numbers(lx, 3, a)) return fail();
            M4 m = m4_identity(); m.m[0][3] = a[0]; m.m[1][3] = a[1]; m.m[2][3] = a[2];
            ctm = m4_mul(ctm, m);
        } else if (d == "Scale") {
            if (!read_numbers(lx, 3, a)) return fail();
            M4 m = m4_identity(); m.m[0][0] = a[0]; m.m[1][1] = a[1]; m.m[2][2] = a[2];
            ctm = m4_mul(ctm, m);
        } else if (d == "Rotate") {
            if (!read_numbers(lx, 4, a)) return fail();
            ctm = m4_mul(ctm, m4_rotate(a[0], a[1], a[2], a[3]));
        } else if (d == "LookAt") {
            if (!read_numbers(lx, 9, a)) return fail();
            M4 w2c;
            if (!m4_look_at(a.data(), w2c)) return fail();
            ctm = m4_mul(ctm, w2c);
            for (int k = 0; k < 3; ++k) { eye[k] = a[k]; look[k] = a[3 + k]; up[k] = a[6 + k]; }
            have_lookat = true;
        } else if (d == "Transform" || d == "ConcatTransform") {
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
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            named["camera"] = ctm;
            M4 c2w;
            if (!m4_inverse(ctm, c2w)) return fail();
            std::memcpy(inf.camera_to_world, c2w.m, sizeof c2w.m);
            inf.fov = float_of(params, "fov", 90.0f);
            inf.lensradius = float_of(params, "lensradius", 0.0f);
            inf.focaldistance = float_of(params, "focaldistance", 1e30f);
            inf.perspective = kind.text == "perspective";
        } else if (d == "Film") {
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            auto res = [](float v) { return !(v >= 1.0f) ? 1u : (v > 524280.0f ? 524280u : (uint32_t)v); };   // what trc_resize accepts
            inf.xres = res(float_of(params, "xresolution", 640.0f));
            inf.yres = res(float_of(params, "yresolution", 480.0f));
        } else if (d == "WorldBegin") { ctm = m4_identity(); named["world"] = ctm; }
        else if (d == "AttributeBegin") { tstack.push_back(ctm); gstack.push_back(g); }
        else if (d == "AttributeEnd") {
            if (tstack.empty() || gstack.empty()) return fail();
            ctm = tstack.back(); tstack.pop_back(); g = gstack.back(); gstack.pop_back();
        } else if (d == "TransformBegin") tstack.push_back(ctm);
        else if (d == "TransformEnd") { if (tstack.empty()) return fail(); ctm = tstack.back(); tstack.pop_back(); }
        else if (d == "ObjectBegin") {
            const PbrtToken name = lx.next();
            if (name.kind != PbrtToken::String || object_depth > 0) return fail();
            ++object_depth; tstack.push_back(ctm); gstack.push_back(g);
            objects[name.text] = ObjectDef{lx.i, ctm, g};
        } else if (d == "ObjectInstance") {
            const PbrtToken name = lx.next();
            if (name.kind != PbrtToken::String) return fail();
            if (object_depth == 0) { instances.push_back(InstanceRef{name.text, ctm}); inf.n_instances++; }
        }
        else if (d == "ObjectEnd") {
            if (object_depth == 0 || tstack.empty() || gstack.empty()) return fail();
            --object_depth; ctm = tstack.back(); tstack.pop_back(); g = gstack.back(); gstack.pop_back();
        } else if (d == "Material") {
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            const bool em = g.emitter; float L[3] = {g.L[0], g.L[1], g.L[2]};
            set_material(g, kind.text, params, strs, named_textures);
            g.emitter = em; std::memcpy(g.L, L, sizeof L);
        } else if (d == "MakeNamedMaterial") {
            PbrtToken name = lx.next();
            if (name.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            GfxState m;
            set_material(m, strs.count("type") ? strs["type"] : std::string("matte"), params, strs, named_textures);
            named_materials[name.text] = m;
        } else if (d == "NamedMaterial") {
            PbrtToken name = lx.next();
            if (name.kind != PbrtToken::String) return fail();
            auto it = named_materials.find(name.text);
            if (it != named_materials.end()) {
                g.material = it->second.material; std::memcpy(g.color, it->second.color, sizeof g.color);
                g.texture = it->second.texture; std::memcpy(g.tex2, it->second.tex2, sizeof g.tex2);
            }
        } else if (d == "Texture") {                                   // Texture "name" "spectrum|float" "class" params
            PbrtToken name = lx.next(), data = lx.next(), cls = lx.next();
            if (name.kind != PbrtToken::String || data.kind != PbrtToken::String || cls.kind != PbrtToken::String ||
                !read_params_with_strings(lx, params, strs)) return fail();
            NamedTexture nt;
            nt.kind = TRC_PBRT_TEX_OTHER;
            nt.tex1[0] = nt.tex1[1] = nt.tex1[2] = 1.0f; nt.tex2[0] = nt.tex2[1] = nt.tex2[2] = 0.0f;    // pbrt's defaults
            if (cls.text == "checkerboard" && float_of(params, "dimension", 2.0f) == 2.0f) {
                nt.kind = TRC_PBRT_TEX_CHECKERBOARD;
                rgb_of(params, "tex1", nt.tex1); rgb_of(params, "tex2", nt.tex2);
            }
            named_textures[name.text] = nt;
        } else if (d == "AreaLightSource") {
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            g.emitter = true;
            g.L[0] = g.L[1] = g.L[2] = 1.0f;
            rgb_of(params, "L", g.L);
        } else if (d == "Shape") {
            PbrtToken kind = lx.next();
            if (kind.kind != PbrtToken::String || !read_params_with_strings(lx, params, strs)) return fail();
            if (object_depth > 0) continue;                         // templates of object instancing: not part of the world
            const M4 world = replaying ? m4_mul(inst_prefix, ctm) : ctm;      // a template's copy: InstanceToWorld x ObjectToWorld
            trc_pbrt_shape ds;
            std::memset(&ds, 0, sizeof ds);
            ds.kind = -1; ds.mapped_type = -1;
            std::memcpy(ds.shape_to_world, world.m, sizeof world.m);
            ds.material = g.material; std::memcpy(ds.color, g.color, sizeof ds.color);
            ds.emitter = g.emitter ? 1 : 0; std::memcpy(ds.L, g.L, sizeof ds.L);
            ds.texture = g.texture; std::memcpy(ds.tex2, g.tex2, sizeof ds.tex2);
            const bool checker = !g.emitter && g.texture == TRC_PBRT_TEX_CHECKERBOARD;
            if (g.texture == TRC_PBRT_TEX_OTHER && !g.emitter) inf.n_unsupported_textures++;
            const int32_t mtype = g.emitter ? (int32_t)TRC_MAT_DIFFUSE : material_type_of(g.material);
            const float* mcolor = g.emitter ? g.L : g.color;
            if (g.material == TRC_PBRT_OTHER && !g.emitter) inf.n_unsupported_materials++;
            if (kind.text == "sphere") {
                ds.kind = TRC_PRIM_SPHERE;
                ds.radius = float_of(params, "radius", 1.0f);
                // centre = shapeToWorld * origin; the matrix must be a similarity (uniform scale) for a sphere to stay one
                const trc_float3 c = f3(world.m[0][3], world.m[1][3], world.m[2][3]);
                const float sx = length(f3(world.m[0][0], world.m[1][0], world.m[2][0]));
                const float sy = length(f3(world.m[0][1], world.m[1][1], world.m[2][1]));
                const float sz = length(f3(world.m[0][2], world.m[1][2], world.m[2][2]));
                if (std::fabs(sx - sy) > 1e-4f * sx || std::fabs(sx - sz) > 1e-4f * sx) { inf.n_unsupported_shapes++; descs.push_back(ds); continue; }
                ds.mapped_type = TRC_PRIM_SPHERE; ds.mapped_index = (uint32_t)s->spheres.size();
                ds.mapped_material = intern_material(mtype, mcolor, checker);
                s->spheres.push_back(make_sphere(ds.radius * sx, c, ds.mapped_material));
            } else if (kind.text == "trianglemesh" || kind.text == "plymesh" || kind.text == "disk" || kind.text == "cylinder" ||
                       kind.text == "cone" || kind.text == "paraboloid" || kind.text == "hyperboloid") {
                // object-space vertices (+ normals, uv) and triangle indices of the shape, then one path for all four
                std::vector<float> Po, No, UVo;
                std::vector<uint32_t> idx;
                if (kind.text == "trianglemesh") {
                    ds.kind = TRC_PBRT_SHAPE_TRIANGLEMESH;
                    const PbrtParam *P = find(params, "P"), *N = find(params, "N"), *I = find(params, "indices");
                    const PbrtParam* UV = find(params, "uv"); if (!UV) UV = find(params, "st");
                    if (!P || P->numbers.size() % 3 != 0 || P->numbers.empty()) return fail();
                    const size_t nv = P->numbers.size() / 3;
                    std::vector<double> seq;
                    if (!I) { if (nv != 3) return fail(); seq = {0, 1, 2}; }
                    const std::vector<double>& idxd = I ? I->numbers : seq;
                    if (idxd.size() % 3 != 0) return fail();
                    if (N && N->numbers.size() != nv * 3) N = nullptr;
                    if (UV && UV->numbers.size() != nv * 2) UV = nullptr;
                    for (double v : P->numbers) Po.push_back((float)v);
                    if (N) for (double v : N->numbers) No.push_back((float)v);
                    if (UV) for (double v : UV->numbers) UVo.push_back((float)v);
                    for (double k : idxd) { if (!(k >= 0 && k < (double)nv)) return fail(); idx.push_back((uint32_t)k); }
                } else if (kind.text == "plymesh") {
                    ds.kind = TRC_PBRT_SHAPE_PLYMESH;
                    auto fn = strs.find("filename");
                    if (fn == strs.end()) return fail();
                    const std::string ply_path = (!fn->second.empty() && fn->second[0] == '/') ? fn->second : dir_of(path) + fn->second;
                    PlyMesh pm;
                    if (!read_ply(ply_path, pm)) return fail();
                    Po.swap(pm.P); No.swap(pm.N); UVo.swap(pm.UV); idx.swap(pm.indices);
                } else {
                    // The quadrics of pbrt-v3 (shapes/disk.cpp, cylinder.cpp, cone.cpp, paraboloid.cpp, hyperboloid.cpp), each a
                    // surface P(u, v) with phi = u * phimax: tessellated into a grid of kQuadricSegments steps of phi over a full
                    // turn x `rows` steps of v (1 where P is linear in v), two triangles per cell, with the surface's own normals
                    //   disk        z = height, r from innerradius (v = 0) to radius (v = 1)
                    //   cylinder    r = radius, z from zmin to zmax
                    //   cone        r = radius (1 - v), z = v height                      (apex cells: one triangle)
                    //   paraboloid  z from zmin to zmax, r = radius sqrt(z / zmax)
                    //   hyperboloid the segment p1 -> p2 swept about z: p = (1 - v) p1 + v p2 rotated by phi
                    enum { QDisk, QCylinder, QCone, QParaboloid, QHyperboloid } q =
                        kind.text == "disk" ? QDisk : kind.text == "cylinder" ? QCylinder : kind.text == "cone" ? QCone :
                        kind.text == "paraboloid" ? QParaboloid : QHyperboloid;
                    ds.kind = q == QDisk ? TRC_PBRT_SHAPE_DISK : q == QCylinder ? TRC_PBRT_SHAPE_CYLINDER : q == QCone ? TRC_PBRT_SHAPE_CONE :
                              q == QParaboloid ? TRC_PBRT_SHAPE_PARABOLOID : TRC_PBRT_SHAPE_HYPERBOLOID;
                    ds.radius = float_of(params, "radius", 1.0f);
                    ds.phimax = std::min(360.0f, std::max(0.0f, float_of(params, "phimax", 360.0f)));
                    float p1[3] = {0, 0, 0}, p2[3] = {1, 1, 1};
                    if (q == QDisk) { ds.zmin = ds.zmax = float_of(params, "height", 0.0f); ds.innerradius = float_of(params, "innerradius", 0.0f); }
                    else if (q == QCone) { ds.zmin = 0.0f; ds.zmax = float_of(params, "height", 1.0f); }
                    else if (q == QHyperboloid) {
                        if (const PbrtParam* a1 = find(params, "p1")) if (a1->numbers.size() == 3) for (int k = 0; k < 3; ++k) p1[k] = (float)a1->numbers[k];
                        if (const PbrtParam* a2 = find(params, "p2")) if (a2->numbers.size() == 3) for (int k = 0; k < 3; ++k) p2[k] = (float)a2->numbers[k];
                        std::memcpy(ds.p1, p1, sizeof p1); std::memcpy(ds.p2, p2, sizeof p2);
                        ds.radius = 0.0f; ds.zmin = std::min(p1[2], p2[2]); ds.zmax = std::max(p1[2], p2[2]);
                    } else {
                        const float z0 = float_of(params, "zmin", q == QParaboloid ? 0.0f : -1.0f), z1 = float_of(params, "zmax", 1.0f);
                        ds.zmin = std::min(z0, z1); ds.zmax = std::max(z0, z1);
                    }
                    const bool bad = !(ds.phimax > 0.0f) ||
                        (q == QDisk && (!(ds.radius > 0.0f) || !(ds.innerradius >= 0.0f && ds.innerradius < ds.radius))) ||
                        (q == QCylinder && (!(ds.radius > 0.0f) || !(ds.zmin < ds.zmax))) ||
                        (q == QCone && (!(ds.radius > 0.0f) || !(ds.zmax > 0.0f))) ||
                        (q == QParaboloid && (!(ds.radius > 0.0f) || !(ds.zmin >= 0.0f) || !(ds.zmin < ds.zmax))) ||
                        (q == QHyperboloid && p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]);
                    if (bad) { inf.n_unsupported_shapes++; descs.push_back(ds); continue; }
                    const uint32_t seg = std::max(3u, (uint32_t)std::ceil(kQuadricSegments * ds.phimax / 360.0f));
                    const uint32_t rows = q == QParaboloid ? 16u : q == QHyperboloid ? 8u : 1u;
                    const float phimax = ds.phimax * 3.14159265358979323846f / 180.0f;
                    for (uint32_t k = 0; k <= seg; ++k) {
                        const float u = (float)k / (float)seg, phi = u * phimax, c = std::cos(phi), sn = std::sin(phi);
                        for (uint32_t r = 0; r <= rows; ++r) {
                            const float v = (float)r / (float)rows;
                            float P[3], N[3];
                            if (q == QDisk) {
                                const float rr = ds.innerradius + v * (ds.radius - ds.innerradius);
                                P[0] = rr * c; P[1] = rr * sn; P[2] = ds.zmin; N[0] = 0; N[1] = 0; N[2] = 1;
                            } else if (q == QCylinder) {
                                P[0] = ds.radius * c; P[1] = ds.radius * sn; P[2] = r ? ds.zmax : ds.zmin; N[0] = c; N[1] = sn; N[2] = 0;
                            } else if (q == QCone) {
                                const float rr = ds.radius * (1.0f - v);
                                P[0] = rr * c; P[1] = rr * sn; P[2] = v * ds.zmax;
                                const float len = std::sqrt(ds.zmax * ds.zmax + ds.radius * ds.radius);
                                N[0] = ds.zmax * c / len; N[1] = ds.zmax * sn / len; N[2] = ds.radius / len;
                            } else if (q == QParaboloid) {
                                const float z = ds.zmin + v * (ds.zmax - ds.zmin), rr = ds.radius * std::sqrt(z / ds.zmax);
                                P[0] = rr * c; P[1] = rr * sn; P[2] = z;
                                // gradient of zmax (x^2 + y^2) / radius^2 - z: outward, away from the axis and downwards
                                const float gx = 2.0f * ds.zmax * P[0] / (ds.radius * ds.radius), gy = 2.0f * ds.zmax * P[1] / (ds.radius * ds.radius);
                                const float len = std::sqrt(gx * gx + gy * gy + 1.0f);
                                N[0] = gx / len; N[1] = gy / len; N[2] = -1.0f / len;
                            } else {
                                const float px = (1 - v) * p1[0] + v * p2[0], py = (1 - v) * p1[1] + v * p2[1], pz = (1 - v) * p1[2] + v * p2[2];
                                P[0] = px * c - py * sn; P[1] = px * sn + py * c; P[2] = pz;
                                const float dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
                                const float du[3] = {-P[1], P[0], 0.0f}, dv[3] = {dx * c - dy * sn, dx * sn + dy * c, dz};      // dP/dphi, dP/dv
                                float n[3] = {du[1] * dv[2] - du[2] * dv[1], du[2] * dv[0] - du[0] * dv[2], du[0] * dv[1] - du[1] * dv[0]};
                                const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                                if (len > 0.0f) { N[0] = n[0] / len; N[1] = n[1] / len; N[2] = n[2] / len; } else { N[0] = 0; N[1] = 0; N[2] = 1; }
                            }
                            for (int a3 = 0; a3 < 3; ++a3) { Po.push_back(P[a3]); No.push_back(N[a3]); }
                            UVo.push_back(u); UVo.push_back(v);
                        }
                    }
                    for (uint32_t k = 0; k < seg; ++k)
                        for (uint32_t r = 0; r < rows; ++r) {
                            const uint32_t a = k * (rows + 1) + r, b = a + 1, c2 = a + rows + 1, d2 = c2 + 1;
                            const bool apex = q == QCone && r + 1 == rows;                 // b and d2 are the apex: one triangle
                            // a paraboloid that starts at z = 0 has its apex in row 0: a and c2 are that one point
                            const bool apex0 = q == QParaboloid && r == 0 && ds.zmin == 0.0f;
                            if (!apex) { idx.push_back(a); idx.push_back(b); idx.push_back(d2); }
                            if (!apex0) { idx.push_back(a); idx.push_back(d2); idx.push_back(c2); }
                        }
                }
                const size_t nv = Po.size() / 3;
                if (nv == 0 || idx.empty() || idx.size() % 3 != 0) return fail();
                const bool hasN = No.size() == nv * 3, hasUV = UVo.size() == nv * 2;
                ds.n_vertices = (uint32_t)nv; ds.n_indices = (uint32_t)idx.size();
                std::vector<trc_float3> Pw(nv);
                for (size_t v = 0; v < nv; ++v) {
                    const float x = Po[3 * v], y = Po[3 * v + 1], z = Po[3 * v + 2];
                    float q[4];
                    for (int r = 0; r < 4; ++r) q[r] = world.m[r][0] * x + world.m[r][1] * y + world.m[r][2] * z + world.m[r][3];
                    const float w = q[3];
                    Pw[v] = f3(w == 1 ? q[0] : q[0] / w, w == 1 ? q[1] : q[1] / w, w == 1 ? q[2] : q[2] / w);
                }
                Quad q;
                if (kind.text == "trianglemesh" && as_axis_aligned_rectangle(Pw, idx, q)) {
                    ds.mapped_type = TRC_PRIM_SQUARE;
                    ds.mapped_material = intern_material(mtype, mcolor, checker);
                    squares.push_back(PendingSquare{q, ds.mapped_material, g.emitter, descs.size()});
                } else {
                    ds.mapped_type = TRC_PRIM_TRIANGLE; ds.mapped_index = (uint32_t)(s->indices.size() / 3); ds.mapped_material = 19;
                    if (!have_tri_material) {
                        tri_material = make_material(mtype);
                        tri_material.textureInfo.albedo = f3(mcolor[0], mcolor[1], mcolor[2]);
                        if (checker) tri_material.textureInfo.type = TRC_TEX_CHECKER;
                        if (mtype == TRC_MAT_METAL || mtype == TRC_MAT_GLASS) tri_material.specular = 1;
                        have_tri_material = true;
                    } else if (tri_material.type != mtype) inf.n_triangle_material_conflicts++;
                    M4 inv;
                    const bool has_inv = m4_inverse(world, inv);
                    const uint32_t base = (uint32_t)s->vertices.size();
                    std::vector<trc_float3> acc(nv, f3(0.0f));
                    if (!hasN || !has_inv)                                       // area-weighted smooth normals
                        for (size_t tt = 0; tt + 2 < idx.size(); tt += 3) {
                            const trc_float3 fn = cross(Pw[idx[tt + 1]] - Pw[idx[tt]], Pw[idx[tt + 2]] - Pw[idx[tt]]);
                            acc[idx[tt]] = acc[idx[tt]] + fn; acc[idx[tt + 1]] = acc[idx[tt + 1]] + fn; acc[idx[tt + 2]] = acc[idx[tt + 2]] + fn;
                        }
                    for (size_t v = 0; v < nv; ++v) {
                        trc_TriangleVertex tv;
                        tv.v[0] = Pw[v].x; tv.v[1] = Pw[v].y; tv.v[2] = Pw[v].z;
                        if (hasN && has_inv) {
                            const float nx = No[3 * v], ny = No[3 * v + 1], nz = No[3 * v + 2];
                            for (int r = 0; r < 3; ++r) tv.n[r] = inv.m[0][r] * nx + inv.m[1][r] * ny + inv.m[2][r] * nz;
                        } else {
                            const float len = length(acc[v]);
                            const trc_float3 n = len > 0.0f ? acc[v] / len : f3(0, 1, 0);
                            tv.n[0] = n.x; tv.n[1] = n.y; tv.n[2] = n.z;
                        }
                        tv.uv[0] = hasUV ? UVo[2 * v] : 0.0f; tv.uv[1] = hasUV ? UVo[2 * v + 1] : 0.0f;
                        s->vertices.push_back(tv);
                    }
                    for (uint32_t k : idx) s->indices.push_back(base + k);
                }
            } else {
                inf.n_unsupported_shapes++;
            }
            descs.push_back(ds);
        } else {
            for (;;) {                                              // any other directive: skip its arguments and parameters
                const PbrtToken nx = lx.peek();
                if (nx.kind == PbrtToken::Number || (nx.kind == PbrtToken::String && nx.text.find_first_of(" \t") == std::string::npos)) lx.next();
                else break;
            }
            if (!read_params_with_strings(lx, params, strs)) return fail();
        }
    }

    // materials: index 19 is what every triangle uses (Triangle.hh:82)
    while (s->materials.size() < 19) s->materials.push_back(make_material(TRC_MAT_LAMBERT));
    if (s->materials.size() == 19) s->materials.push_back(tri_material); else s->materials[19] = tri_material;

    // squares: non-emitters first, emitters from index 5 on (squareList[5] / [6] are THE lights of traceMIS)
    std::vector<const PendingSquare*> plain, lights;
    for (const PendingSquare& q : squares) (q.emitter ? lights : plain).push_back(&q);
    std::vector<bool> in_bvh;
    auto add_square = [&](const PendingSquare* q, bool leaf) {
        const int ak = q->q.axis_k, ai = ak == 0 ? 1 : 0, aj = ak == 2 ? 1 : 2;
        s->squares.push_back(make_square((uint8_t)ai, q->q.i0, q->q.i1, (uint8_t)aj, q->q.j0, q->q.j1, (uint8_t)ak, q->q.k, q->material));
        in_bvh.push_back(leaf);
        if (leaf) { descs[q->desc].mapped_index = (uint32_t)s->squares.size() - 1; }
    };
    size_t pi = 0;
    if (!lights.empty()) {
        for (; pi < plain.size() && s->squares.size() < 5; ++pi) add_square(plain[pi], true);
        // padding squares: never inserted in the BVH, never sampled (only indices 5 and 6 are)
        while (s->squares.size() < 5) { s->squares.push_back(make_square(0, 0, 1, 2, 0, 1, 1, -3.0e30f, 0)); in_bvh.push_back(false); }
        add_square(lights[0], true);
        if (lights.size() > 1) add_square(lights[1], true);
        else { s->squares.push_back(s->squares.back()); in_bvh.push_back(false); }             // the lone light serves both slots
        for (size_t k = 2; k < lights.size(); ++k) add_square(lights[k], true);
        inf.mis_ready = 1;
    }
    for (; pi < plain.size(); ++pi) add_square(plain[pi], true);

    // leaves in the reference's order: cubes (none here), squares, spheres, triangles (AAPLRenderer.mm:454-468,546-603)
    std::vector<trc_BVH> leaves;
    trc_BVH leaf;
    for (uint32_t i = 0; i < s->squares.size(); ++i)
        if (in_bvh[i]) { trc_host_build_node(&s->squares[i].boundingBOX, &s->squares[i].model_matrix, TRC_PRIM_SQUARE, i, &leaf); leaves.push_back(leaf); }
    for (uint32_t i = 0; i < s->spheres.size(); ++i) {
        trc_host_build_node(&s->spheres[i].boundingBOX, &s->spheres[i].model_matrix, TRC_PRIM_SPHERE, i, &leaf); leaves.push_back(leaf);
    }
    const trc_float4x4 ident = identity4x4();
    for (uint32_t tix = 0; tix < s->indices.size() / 3; ++tix) {
        const trc_TriangleVertex& A = s->vertices[s->indices[3 * tix]];
        const trc_TriangleVertex& B = s->vertices[s->indices[3 * tix + 1]];
        const trc_TriangleVertex& C = s->vertices[s->indices[3 * tix + 2]];
        trc_AABB box;
        box.maxi = f3(std::max({A.v[0], B.v[0], C.v[0]}), std::max({A.v[1], B.v[1], C.v[1]}), std::max({A.v[2], B.v[2], C.v[2]}));
        box.mini = f3(std::min({A.v[0], B.v[0], C.v[0]}), std::min({A.v[1], B.v[1], C.v[1]}), std::min({A.v[2], B.v[2], C.v[2]}));
        trc_host_build_node(&box, &ident, TRC_PRIM_TRIANGLE, tix, &leaf);
        leaves.push_back(leaf);
    }
    if (leaves.size() < 2) return fail(TRC_ERR_UNSUPPORTED);       // a tree needs two leaves (trc_upload_scene)
    s->bvh.resize(2 * leaves.size() - 1);
    std::copy(leaves.begin(), leaves.end(), s->bvh.begin());
    uint32_t n_nodes = 0;
    trc_status st = trc_host_build_tree(s->bvh.data(), (uint32_t)leaves.size(), &n_nodes);
    if (st != TRC_OK) return fail(st);

    inf.n_shapes = (uint32_t)descs.size();
    if (out_camera) {
        if (!have_lookat) {                                         // camera frame from cameraToWorld: origin, +z, +y
            for (int k = 0; k < 3; ++k) { eye[k] = inf.camera_to_world[4 * k + 3]; look[k] = eye[k] + inf.camera_to_world[4 * k + 2]; up[k] = inf.camera_to_world[4 * k + 1]; }
        }
        const float aspect = (float)inf.xres / (float)inf.yres;
        // pbrt's fov spans the SHORTER image axis; MakeCamera takes the vertical one
        float vfov = inf.fov * 3.14159265358979323846f / 180.0f;
        if (aspect < 1.0f) vfov = 2.0f * std::atan(std::tan(vfov * 0.5f) / aspect);
        const trc_float3 d = f3(look[0] - eye[0], look[1] - eye[1], look[2] - eye[2]);
        const float focus = inf.focaldistance < 1e29f ? inf.focaldistance : length(d);
        trc_host_make_camera(out_camera, eye, look, up, 2.0f * inf.lensradius, aspect, vfov, focus);
    }
    if (info) *info = inf;
    if (shapes) for (uint32_t k = 0; k < capacity && k < descs.size(); ++k) shapes[k] = descs[k];
    *out_scene = s;
    return TRC_OK;
}
