// pbrt_text.hpp -- reading pbrt-v3 scene text on the host (shared by the density and the triangle-mesh readers).
#pragma once

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace trc {

// text of a pbrt file with `#` comments removed and `Include "file"` statements (relative to the including file,
// as pbrt-v3 and minipbrt resolve them) spliced in; the reference's cloud/cloud.pbrt pulls its medium in that way
inline bool read_pbrt_text(const std::string& path, int depth, std::string& text) {
    if (depth > 8) return false;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string raw;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) raw.append(chunk, got);
    std::fclose(f);
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    size_t i = 0;
    bool in_string = false;
    while (i < raw.size()) {
        const char c = raw[i];
        if (c == '"') in_string = !in_string;
        if (!in_string && c == '#') {                               // comment to the end of the line
            while (i < raw.size() && raw[i] != '\n') ++i;
            continue;
        }
        const bool at_word = !in_string && raw.compare(i, 7, "Include") == 0 &&
                             (i == 0 || std::isspace((unsigned char)raw[i - 1])) &&
                             (i + 7 < raw.size() && (std::isspace((unsigned char)raw[i + 7]) || raw[i + 7] == '"'));
        if (at_word) {
            size_t q0 = raw.find('"', i + 7);
            size_t q1 = q0 == std::string::npos ? q0 : raw.find('"', q0 + 1);
            if (q1 == std::string::npos) return false;
            const std::string name = raw.substr(q0 + 1, q1 - q0 - 1);
            const std::string child = (!name.empty() && name[0] == '/') ? name : dir + name;
            if (!read_pbrt_text(child, depth + 1, text)) return false;
            text.push_back('\n');
            i = q1 + 1;
            continue;
        }
        text.push_back(c);
        ++i;
    }
    return true;
}

// ---- tokens, parameter lists and 4x4 matrices of the pbrt-v3 text format (shared by mesh.cpp and pbrt_scene.cpp)
struct M4 { float m[4][4]; };     // row-major, p' = M * p, like pbrt's Matrix4x4
inline M4 m4_identity() { M4 r; std::memset(&r, 0, sizeof r); for (int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline bool m4_inverse(const M4& a, M4& out) {                      // Gauss-Jordan with partial pivoting, in double
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i][j]; w[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (w[p][c] == 0.0) return false;
        for (int j = 0; j < 8; ++j) std::swap(w[c][j], w[p][j]);
        const double inv = 1.0 / w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = w[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[i][j] = (float)w[i][4 + j];
    return true;
}

// pbrt-v3 Rotate(theta degrees, axis) and LookAt (the world-to-camera matrix), api.cpp / transform.cpp
inline M4 m4_rotate(float degrees, float x, float y, float z) {
    const float len = std::sqrt(x * x + y * y + z * z);
    const float ax = x / len, ay = y / len, az = z / len;
    const float th = degrees * 3.14159265358979323846f / 180.0f, sn = std::sin(th), cs = std::cos(th);
    M4 m = m4_identity();
    m.m[0][0] = ax * ax + (1 - ax * ax) * cs; m.m[0][1] = ax * ay * (1 - cs) - az * sn; m.m[0][2] = ax * az * (1 - cs) + ay * sn;
    m.m[1][0] = ax * ay * (1 - cs) + az * sn; m.m[1][1] = ay * ay + (1 - ay * ay) * cs; m.m[1][2] = ay * az * (1 - cs) - ax * sn;
    m.m[2][0] = ax * az * (1 - cs) - ay * sn; m.m[2][1] = ay * az * (1 - cs) + ax * sn; m.m[2][2] = az * az + (1 - az * az) * cs;
    return m;
}
inline bool m4_look_at(const float a[9], M4& w2c) {
    auto norm = [](float v[3]) { const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (l == 0) return false; v[0] /= l; v[1] /= l; v[2] /= l; return true; };
    float dir[3] = {a[3] - a[0], a[4] - a[1], a[5] - a[2]}, up[3] = {a[6], a[7], a[8]};
    if (!norm(dir) || !norm(up)) return false;
    float right[3] = {up[1] * dir[2] - up[2] * dir[1], up[2] * dir[0] - up[0] * dir[2], up[0] * dir[1] - up[1] * dir[0]};
    if (!norm(right)) return false;
    const float nup[3] = {dir[1] * right[2] - dir[2] * right[1], dir[2] * right[0] - dir[0] * right[2], dir[0] * right[1] - dir[1] * right[0]};
    M4 c2w = m4_identity();
    for (int r = 0; r < 3; ++r) { c2w.m[r][0] = right[r]; c2w.m[r][1] = nup[r]; c2w.m[r][2] = dir[r]; c2w.m[r][3] = a[r]; }
    return m4_inverse(c2w, w2c);
}

struct PbrtToken { enum Kind { Word, String, Number, Open, Close, End } kind; std::string text; double value; };

struct PbrtLexer {
    const std::string& s;
    size_t i = 0;
    explicit PbrtLexer(const std::string& text) : s(text) {}
    PbrtToken next() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
        PbrtToken t; t.value = 0;
        if (i >= s.size()) { t.kind = PbrtToken::End; return t; }
        const char c = s[i];
        if (c == '[') { ++i; t.kind = PbrtToken::Open; return t; }
        if (c == ']') { ++i; t.kind = PbrtToken::Close; return t; }
        if (c == '"') {
            const size_t e = s.find('"', i + 1);
            t.kind = PbrtToken::String;
            t.text = s.substr(i + 1, (e == std::string::npos ? s.size() : e) - i - 1);
            i = e == std::string::npos ? s.size() : e + 1;
            return t;
        }
        if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
            char* end = nullptr;
            t.value = std::strtod(s.c_str() + i, &end);
            if (end != s.c_str() + i) { t.kind = PbrtToken::Number; i = (size_t)(end - s.c_str()); return t; }
        }
        size_t e = i;
        while (e < s.size() && !std::isspace((unsigned char)s[e]) && s[e] != '[' && s[e] != ']' && s[e] != '"') ++e;
        if (e == i) ++e;
        t.kind = PbrtToken::Word; t.text = s.substr(i, e - i); i = e;
        return t;
    }
    PbrtToken peek() { const size_t save = i; PbrtToken t = next(); i = save; return t; }
};

// numbers of a directive's fixed arguments (optionally bracketed, as in `Transform [ ... ]`)
inline bool read_numbers(PbrtLexer& lx, size_t n, std::vector<float>& out) {
    out.clear();
    bool bracket = false;
    if (lx.peek().kind == PbrtToken::Open) { lx.next(); bracket = true; }
    for (size_t k = 0; k < n; ++k) {
        PbrtToken t = lx.next();
        if (t.kind != PbrtToken::Number) return false;
        out.push_back((float)t.value);
    }
    if (bracket && lx.next().kind != PbrtToken::Close) return false;
    return true;
}

struct PbrtParam { std::string type, name; std::vector<double> numbers; };

// parameter list after a directive: ("type name" value | [ values ])*
inline bool read_params(PbrtLexer& lx, std::vector<PbrtParam>& out) {
    out.clear();
    while (lx.peek().kind == PbrtToken::String) {
        if (lx.peek().text.find_first_of(" \t") == std::string::npos) break;   // not a "type name" declaration
        PbrtToken decl = lx.next();
        PbrtParam p;
        const size_t sp = decl.text.find_first_of(" \t");
        p.type = decl.text.substr(0, sp);
        size_t b = decl.text.find_first_not_of(" \t", sp);
        p.name = b == std::string::npos ? std::string() : decl.text.substr(b);
        PbrtToken v = lx.next();
        if (v.kind == PbrtToken::Open) {
            for (;;) {
                PbrtToken e = lx.next();
                if (e.kind == PbrtToken::Close) break;
                if (e.kind == PbrtToken::End) return false;
                if (e.kind == PbrtToken::Number) p.numbers.push_back(e.value);
            }
        } else if (v.kind == PbrtToken::Number) {
            p.numbers.push_back(v.value);
        } else if (v.kind != PbrtToken::String && v.kind != PbrtToken::Word) {
            return false;
        }
        out.push_back(std::move(p));
    }
    return true;
}


}  // namespace trc
