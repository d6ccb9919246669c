// pbrt_text.hpp -- reading pbrt-v3 scene text on the host (shared by the density and the triangle-mesh readers).
#pragma once

#include <cctype>
#include <cstdio>
#include <string>

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


}  // namespace trc
