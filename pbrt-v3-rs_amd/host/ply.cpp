// PLY reader for Shape "plymesh" (shapes/src/plymesh.rs:21-249): ascii / binary_little_endian / binary_big_endian,
// vertex element with x y z [nx ny nz] [u v | s t | texture_u texture_v | texture_s texture_t], face element with a
// vertex_indices list of 3 or 4 entries; a quad (a b c d) becomes (a b c) (d a c) as the reference splits it (:219-229).
#include "pbrt_host.hpp"
#include <cstring>
#include <fstream>
#include <sstream>

namespace pbrt_host {
namespace {

enum class PT { I8, U8, I16, U16, I32, U32, F32, F64, Bad };
PT parse_type(const std::string& t) {
    if (t == "char" || t == "int8") return PT::I8;
    if (t == "uchar" || t == "uint8") return PT::U8;
    if (t == "short" || t == "int16") return PT::I16;
    if (t == "ushort" || t == "uint16") return PT::U16;
    if (t == "int" || t == "int32") return PT::I32;
    if (t == "uint" || t == "uint32") return PT::U32;
    if (t == "float" || t == "float32") return PT::F32;
    if (t == "double" || t == "float64") return PT::F64;
    return PT::Bad;
}
size_t type_size(PT t) {
    switch (t) { case PT::I8: case PT::U8: return 1; case PT::I16: case PT::U16: return 2; case PT::I32: case PT::U32: case PT::F32: return 4; case PT::F64: return 8; default: return 0; }
}
struct Prop { std::string name; bool is_list = false; PT count_type = PT::Bad, type = PT::Bad; };
struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };

struct Reader {
    std::istream& in; int fmt;  // 0 ascii, 1 little, 2 big
    bool ok = true;
    double scalar(PT t) {
        if (fmt == 0) {
            std::string tok;
            if (!(in >> tok)) { ok = false; return 0; }
            if (t == PT::F32) return (double)std::strtof(tok.c_str(), nullptr);  // round the decimal once, to f32
            if (t == PT::F64) return std::strtod(tok.c_str(), nullptr);
            return (double)std::strtoll(tok.c_str(), nullptr, 10);
        }
        unsigned char b[8]; const size_t n = type_size(t);
        if (!in.read((char*)b, (std::streamsize)n)) { ok = false; return 0; }
        if (fmt == 2) for (size_t i = 0; i < n / 2; i++) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
            case PT::I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
            case PT::U8: return b[0];
            case PT::I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
            case PT::U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
            case PT::I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
            case PT::U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
            case PT::F32: { float v; std::memcpy(&v, b, 4); return v; }
            case PT::F64: { double v; std::memcpy(&v, b, 8); return v; }
            default: ok = false; return 0;
        }
    }
};

}  // namespace

bool read_ply(const std::string& path, PlyMesh& out, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open file"; return false; }
    std::string line;
    if (!std::getline(f, line) || line.compare(0, 3, "ply") != 0) { err = "missing 'ply' magic"; return false; }
    int fmt = -1; std::vector<Elem> elems;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line); std::string kw; ls >> kw;
        if (kw == "format") { std::string v; ls >> v; fmt = v == "ascii" ? 0 : v == "binary_little_endian" ? 1 : v == "binary_big_endian" ? 2 : -1; }
        else if (kw == "comment" || kw == "obj_info" || kw.empty()) continue;
        else if (kw == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
        else if (kw == "property") {
            if (elems.empty()) { err = "property before any element"; return false; }
            Prop p; std::string t; ls >> t;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.is_list = true; p.count_type = parse_type(ct); p.type = parse_type(it); if (p.count_type == PT::Bad) { err = "bad list count type"; return false; } }
            else { p.type = parse_type(t); ls >> p.name; }
            if (p.type == PT::Bad) { err = "unknown property type in '" + line + "'"; return false; }
            elems.back().props.push_back(p);
        }
        else if (kw == "end_header") break;
        else { err = "unexpected header line '" + line + "'"; return false; }
    }
    if (fmt < 0) { err = "unknown or missing format"; return false; }
    Reader rd{f, fmt};
    bool has_n = true, has_uv = true;
    for (const Elem& e : elems) {
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        for (size_t k = 0; k < e.count; k++) {
            float p[3] = {0, 0, 0}, n[3] = {0, 0, 0}, uv[2] = {0, 0}; int nc = 0, uvc = 0;
            for (const Prop& pr : e.props) {
                if (pr.is_list) {
                    const long cnt = (long)rd.scalar(pr.count_type);
                    if (!rd.ok || cnt < 0 || cnt > (1 << 20)) { err = "truncated or corrupt list"; return false; }
                    long vi[4] = {0, 0, 0, 0};
                    for (long j = 0; j < cnt; j++) { const double v = rd.scalar(pr.type); if (j < 4) vi[j] = (long)v; }
                    if (!rd.ok) { err = "truncated data"; return false; }
                    if (is_face && pr.name == "vertex_indices") {
                        if (cnt != 3 && cnt != 4) { err = "Only triangles and quads are supported!"; return false; }
                        out.indices.push_back((uint32_t)vi[0]); out.indices.push_back((uint32_t)vi[1]); out.indices.push_back((uint32_t)vi[2]);
                        if (cnt == 4) { out.indices.push_back((uint32_t)vi[3]); out.indices.push_back((uint32_t)vi[0]); out.indices.push_back((uint32_t)vi[2]); }
                    }
                    continue;
                }
                const double v = rd.scalar(pr.type);
                if (!rd.ok) { err = "truncated data"; return false; }
                if (!is_vertex || (pr.type != PT::F32 && pr.type != PT::F64)) continue;
                const float fv = (float)v; const std::string& nm = pr.name;
                if (nm == "x") p[0] = fv; else if (nm == "y") p[1] = fv; else if (nm == "z") p[2] = fv;
                else if (nm == "nx") { n[0] = fv; nc++; } else if (nm == "ny") { n[1] = fv; nc++; } else if (nm == "nz") { n[2] = fv; nc++; }
                else if (nm == "u" || nm == "s" || nm == "texture_u" || nm == "texture_s") { uv[0] = fv; uvc++; }
                else if (nm == "v" || nm == "t" || nm == "texture_v" || nm == "texture_t") { uv[1] = fv; uvc++; }
            }
            if (is_vertex) {
                out.P.insert(out.P.end(), p, p + 3);
                has_n = has_n && nc == 3; if (has_n) out.N.insert(out.N.end(), n, n + 3);
                has_uv = has_uv && uvc == 2; if (has_uv) out.UV.insert(out.UV.end(), uv, uv + 2);
            }
        }
    }
    if (!has_n) out.N.clear();
    if (!has_uv) out.UV.clear();
    const size_t nv = out.P.size() / 3;
    for (uint32_t i : out.indices) if (i >= nv) { err = "vertex index " + std::to_string(i) + " out of bounds"; return false; }
    return true;
}

}  // namespace pbrt_host
