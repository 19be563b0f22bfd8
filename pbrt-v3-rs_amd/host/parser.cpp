// Tokenizer + directive dispatch for the .pbrt scene format (subset of api/src/parser/grammar.pest).
// Directives outside the hot-path scope (media, object instancing, animated transforms) stop the parse with an
// error that names them: silently skipping them would render a different image than the reference.
#include "pbrt_host.hpp"
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pbrt_host {
namespace {

struct Token { enum Kind { Ident, Str, Num, LBracket, RBracket, End } kind = End; std::string text; int line = 0; };

struct Lexer {
    const std::string& s; size_t i = 0; int line = 1;
    explicit Lexer(const std::string& src) : s(src) {}
    Token next() {
        for (;;) {  // whitespace and '#' comments
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n')) { if (s[i] == '\n') line++; i++; }
            if (i < s.size() && s[i] == '#') { while (i < s.size() && s[i] != '\n') i++; continue; }
            break;
        }
        Token t; t.line = line;
        if (i >= s.size()) return t;
        const char c = s[i];
        if (c == '[') { i++; t.kind = Token::LBracket; return t; }
        if (c == ']') { i++; t.kind = Token::RBracket; return t; }
        if (c == '"') {
            size_t j = ++i;
            while (j < s.size() && s[j] != '"') { if (s[j] == '\n') line++; j++; }
            t.kind = Token::Str; t.text = s.substr(i, j - i); i = j < s.size() ? j + 1 : j;
            return t;
        }
        size_t j = i;
        while (j < s.size() && !std::strchr(" \t\r\n[]\"#", s[j])) j++;
        t.text = s.substr(i, j - i); i = j;
        const char f = t.text[0];
        t.kind = ((f >= '0' && f <= '9') || f == '-' || f == '+' || f == '.') ? Token::Num : Token::Ident;
        return t;
    }
};

// parse_float_file (core/src/float_file): numbers separated by white space, '#' starts a comment that runs to the end of the line
bool read_float_file(const std::string& path, std::vector<float>& out, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "Error reading file '" + path + "': " + std::strerror(errno); return false; }
    std::string line; int line_no = 0;
    while (std::getline(f, line)) {
        line_no++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ls(line);
        std::string tok;
        while (ls >> tok) {
            char* end = nullptr; const float v = std::strtof(tok.c_str(), &end);
            if (!end || *end) { err = "Error parsing floating point number '" + tok + "', line " + std::to_string(line_no) + "."; return false; }
            out.push_back(v);
        }
    }
    return true;
}

struct Parser {
    Api& api; std::string scene_dir; RenderReport* report; int depth;
    Lexer lx; Token cur;
    std::string err;
    Parser(const std::string& src, Api& a, const std::string& dir, RenderReport* rep, int d) : api(a), scene_dir(dir), report(rep), depth(d), lx(src) { cur = lx.next(); }
    void advance() { cur = lx.next(); }
    bool fail(const std::string& m) { if (err.empty()) err = "line " + std::to_string(cur.line) + ": " + m; return false; }

    bool number(float& out) {  // Rust's str::parse::<f32> and strtof both round the decimal correctly
        if (cur.kind != Token::Num) return fail("expected a number, found '" + cur.text + "'");
        char* end = nullptr; errno = 0;
        out = std::strtof(cur.text.c_str(), &end);
        if (!end || *end) return fail("malformed number '" + cur.text + "'");
        advance();
        return true;
    }
    bool numbers(float* out, int n) { for (int i = 0; i < n; i++) if (!number(out[i])) return false; return true; }
    bool quoted(std::string& out) {
        if (cur.kind != Token::Str) return fail("expected a quoted string, found '" + cur.text + "'");
        out = cur.text; advance();
        return true;
    }
    bool bracketed16(float* out) {
        if (cur.kind != Token::LBracket) return fail("expected '['");
        advance();
        if (!numbers(out, 16)) return false;
        if (cur.kind != Token::RBracket) return fail("expected ']' after 16 values");
        advance();
        return true;
    }

    // "type name" value | [ values ]
    bool param_list(ParamSet& ps) {
        while (cur.kind == Token::Str) {
            std::istringstream decl(cur.text);
            std::string type, name, extra;
            decl >> type >> name;
            if (type.empty() || name.empty() || (decl >> extra)) return fail("parameter declaration \"" + cur.text + "\" should be \"type name\"");
            advance();
            std::vector<Token> vals;
            if (cur.kind == Token::LBracket) {
                advance();
                while (cur.kind != Token::RBracket) {
                    if (cur.kind == Token::End) return fail("unterminated '[' in parameter '" + name + "'");
                    if (cur.kind != Token::Num && cur.kind != Token::Str) return fail("unexpected token in value list of '" + name + "'");
                    vals.push_back(cur); advance();
                }
                advance();
            } else if (cur.kind == Token::Num || cur.kind == Token::Str) {
                vals.push_back(cur); advance();
            } else return fail("parameter '" + name + "' has no value");

            auto as_floats = [&](std::vector<float>& out) {
                for (auto& v : vals) {
                    if (v.kind != Token::Num) return fail("parameter '" + name + "' expects numbers");
                    char* end = nullptr; float f = std::strtof(v.text.c_str(), &end);
                    if (!end || *end) return fail("malformed number '" + v.text + "'");
                    out.push_back(f);
                }
                return true;
            };
            if (type == "float" || type == "point" || type == "point3" || type == "point2" || type == "vector" || type == "vector3" || type == "vector2" ||
                type == "normal" || type == "normal3" || type == "rgb" || type == "color" || type == "colour") {
                std::vector<float> f;
                if (!as_floats(f)) return false;
                const int arity = (type == "float") ? 1 : (type == "point2" || type == "vector2") ? 2 : 3;
                if (f.size() % arity) return fail("parameter '" + name + "': length is not divisible by " + std::to_string(arity));
                ps.floats[name] = f;
            } else if (type == "xyz") {  // RGBSpectrum::from_xyz (core/src/spectrum/common.rs:337-343)
                std::vector<float> f;
                if (!as_floats(f)) return false;
                if (f.size() % 3) return fail("parameter '" + name + "': length is not divisible by 3");
                for (size_t k = 0; k + 2 < f.size(); k += 3) {
                    const float x = f[k], y = f[k + 1], z = f[k + 2];
                    f[k] = 3.240479f * x - 1.537150f * y - 0.498535f * z;
                    f[k + 1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
                    f[k + 2] = 0.055648f * x - 0.204043f * y + 1.057311f * z;
                }
                ps.floats[name] = f;
            } else if (type == "integer") {
                std::vector<int> iv;
                for (auto& v : vals) {
                    if (v.kind != Token::Num) return fail("parameter '" + name + "' expects integers");
                    char* end = nullptr; long l = std::strtol(v.text.c_str(), &end, 10);
                    if (!end || *end) return fail("malformed integer '" + v.text + "'");
                    iv.push_back((int)l);
                }
                ps.ints[name] = iv;
            } else if (type == "bool") {
                std::vector<bool> bv;
                for (auto& v : vals) {
                    if (v.text == "true") bv.push_back(true); else if (v.text == "false") bv.push_back(false);
                    else return fail("parameter '" + name + "' expects \"true\" or \"false\"");
                }
                ps.bools[name] = bv;
            } else if (type == "string" || type == "texture") {
                std::vector<std::string> sv;
                for (auto& v : vals) { if (v.kind != Token::Str) return fail("parameter '" + name + "' expects strings"); sv.push_back(v.text); }
                (type == "string" ? ps.strings : ps.textures)[name] = sv;
            } else if (type == "blackbody") {  // ParamSet::add_blackbody_spectrum (paramset/mod.rs:236-249): (temperature, scale) pairs
                std::vector<float> f, rgb;
                if (!as_floats(f)) return false;
                if (f.size() % 2) return fail("parameter '" + name + "': blackbody values come in (temperature, scale) pairs");
                for (size_t k = 0; k + 1 < f.size(); k += 2) { float c[3]; pbrt_hip_host_blackbody_rgb(f[k], f[k + 1], c); rgb.insert(rgb.end(), c, c + 3); }
                ps.floats[name] = rgb;
            } else if (type == "spectrum") {
                std::vector<float> rgb;
                if (!vals.empty() && vals[0].kind == Token::Str) {  // SPD files (add_sampled_spectrum_files, :268-315): one spectrum per file name
                    for (auto& v : vals) {
                        if (v.kind != Token::Str) return fail("parameter '" + name + "' mixes file names and numbers");
                        const std::string path = (v.text.empty() || v.text[0] == '/' || scene_dir.empty()) ? v.text : scene_dir + "/" + v.text;
                        std::vector<float> pairs; std::string ferr;
                        float c[3] = {0.0f, 0.0f, 0.0f};
                        if (!read_float_file(path, pairs, ferr)) api.warn("Unable to read SPD file '" + v.text + "'. Using black distribution. " + ferr);
                        else {
                            if (pairs.size() % 2) api.warn("Extra value found in spectrum file '" + v.text + "'. Ignoring it.");
                            if (pairs.size() >= 2) pbrt_hip_host_sampled_rgb(pairs.data(), pairs.size() / 2, c);
                        }
                        rgb.insert(rgb.end(), c, c + 3);
                    }
                } else {  // inline (wavelength, value) pairs: ONE spectrum (add_sampled_spectrum, :255-262)
                    std::vector<float> f;
                    if (!as_floats(f)) return false;
                    if (f.size() < 2 || f.size() % 2) return fail("parameter '" + name + "': a sampled spectrum is a list of (wavelength, value) pairs");
                    float c[3]; pbrt_hip_host_sampled_rgb(f.data(), f.size() / 2, c);
                    rgb.assign(c, c + 3);
                }
                ps.floats[name] = rgb;
            } else return fail("unknown parameter type '" + type + "'");
        }
        return true;
    }

    bool name_and_params(std::string& name, ParamSet& ps) { return quoted(name) && param_list(ps); }

    bool run() {
        while (cur.kind != Token::End) {
            if (cur.kind != Token::Ident) return fail("expected a directive, found '" + cur.text + "'");
            const std::string d = cur.text;
            advance();
            std::string name; ParamSet ps; float v[16];
            if (d == "Identity") api.pbrt_identity();
            else if (d == "Translate") { if (!numbers(v, 3)) return false; api.pbrt_translate(v[0], v[1], v[2]); }
            else if (d == "Scale") { if (!numbers(v, 3)) return false; api.pbrt_scale(v[0], v[1], v[2]); }
            else if (d == "Rotate") { if (!numbers(v, 4)) return false; api.pbrt_rotate(v[0], v[1], v[2], v[3]); }
            else if (d == "LookAt") { if (!numbers(v, 9)) return false; api.pbrt_look_at(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
            else if (d == "ConcatTransform") { if (!bracketed16(v)) return false; api.pbrt_concat_transform(v); }
            else if (d == "Transform") { if (!bracketed16(v)) return false; api.pbrt_transform(v); }
            else if (d == "CoordinateSystem") { if (!quoted(name)) return false; api.pbrt_coordinate_system(name); }
            else if (d == "CoordSysTransform") { if (!quoted(name)) return false; api.pbrt_coord_sys_transform(name); }
            else if (d == "Camera") { if (!name_and_params(name, ps)) return false; api.pbrt_camera(name, ps); }
            else if (d == "Film") { if (!name_and_params(name, ps)) return false; api.pbrt_film(name, ps); }
            else if (d == "Sampler") { if (!name_and_params(name, ps)) return false; api.pbrt_sampler(name, ps); }
            else if (d == "PixelFilter") { if (!name_and_params(name, ps)) return false; api.pbrt_pixel_filter(name, ps); }
            else if (d == "Accelerator") { if (!name_and_params(name, ps)) return false; api.pbrt_accelerator(name, ps); }
            else if (d == "Integrator" || d == "SurfaceIntegrator") { if (!name_and_params(name, ps)) return false; api.pbrt_integrator(name, ps); }
            else if (d == "WorldBegin") api.pbrt_world_begin();
            else if (d == "AttributeBegin") api.pbrt_attribute_begin();
            else if (d == "AttributeEnd") api.pbrt_attribute_end();
            else if (d == "TransformBegin") api.pbrt_transform_begin();
            else if (d == "TransformEnd") api.pbrt_transform_end();
            else if (d == "ReverseOrientation") api.pbrt_reverse_orientation();
            else if (d == "Material") { if (!name_and_params(name, ps)) return false; api.pbrt_material(name, ps); }
            else if (d == "MakeNamedMaterial") { if (!name_and_params(name, ps)) return false; api.pbrt_make_named_material(name, ps); }
            else if (d == "NamedMaterial") { if (!quoted(name)) return false; api.pbrt_named_material(name); }
            else if (d == "Texture") {
                std::string type, cls;
                if (!quoted(name) || !quoted(type) || !quoted(cls) || !param_list(ps)) return false;
                api.pbrt_texture(name, type, cls, ps, scene_dir);
            }
            else if (d == "LightSource") { if (!name_and_params(name, ps)) return false; api.pbrt_light_source(name, ps, scene_dir); }
            else if (d == "AreaLightSource") { if (!name_and_params(name, ps)) return false; api.pbrt_area_light_source(name, ps); }
            else if (d == "Shape") { if (!name_and_params(name, ps)) return false; api.pbrt_shape(name, ps, scene_dir); }
            else if (d == "ObjectBegin") { if (!quoted(name)) return false; api.pbrt_object_begin(name); }
            else if (d == "ObjectEnd") api.pbrt_object_end();
            else if (d == "ObjectInstance") { if (!quoted(name)) return false; api.pbrt_object_instance(name); }
            else if (d == "Include") {
                if (!quoted(name)) return false;
                if (depth > 32) return fail("Include nesting too deep");
                std::string path = (name[0] != '/' && !scene_dir.empty()) ? scene_dir + "/" + name : name;  // relative to the scene's folder (parser/mod.rs:63-71)
                std::ifstream f(path, std::ios::binary);
                if (!f) return fail("cannot open include file '" + path + "'");
                std::stringstream buf; buf << f.rdbuf();
                const std::string text = buf.str();
                Parser sub(text, api, scene_dir, report, depth + 1);
                if (!sub.run()) return fail("in '" + path + "': " + sub.err);
            }
            else if (d == "WorldEnd") {
                RenderReport local;
                RenderReport& rep = report ? *report : local;
                const int rc = api.pbrt_world_end(rep);
                if (rc != 0) return fail(api.error.empty() ? ("WorldEnd failed with status " + std::to_string(rc)) : api.error);
            }
            else if (d == "MakeNamedMedium" || d == "MediumInterface" || d == "ActiveTransform" || d == "TransformTimes")
                return fail("directive '" + d + "' is outside the hot-path scope of this host (SURVEY §8f)");
            else return fail("unknown directive '" + d + "'");
            if (!api.error.empty()) return fail(api.error);
        }
        return true;
    }
};

}  // namespace

bool parse_string(const std::string& text, const std::string& scene_dir, Api& api, RenderReport* report_out) {
    if (!api.error.empty()) return false;
    Parser p(text, api, scene_dir, report_out, 0);
    if (!p.run()) { if (api.error.empty() || api.error != p.err) api.error = p.err; return false; }
    return true;
}

bool parse_file(const std::string& path, Api& api, RenderReport* report_out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { api.error = "cannot open scene file '" + path + "'"; return false; }
    std::stringstream buf; buf << f.rdbuf();
    const size_t slash = path.find_last_of('/');
    return parse_string(buf.str(), slash == std::string::npos ? std::string(".") : path.substr(0, slash), api, report_out);
}

}  // namespace pbrt_host
