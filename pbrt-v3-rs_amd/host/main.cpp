// pbrt_hip_render — command-line front end (src/main.rs + core/src/app/options.rs): parse scene files, render each on
// one MI355X (or, with --devices, on several of one node) through libpbrt_hip.so, write the image.  No CPU rendering path exists: without a device it exits non-zero.
#include "pbrt_host.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

static void usage() {
    std::fprintf(stderr,
                 "usage: pbrt_hip_render [options] <scene.pbrt>...\n"
                 "  --outfile FILE         write the image to FILE (PFM)\n"
                 "  --cropwindow X0 X1 Y0 Y1\n"
                 "  --device N             GPU ordinal (default 0)\n"
                 "  --devices A,B,...      render every frame on several GPUs of this node (tiles dealt round-robin, film tiles gathered over RCCL); 'all' = every visible GPU\n"
                 "  --tile-size N          sample tile edge (default 16, as the reference)\n"
                 "  --sobol-tables FILE    raw Sobol generator matrices (needed for Sampler \"sobol\")\n"
                 "  --check                parse and validate only: no GPU is touched and nothing is rendered\n"
                 "  --convert-image IN OUT       decode an image file (PFM, TGA, PNG, EXR) the way ImageTexture would see it and write it as .pfm / .exr / .png / .tga\n"
                 "  --quiet                no warnings / statistics\n");
}

int main(int argc, char** argv) {
    std::string outfile, sobol; int device = 0, tile = 16; bool quiet = false, has_crop = false, check = false; float crop[4] = {0, 1, 0, 1};
    std::vector<std::string> files;
    std::vector<int> devices; bool multi = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { usage(); std::exit(2); } };
        if (a == "--outfile" || a == "-o") { need(1); outfile = argv[++i]; }
        else if (a == "--cropwindow") { need(4); for (int k = 0; k < 4; k++) crop[k] = std::strtof(argv[++i], nullptr); has_crop = true; }
        else if (a == "--device") { need(1); device = std::atoi(argv[++i]); }
        else if (a == "--devices") {
            need(1); multi = true;
            const std::string v = argv[++i];
            if (v != "all")
                for (size_t p = 0; p < v.size();) { size_t q = v.find(',', p); if (q == std::string::npos) q = v.size(); devices.push_back(std::atoi(v.substr(p, q - p).c_str())); p = q + 1; }
        }
        else if (a == "--tile-size") { need(1); tile = std::atoi(argv[++i]); }
        else if (a == "--sobol-tables") { need(1); sobol = argv[++i]; }
        else if (a == "--quiet") quiet = true;
        else if (a == "--check") check = true;
        else if (a == "--convert-image") {
            need(2);
            std::vector<float> rgb; int w = 0, h = 0; std::string err;
            if (!pbrt_host::read_image(argv[i + 1], rgb, w, h, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            if (!pbrt_host::write_image(argv[i + 2], rgb.data(), w, h, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            return 0;
        }
        else if (a == "--help" || a == "-h") { usage(); return 0; }
        else if (a.size() > 1 && a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 2; }
        else files.push_back(a);
    }
    if (files.empty()) { usage(); return 2; }
    for (const std::string& fn : files) {
        std::unique_ptr<pbrt_host::Api> api_owner((multi && !check) ? new pbrt_host::Api(devices) : new pbrt_host::Api(check ? -1 : device));
        pbrt_host::Api& api = *api_owner;
        if (!api.error.empty()) { std::fprintf(stderr, "Error: %s\n", api.error.c_str()); return 3; }
        api.override_outfile = outfile; api.sobol_tables_file = sobol; api.tile_size = tile; api.quiet = quiet;
        api.has_crop_override = has_crop; std::memcpy(api.crop_override, crop, sizeof crop);
        pbrt_host::RenderReport rep;
        if (!pbrt_host::parse_file(fn, api, &rep)) { std::fprintf(stderr, "Error: %s: %s\n", fn.c_str(), api.error.c_str()); return 1; }
        if (check) {
            std::printf("{\"file\": \"%s\", \"triangles\": %llu, \"lights\": %llu, \"instances\": %llu, \"xres\": %d, \"yres\": %d, \"crop\": [%d, %d, %d, %d], \"spp\": %d, \"max_depth\": %d, "
                        "\"light_strategy\": %d, \"pixel_bounds\": [%d, %d, %d, %d], \"out_file\": \"%s\", \"warnings\": %zu}\n",
                        fn.c_str(), (unsigned long long)rep.n_triangles, (unsigned long long)rep.n_lights, (unsigned long long)rep.n_instances, rep.xres, rep.yres, rep.crop[0], rep.crop[1], rep.crop[2], rep.crop[3],
                        rep.spp, rep.max_depth, rep.light_strategy, rep.pixel_bounds[0], rep.pixel_bounds[1], rep.pixel_bounds[2], rep.pixel_bounds[3], rep.out_file.c_str(), rep.warnings.size());
            continue;
        }
        if (!quiet && !rep.out_file.empty()) {
            const PbrtHipStats& s = rep.stats;
            const double rays = (double)(s.regular_rays + s.shadow_rays);
            std::printf("%s: %llu triangles, %llu lights, BVH %.3f s, render %.3f s, %.1f Mrays/s (%llu regular + %llu shadow rays), zero-radiance paths %.2f%% -> %s\n",
                        fn.c_str(), (unsigned long long)rep.n_triangles, (unsigned long long)rep.n_lights, rep.build_seconds, s.render_seconds,
                        s.render_seconds > 0 ? rays / s.render_seconds * 1e-6 : 0.0, (unsigned long long)s.regular_rays, (unsigned long long)s.shadow_rays,
                        s.paths_total ? 100.0 * (double)s.paths_zero_radiance / (double)s.paths_total : 0.0, rep.out_file.c_str());
        }
    }
    return 0;
}
