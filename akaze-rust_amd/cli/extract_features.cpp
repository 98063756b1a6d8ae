// extract_features INPUT OUTPUT [-d DIRECTORY] [-o PATH] — akaze-util/src/bin/extract_features.rs:18-112.
#include "cli_common.hpp"

int main(int argc, char** argv) {
    const cli::Spec spec{
        "KAZE extractor.",
        "A Rust implementation of the KAZE visual feature extractor (here: its MI355X build). See\n"
        "https://github.com/pablofdezalc/kaze for the original authors' project.\n"
        "Set AKAZE_LOG to debug for more verbose output.",
        {{"INPUT", "The input image."}, {"OUTPUT", "The output extractions. Extension can be JSON or CBOR."}},
        {{'d', "debug_path", "DIRECTORY", "Sets a directory to write debug information to."},
         {'o', "options", "PATH", "A JSON file containing options."}}};
    const cli::Args a = cli::parse(spec, argc, argv);
    const cli::Timer timer;
    const char* input = a.pos[0].c_str();
    const char* output = a.pos[1].c_str();
    CLI_INFO("Input image path is %s, output extractions path is %s.", input, output);

    akz_config options;
    akz_config_default(&options);
    if (const char* op = a.get("options")) {  // read the file if it exists, else write the defaults to it (:66-83)
        if (cli::file_exists(op)) {
            CLI_INFO("Reading options file from %s", op);
            FILE* f = fopen(op, "rb");
            std::string text;
            char buf[4096];
            size_t n;
            while (f && (n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
            if (f) fclose(f);
            CLI_TRY(akz_config_from_json(text.c_str(), &options));
        } else {
            char buf[1024];
            uint64_t len = 0;
            CLI_TRY(akz_config_to_json(&options, buf, sizeof(buf), &len));
            FILE* f = fopen(op, "wb");
            if (!f || fwrite(buf, 1, len, f) != len) { fprintf(stderr, "error: cannot write %s\n", op); return 1; }
            fclose(f);
            CLI_INFO("Writing options file from %s", op);
        }
    } else {
        CLI_DEBUG("Using default options.");
    }

    akz_ctx* ctx = cli::open_context();
    const char* debug_dir = a.get("debug_path");
    akz_result* res = nullptr;
    const cli::Features f = cli::extract(ctx, input, options, debug_dir != nullptr, &res);
    CLI_TRY(akz_write_features(output, f.keypoints.data(), f.keypoints.size(), f.descriptors.data(), f.desc_bytes));
    CLI_INFO("Done, extracted %zu features.", f.keypoints.size());
    if (debug_dir) {
        CLI_INFO("Writing scale space since --debug_path/-d option was specified.");
        if (!cli::mkdir_p(debug_dir)) { fprintf(stderr, "error: cannot create %s\n", debug_dir); return 1; }
        CLI_TRY(akz_write_evolutions(res, 0, debug_dir));
        uint32_t w = 0, h = 0;
        uint8_t* rgb = nullptr;
        CLI_TRY(akz_image_load_rgb(input, &w, &h, &rgb));
        CLI_TRY(akz_draw_keypoints(rgb, w, h, f.keypoints.data(), f.keypoints.size()));
        const std::string kp_path = std::string(debug_dir) + "/keypoints.png";
        if (akz_image_save_png(kp_path.c_str(), rgb, w, h, 3) == AKZ_OK) CLI_DEBUG("Wrote keypoint image successfully.");
        else CLI_DEBUG("Could not write keypoint image for some reason, skipping.");
        akz_image_free(rgb);
    } else {
        CLI_DEBUG("Argument --debug_path/-d was not given, not writing debug directory.");
    }
    akz_result_free(res);
    akz_ctx_destroy(ctx);
    CLI_DEBUG("Total duration: %.3fs", timer.seconds());
    return 0;
}
