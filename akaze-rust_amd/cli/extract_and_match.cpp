// extract_and_match INPUT_0 INPUT_1 OUTPUT_PREFIX [-m IMAGE_FILE_PATH] — akaze-util/src/bin/extract_and_match.rs:12-136.
#include "cli_common.hpp"

int main(int argc, char** argv) {
    const cli::Spec spec{
        "Extract and match KAZE image features..",
        "A Rust implementation of the KAZE visual feature extractor and matching (here: its MI355X build).\n"
        "See https://github.com/pablofdezalc/kaze for the original authors' project.\n"
        "Set AKAZE_LOG to debug for more verbose output. This executable runs the entire\n"
        "pipeline end-to-end for two images. For more granular control, see the binaries\n"
        "extract_features and match_features.",
        {{"INPUT_0", "The first input image."}, {"INPUT_1", "The second input image."},
         {"OUTPUT_PREFIX", "The output prefix for all files."}},
        {{'m', "match_image", "IMAGE_FILE_PATH", "Sets a path to write the match image to."}}};
    const cli::Args a = cli::parse(spec, argc, argv);
    const cli::Timer timer;
    if (const char* mi = a.get("match_image")) cli::require_png_path(mi, "--match_image");
    const char* in0 = a.pos[0].c_str();
    const char* in1 = a.pos[1].c_str();
    const std::string prefix = a.pos[2];
    CLI_INFO("Input image paths are %s/%s, output extractions path is %s, threshold is %d.", in0, in1, prefix.c_str(), 10);
    akz_config options;
    akz_config_default(&options);
    const std::string e0 = prefix + "-extractions_0.cbor", e1 = prefix + "-extractions_1.cbor", mp = prefix + "-matches.cbor";

    akz_ctx* ctx = cli::open_context();
    const cli::Features f0 = cli::extract(ctx, in0, options, false);
    CLI_TRY(akz_write_features(e0.c_str(), f0.keypoints.data(), f0.keypoints.size(), f0.descriptors.data(), f0.desc_bytes));
    CLI_INFO("Done, extracted %zu features from image 0.", f0.keypoints.size());
    const cli::Features f1 = cli::extract(ctx, in1, options, false);
    CLI_TRY(akz_write_features(e1.c_str(), f1.keypoints.data(), f1.keypoints.size(), f1.descriptors.data(), f1.desc_bytes));
    CLI_INFO("Done, extracted %zu features from image 1, proceeding with matching.", f1.keypoints.size());
    const std::vector<akz_match> m = cli::match(ctx, f0, f1);
    CLI_INFO("Got %zu matches.", m.size());
    CLI_TRY(akz_write_matches(mp.c_str(), m.data(), m.size()));
    if (const char* mi = a.get("match_image")) {
        CLI_INFO("Writing scale space");
        uint32_t w0 = 0, h0 = 0, w1 = 0, h1 = 0, ow = 0, oh = 0;
        uint8_t *r0 = nullptr, *r1 = nullptr, *out = nullptr;
        CLI_TRY(akz_image_load_rgb(in0, &w0, &h0, &r0));
        CLI_TRY(akz_image_load_rgb(in1, &w1, &h1, &r1));
        CLI_TRY(akz_draw_matches(r0, w0, h0, r1, w1, h1, f0.keypoints.data(), f0.keypoints.size(), f1.keypoints.data(),
                                 f1.keypoints.size(), m.data(), m.size(), &ow, &oh, &out));
        if (akz_image_save_png(mi, out, ow, oh, 3) == AKZ_OK) CLI_DEBUG("Wrote matches image successfully.");
        else CLI_DEBUG("Could not write matches image for some reason, skipping.");
        akz_image_free(r0); akz_image_free(r1); akz_image_free(out);
    }
    akz_ctx_destroy(ctx);
    CLI_DEBUG("Total duration: %.3fs", timer.seconds());
    return 0;
}
