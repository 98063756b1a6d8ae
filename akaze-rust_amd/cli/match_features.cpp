// match_features INPUT_EXTRACTIONS_0 INPUT_EXTRACTIONS_1 OUTPUT [-t FLOAT] — akaze-util/src/bin/match_features.rs:14-85.
// (The reference binary looks its two inputs up under lower-case names that clap does not know
// (match_features.rs:52-53) and panics before doing any work; this tool does what its help text says.  As in
// the reference, the threshold option is parsed and logged but the matcher runs with the fixed arguments
// 0.86 / 1000 / 3.0, :70-78.)
#include "cli_common.hpp"

int main(int argc, char** argv) {
    const cli::Spec spec{
        "Feature matching using Hamming distance for AKAZE features.",
        "A Rust implementation of the KAZE visual feature matching using\n"
        "Hamming distance for binary descriptors (here: its MI355X build). For use with AKAZE.\n"
        "Set AKAZE_LOG to debug for more verbose output.",
        {{"INPUT_EXTRACTIONS_0", "The input extraction results for image 0."},
         {"INPUT_EXTRACTIONS_1", "The input extraction results for image 1."},
         {"OUTPUT", "The output matches."}},
        {{'t', "threshold", "FLOAT", "The distance threshold for the matcher."}}};
    const cli::Args a = cli::parse(spec, argc, argv);
    const cli::Timer timer;
    const char* thr = a.get("threshold");
    const double threshold = thr ? atof(thr) : 10.0;
    CLI_INFO("Input extractions: %s/%s, output matches: %s, threshold: %g.", a.pos[0].c_str(), a.pos[1].c_str(),
             a.pos[2].c_str(), threshold);
    const cli::Features f0 = cli::read_features(a.pos[0].c_str());
    const cli::Features f1 = cli::read_features(a.pos[1].c_str());
    akz_ctx* ctx = cli::open_context();
    const std::vector<akz_match> m = cli::match(ctx, f0, f1);
    CLI_TRY(akz_write_matches(a.pos[2].c_str(), m.data(), m.size()));
    CLI_DEBUG("Done, got %zu matches, total duration: %.3fs", m.size(), timer.seconds());
    akz_ctx_destroy(ctx);
    return 0;
}
