// lacx_cli -- the `encode` command of the reference's command-line tool on the MI355X path (SURVEY row f-4;
// ref src/main.cpp:609-710): same positional arguments, the flags that change the output (--stereo-mode=lr|ms,
// --no-partitioning, --threads=N) with the same meaning and rejection rules, the same messages, staged output
// (written next to the target, renamed on success).  The WAV file goes through lacx_wav_parse / lacx_encode_wav,
// i.e. the raw data chunk is what crosses PCIe.  Decode and selftest stay with the reference's tool.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "lacx.h"

namespace {

void usage() {
    std::cerr << "Usage:\n  lacx_cli encode input.wav output.lac [--stereo-mode=lr|ms] [--threads=N] [--no-partitioning]\n";
}

// --threads=N with N a positive decimal integer (ref src/main.cpp:560-584); anything else is a usage error
bool parse_threads(const std::string& flag, unsigned long long& out, bool& bad) {
    const std::string prefix = "--threads=";
    if (flag.compare(0, prefix.size(), prefix) != 0) return false;
    const std::string v = flag.substr(prefix.size());
    bad = v.empty() || v.size() > 9;
    for (char c : v) bad = bad || c < '0' || c > '9';
    if (!bad) {
        out = std::stoull(v);
        bad = out == 0;
    }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4 || std::string(argv[1]) != "encode") {
        usage();
        return 1;
    }
    const std::string in_path = argv[2], out_path = argv[3];
    if (in_path == out_path) {
        std::cerr << "Input and output paths must be different\n";
        return 1;
    }
    uint8_t stereo_mode = 2;
    bool partitioning = true;
    unsigned long long threads = 0;
    for (int i = 4; i < argc; ++i) {
        const std::string flag = argv[i];
        bool bad = false;
        if (flag == "--no-partitioning") {
            partitioning = false;
        } else if (flag == "--stereo-mode=lr") {
            stereo_mode = 0;
        } else if (flag == "--stereo-mode=ms") {
            stereo_mode = 1;
        } else if (parse_threads(flag, threads, bad)) {
            if (bad) {
                std::cerr << "Error: --threads requires a positive integer\n";
                return 1;
            }
        } else {
            usage();
            return 1;
        }
    }
    std::ifstream in(in_path, std::ios::binary);
    std::vector<uint8_t> wav;
    if (in) wav.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    lacx_wav_info info{};
    if (!in || lacx_wav_parse(wav.data(), wav.size(), &info) != LACX_OK) {
        std::cerr << "Failed to read WAV: " << in_path << "\n";
        return 1;
    }
    lacx_config cfg{};
    cfg.sample_rate = info.sample_rate;
    cfg.bit_depth = (uint8_t)info.bit_depth;
    cfg.stereo_mode = info.channels == 1 ? 0 : stereo_mode;
    cfg.zero_run_enabled = 1;
    cfg.partitioning_enabled = partitioning ? 1 : 0;
    cfg.device = -1;
    cfg.emit_threads = (uint32_t)threads;
    lacx_encoder* enc = nullptr;
    if (lacx_encoder_create(&cfg, &enc) != LACX_OK) {
        std::cerr << "Error: lacx_encoder_create failed\n";
        return 1;
    }
    uint8_t* lac = nullptr;
    uint64_t size = 0;
    const int rc = lacx_encode_wav(enc, wav.data(), wav.size(), &lac, &size);
    if (rc != LACX_OK) {
        std::cerr << "Error: " << lacx_last_error(enc) << "\n";
        lacx_encoder_destroy(enc);
        return 1;
    }
    const std::string tmp = out_path + ".lacx-partial";
    bool ok = false;
    {
        std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
        ok = out && out.write(reinterpret_cast<const char*>(lac), (std::streamsize)size) && out.flush();
    }
    ok = ok && std::rename(tmp.c_str(), out_path.c_str()) == 0;
    lacx_free(lac);
    lacx_encoder_destroy(enc);
    if (!ok) {
        std::remove(tmp.c_str());
        std::cerr << "Failed to write LAC file: " << out_path << "\n";
        return 1;
    }
    std::cout << "Encoded " << in_path << " -> " << out_path << " (" << size << " bytes)\n";
    return 0;
}
