// lacx_cli -- the `encode` command of the reference's command-line tool on the MI355X path (SURVEY row f-4;
// ref src/main.cpp:609-710): same positional arguments, every flag of the reference's encode command with the same
// meaning and rejection rules (--stereo-mode=lr|ms, --no-partitioning, --threads=N, the --debug-* family), LAC_THREADS
// resolved by the tool and not by the library (ref :586-591), the same-file check on the resolved paths (ref :433-444),
// the same messages, staged output (written next to the target, renamed on success).  The WAV file goes through
// lacx_wav_parse / lacx_encode_wav_view: the raw data chunk is what crosses PCIe, the .lac is written to the file
// straight from the encoder's pinned result buffer.  Decode and selftest stay with the reference's tool.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "lacx.h"

namespace {

void usage() {
    std::cerr << "Usage:\n  lacx_cli encode input.wav output.lac [--stereo-mode=lr|ms] [--threads=N] [--debug-threads] [--debug-lpc] "
                 "[--debug-stereo-est] [--debug-zr] [--debug-partitions] [--no-partitioning]\n";
}

// A positive decimal integer and nothing else (ref src/main.cpp:560-584, src/codec/lac/thread_limit.hpp:10-28).
bool positive_integer(const std::string& v, unsigned long long& out) {
    if (v.empty() || v.size() > 18) return false;
    for (char c : v)
        if (c < '0' || c > '9') return false;
    out = std::stoull(v);
    return out != 0;
}

// Same target, however the two paths are spelled: an existing file reached twice (hard link, symlink, ./ prefixes), or
// two spellings that normalise to one path when the output does not exist yet (ref src/main.cpp:433-444).
bool same_file(const std::string& a, const std::string& b) {
    std::error_code ec;
    if (std::filesystem::equivalent(a, b, ec)) return true;
    ec.clear();
    const std::filesystem::path na = std::filesystem::weakly_canonical(a, ec);
    if (ec) return false;
    const std::filesystem::path nb = std::filesystem::weakly_canonical(b, ec);
    return !ec && na == nb;
}

struct Encoded {
    const uint8_t* data = nullptr;
    uint64_t size = 0;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4 || std::string(argv[1]) != "encode") {
        usage();
        return 1;
    }
    const std::string in_path = argv[2], out_path = argv[3];
    if (same_file(in_path, out_path)) {
        std::cerr << "Input and output paths must be different\n";
        return 1;
    }
    uint8_t stereo_mode = 2;
    bool partitioning = true, debug_threads = false, debug_zr = false;
    unsigned long long threads = 0;
    for (int i = 4; i < argc; ++i) {
        const std::string flag = argv[i];
        const std::string tprefix = "--threads=";
        if (flag == "--no-partitioning") {
            partitioning = false;
        } else if (flag == "--stereo-mode=lr") {
            stereo_mode = 0;
        } else if (flag == "--stereo-mode=ms") {
            stereo_mode = 1;
        } else if (flag == "--debug-threads") {
            debug_threads = true;
        } else if (flag == "--debug-zr") {
            debug_zr = true;
        } else if (flag == "--debug-lpc" || flag == "--debug-stereo-est" || flag == "--debug-partitions") {
            // accepted like the reference does; its per-block log lines only exist in debug builds (LAC_DEBUG_LOG)
        } else if (flag.compare(0, tprefix.size(), tprefix) == 0) {
            if (!positive_integer(flag.substr(tprefix.size()), threads)) {
                std::cerr << "Error: --threads requires a positive integer\n";
                return 1;
            }
        } else {
            usage();
            return 1;
        }
    }
    if (threads == 0) {  // --threads wins, else LAC_THREADS (ref src/main.cpp:586-591)
        const char* env = std::getenv("LAC_THREADS");
        if (env && *env) {
            const std::string v = env;
            bool digits = true;
            for (char c : v) digits = digits && c >= '0' && c <= '9';
            if (digits && v.size() > 18) {
                std::cerr << "Error: LAC_THREADS is too large\n";
                return 1;
            }
            if (!positive_integer(v, threads)) {
                std::cerr << "Error: LAC_THREADS must be a positive integer\n";
                return 1;
            }
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
    cfg.device = LACX_DEVICE_ALL;  // every visible device: the blocks fan out over them (the reference spreads them over its threads)
    cfg.emit_threads = (uint32_t)(threads > 0xFFFFFFFFull ? 0xFFFFFFFFull : threads);
    lacx_encoder* enc = nullptr;
    if (lacx_encoder_create(&cfg, &enc) != LACX_OK) {
        std::cerr << "Error: lacx_encoder_create failed\n";
        return 1;
    }
    Encoded lac;
    if (lacx_encode_wav_view(enc, wav.data(), wav.size(), &lac.data, &lac.size) != LACX_OK) {
        std::cerr << "Error: " << lacx_last_error(enc) << "\n";
        lacx_encoder_destroy(enc);
        return 1;
    }
    if (debug_zr) {  // the same stream without the zero-run mode, for the gain line (ref src/main.cpp:677-689)
        lacx_config base_cfg = cfg;
        base_cfg.zero_run_enabled = 0;
        lacx_encoder* base = nullptr;
        Encoded b;
        if (lacx_encoder_create(&base_cfg, &base) != LACX_OK ||
            lacx_encode_wav_view(base, wav.data(), wav.size(), &b.data, &b.size) != LACX_OK) {
            std::cerr << "Error: " << (base ? lacx_last_error(base) : "lacx_encoder_create failed") << "\n";
            if (base) lacx_encoder_destroy(base);
            lacx_encoder_destroy(enc);
            return 1;
        }
        const double gain = b.size ? (1.0 - (double)lac.size / (double)b.size) * 100.0 : 0.0;
        std::cout << "[debug-zr] baseline_bytes=" << b.size << " zr_bytes=" << lac.size << " gain=" << gain << "%\n";
        lacx_encoder_destroy(base);
    }
    const std::string tmp = out_path + ".lacx-partial";
    bool ok = false;
    {
        std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
        ok = out && out.write(reinterpret_cast<const char*>(lac.data), (std::streamsize)lac.size) && out.flush();
    }
    ok = ok && !same_file(in_path, out_path) && std::rename(tmp.c_str(), out_path.c_str()) == 0;
    const uint64_t size = lac.size;
    lacx_encoder_destroy(enc);  // the view dies with the encoder
    if (!ok) {
        std::remove(tmp.c_str());
        std::cerr << "Failed to write LAC file: " << out_path << "\n";
        return 1;
    }
    std::cout << "Encoded " << in_path << " -> " << out_path << " (" << size << " bytes)\n";
    if (debug_threads) {
        // The block loop runs on the device; on the host the encode call uses the calling thread (the emit pool only
        // exists with LACX_FLAG_HOST_EMIT), which is what the reference reports for a one-thread run (ref :699-709).
        std::cout << "Thread usage: 1 threads\n  " << std::this_thread::get_id() << "\n";
        std::cout << "WARNING: Multi-threading not active (single-threaded execution).\n";
    }
    return 0;
}
