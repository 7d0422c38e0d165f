// fuzz_host.cpp — test harness (built by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined, CPU only):
// feeds the host-side parsers that accept untrusted bytes (JPEG markers + Huffman decoding, the TFL3 flatbuffer reader with
// its DENSIFY walk) the shipped files, byte-mutated copies of them and hand-built malformed inputs.  Every input must end in
// a result or a C++ exception; the sanitizers turn any out-of-bounds access into a crash.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../rs-face-detection-tflite_amd/csrc/jpeg.hpp"
#include "../rs-face-detection-tflite_amd/csrc/tflite_graph.hpp"

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

int main(int argc, char** argv) {
    long ok = 0, refused = 0;
    for (int a = 1; a < argc; a++) {
        const std::string path = argv[a];
        const bool is_jpeg = path.size() > 4 && path.substr(path.size() - 4) == ".jpg";
        const std::vector<uint8_t> orig = slurp(argv[a]);
        if (orig.empty()) { std::fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        const int rounds = is_jpeg ? 1500 : 150;
        for (int r = 0; r <= rounds; r++) {
            std::vector<uint8_t> b = orig;
            if (r > 0) {
                const int edits = 1 + static_cast<int>(rnd() % 6);
                for (int e = 0; e < edits; e++) {
                    // JPEG: favour the header region (tables, frame, scan headers); tflite: anywhere (offsets, vtables, metadata)
                    const size_t span = is_jpeg && (rnd() & 1) ? std::min<size_t>(b.size(), 700) : b.size();
                    b[rnd() % span] = static_cast<uint8_t>(rnd());
                }
                if (rnd() % 8 == 0) b.resize(rnd() % b.size() + 1);
            }
            try {
                if (is_jpeg) {
                    int w = 0, h = 0;
                    mi::jpeg_parse_size(b.data(), b.size(), &w, &h);
                    mi::JpegFrame f;
                    mi::jpeg_entropy_decode(b.data(), b.size(), &f);
                } else {
                    mi::Graph g = mi::parse_tflite(b.data(), b.size());
                    (void)g;
                }
                ok++;
            } catch (const std::exception&) {
                refused++;
            }
        }
    }
    std::printf("ok %ld refused %ld\n", ok, refused);
    return 0;
}
