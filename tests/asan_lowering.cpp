// AddressSanitizer driver of tests/test_host_sanitizers.py: parse_tflite + build_plan (the host-only lowering, fuse levels 5 and 2) of every blob
// file given on the command line; every blob must give a plan or an exception (mi_*_create_from_bytes takes untrusted bytes).
#include <cstdio>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <vector>
#include "plan.hpp"
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; i++) {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<unsigned char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        for (int lvl : {5, 2}) {
            try { std::string s = mi::build_plan(mi::parse_tflite(b.data(), b.size()), lvl).describe(); ok++; }
            catch (const std::exception&) { bad++; }
        }
    }
    std::printf("ok %d refused %d\n", ok, bad);
    return 0;
}
