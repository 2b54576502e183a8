// ASan/UBSan driver for the blob parser (host only): reads blobs from files given on the command line
#include <cstdio>
#include <vector>
#include "hg_common.hpp"
int main(int argc, char** argv) {
    int ok = 0, rej = 0;
    for (int i = 1; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) continue;
        std::vector<unsigned char> b;
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f);
        try {
            auto t = hg::parse_blob(b.data(), b.size());
            (void)hg::tree_flops(*t);
            ++ok;
        } catch (const hg::Error&) { ++rej; }
    }
    printf("parsed %d rejected %d\n", ok, rej);
    return 0;
}
