// fast_inflate.h against zlib (tests/test_host_api.py::test_fast_inflate_fuzz, built with ASan + UBSan): random data of five
// kinds deflated at every level and strategy (stored, fixed and dynamic blocks, flushes in the middle), round trip must be exact;
// single-bit corruptions, truncation and wrong output sizes must never be accepted with wrong bytes.  usage: fast_inflate_fuzz [trials]
#include "fast_inflate.h"
#include <zlib.h>
#include <vector>
#include <random>
#include <cstdio>
#include <cstdlib>
#include <chrono>
using namespace lzb_vio;
int main(int argc, char** argv) {
    std::mt19937 rng(123);
    finf::Tables* T = new finf::Tables;
    int ok = 0, total = 0, corrupt_rejected = 0, corrupt_total = 0;
    const int trials = argc > 1 ? atoi(argv[1]) : 3000;
    for (int trial = 0; trial < trials; trial++) {
        size_t n = (trial % 7 == 0) ? rng() % 64 + 1 : rng() % 300000 + 1;
        std::vector<uint8_t> src(n);
        int kind = rng() % 5;
        uint8_t prev = 0;
        for (size_t i = 0; i < n; i++) {
            switch (kind) {
            case 0: src[i] = (uint8_t)rng(); break;                                  // incompressible
            case 1: src[i] = (uint8_t)((rng() % 100) < 90 ? prev : rng()); break;    // runs
            case 2: src[i] = (uint8_t)(rng() % 4); break;                            // small alphabet
            case 3: src[i] = (uint8_t)(i % 251 + (rng() % 3)); break;                // periodic
            default: src[i] = (uint8_t)(128 + (int)(rng() % 17) - 8); break;         // filtered-image-like
            }
            prev = src[i];
        }
        int level = trial % 10, strat = (trial / 10) % 5;
        z_stream zs; memset(&zs, 0, sizeof zs);
        static const int strats[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        deflateInit2(&zs, level, Z_DEFLATED, trial % 8 == 0 ? 9 + trial % 7 : 15, 8, strats[strat]);      // small windows too
        std::vector<uint8_t> comp(deflateBound(&zs, n) + 32);
        zs.next_in = src.data(); zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        // several flushes in the middle: more blocks, empty stored blocks
        if (trial % 3 == 0 && n > 100) { zs.avail_in = (uInt)(n / 3); deflate(&zs, Z_FULL_FLUSH); zs.avail_in = (uInt)(n - n / 3); }
        deflate(&zs, Z_FINISH);
        size_t cn = zs.total_out; deflateEnd(&zs);
        comp.resize(cn + 16, 0);
        std::vector<uint8_t> out(n + finf::kSlack);
        bool r = finf::inflate_zlib(comp.data(), cn, out.data(), n, *T);
        total++;
        if (r && memcmp(out.data(), src.data(), n) == 0) ok++;
        else printf("MISMATCH trial %d n %zu level %d strat %d r %d\n", trial, n, level, strat, (int)r);
        // corruptions: flip a byte; must not report success unless the output still equals the source (then the flip was in slack bits)
        if (cn > 8) {
            for (int c = 0; c < 4; c++) {
                std::vector<uint8_t> bad = comp;
                size_t pos = rng() % cn; bad[pos] ^= (uint8_t)(1u << (rng() % 8));
                std::vector<uint8_t> o2(n + finf::kSlack);
                bool r2 = finf::inflate_zlib(bad.data(), cn, o2.data(), n, *T);
                corrupt_total++;
                if (!r2) corrupt_rejected++;
                else if (memcmp(o2.data(), src.data(), n) != 0) printf("ACCEPTED CORRUPT trial %d pos %zu\n", trial, pos);
            }
            // truncated and wrong-size requests
            std::vector<uint8_t> o3(n + 8 + finf::kSlack);
            if (finf::inflate_zlib(comp.data(), cn - 1, o3.data(), n, *T)) printf("ACCEPTED TRUNCATED %d\n", trial);
            if (finf::inflate_zlib(comp.data(), cn, o3.data(), n + 1, *T)) printf("ACCEPTED LONGER %d\n", trial);
            if (n > 1 && finf::inflate_zlib(comp.data(), cn, o3.data(), n - 1, *T)) printf("ACCEPTED SHORTER %d\n", trial);
        }
    }
    printf("round trips ok %d / %d; corrupted streams rejected %d / %d (the rest decoded to the right bytes)\n", ok, total, corrupt_rejected, corrupt_total);
    delete T;
    return ok == total ? 0 : 1;
}
