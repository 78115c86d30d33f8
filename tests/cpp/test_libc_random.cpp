// CPU test: gf::LibcRandom steps glibc's rand() stream inline and hands it back in step (graphflow_amd/csrc/libc_random.h).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../graphflow_amd/csrc/libc_random.h"

static int fails = 0;
#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);   \
            ++fails;                                                  \
        }                                                             \
    } while (0)

int main() {
    for (unsigned seed : {1u, 5u, 7u, 123456789u}) {
        srand(seed);
        std::vector<int> ref(5000);
        for (int &v : ref) v = rand();
        srand(seed);
        std::vector<int> got;
        for (int i = 0; i < 100; ++i) got.push_back(rand());           // plain calls first
        {
            gf::LibcRandom g(true);
            CHECK(g.fast);                                             // glibc's default generator: the inline path is taken
            for (int i = 0; i < 3000; ++i) got.push_back(g.next());    // borrowed (crosses the 31-word wrap many times)
        }
        for (int i = 0; i < 900; ++i) got.push_back(rand());           // libc continues where the borrower stopped
        {
            gf::LibcRandom g(true);
            for (int i = 0; i < 1000; ++i) got.push_back(g.next());
        }
        CHECK(got.size() == ref.size());
        for (size_t i = 0; i < ref.size(); ++i)
            if (got[i] != ref[i]) {
                std::printf("seed %u: draw %zu differs (%d vs %d)\n", seed, i, got[i], ref[i]);
                ++fails;
                break;
            }
    }
    {   // a caller on another generator type (initstate with 8 bytes = TYPE_0) is left alone: every draw is a plain rand()
        static char small[8];
        char *prev = initstate(3u, small, sizeof small);
        const int a = rand();
        (void)initstate(3u, small, sizeof small);
        gf::LibcRandom g(true);
        CHECK(!g.fast);
        CHECK(g.next() == a);
        (void)setstate(prev);
    }
    std::printf(fails ? "test_libc_random: %d failure(s)\n" : "test_libc_random: ok\n", fails);
    return fails ? 1 : 0;
}
