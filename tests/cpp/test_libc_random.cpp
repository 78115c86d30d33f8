// CPU test: gf::LibcRandom steps glibc's rand() stream inline and hands it back in step (graphflow_amd/csrc/libc_random.h).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>
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
    {   // two threads borrowing at once (one model per worker thread, SMP_sigma_pairgraphs::Threaded_ComputeGradient): the borrow is
        // serialised, libc ends up on ITS OWN array again (not on a dead thread's stack, not on the parking buffer), and the 2 x 200 x 50
        // draws the threads made are exactly the next 20,000 values of the stream, whoever drew them
        srand(77u);
        std::vector<int> ref(20000 + 16);
        for (int &v : ref) v = rand();
        srand(77u);
        char probe_buf[128];
        char *own = initstate(1u, probe_buf, sizeof probe_buf);   // own = libc's static array
        (void)setstate(own);
        srand(77u);
        std::vector<int> drawn[2];
        auto worker = [&](int t) {
            for (int rep = 0; rep < 200; ++rep) {
                gf::LibcRandom g(true);
                for (int i = 0; i < 50; ++i) drawn[t].push_back(g.next());
            }
        };
        std::thread a(worker, 0), b(worker, 1);
        a.join();
        b.join();
        char *now = initstate(1u, probe_buf, sizeof probe_buf);
        CHECK(now == own);                                             // the state pointer afterwards
        (void)setstate(now);
        std::vector<int> all(drawn[0]);
        all.insert(all.end(), drawn[1].begin(), drawn[1].end());
        std::vector<int> want(ref.begin(), ref.begin() + 20000);
        std::sort(all.begin(), all.end());
        std::sort(want.begin(), want.end());
        CHECK(all == want);
        for (int i = 0; i < 16; ++i) CHECK(rand() == ref[20000 + i]);  // and libc continues in step
    }
    std::printf(fails ? "test_libc_random: %d failure(s)\n" : "test_libc_random: ok\n", fails);
    return fails ? 1 : 0;
}
