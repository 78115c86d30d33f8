// prep_sanitize.cpp -- the host graph preparation (smp_prep.cpp) alone, for -fsanitize=address,undefined and
// -fsanitize=thread runs on a CPU-only box (tests/test_prep_sanitizers.py).  Random molecule-like graphs (a spanning
// tree plus a few extra bonds), several batches through ONE BatchLayout so the vector-reuse path is exercised, and a
// thread-count independence check: 1 thread and many threads must produce identical tables.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "smp_prep.h"

static void make_batch(unsigned seed, int nMol, int F, std::vector<int> *nV, std::vector<int> *adj, std::vector<double> *feat) {
    srand(seed);
    nV->clear();
    adj->clear();
    feat->clear();
    for (int m = 0; m < nMol; ++m) {
        const int V = 1 + rand() % 24;
        std::vector<int> a((size_t)V * V, 0);
        for (int v = 1; v < V; ++v) {
            const int u = rand() % v;
            a[u * V + v] = a[v * V + u] = 1;
        }
        for (int e = 0; e < V / 6; ++e) {
            const int u = rand() % V, v = rand() % V;
            if (u != v) a[u * V + v] = a[v * V + u] = 1;
        }
        nV->push_back(V);
        adj->insert(adj->end(), a.begin(), a.end());
        for (int v = 0; v < V; ++v)
            for (int f = 0; f < F; ++f) feat->push_back(f == rand() % F ? 1.0 : 0.0);
    }
}

int main() {
    gfsmp::Config cfg = {3, 8, 5, 3, 10, 1};
    gfsmp::BatchLayout reused, single;
    std::vector<int> nV, adj;
    std::vector<double> feat;
    int bad = 0;
    for (unsigned round = 0; round < 4; ++round) {
        make_batch(100 + round, 96 + 40 * (round % 2), cfg.nFeatures, &nV, &adj, &feat);
        unsetenv("GF_PREP_THREADS");
        gfsmp::build_batch(cfg, (int)nV.size(), &nV[0], &adj[0], &feat[0], NULL, &reused);
        setenv("GF_PREP_THREADS", "1", 1);
        gfsmp::build_batch(cfg, (int)nV.size(), &nV[0], &adj[0], &feat[0], NULL, &single);
        for (int l = 1; l <= cfg.nLevels; ++l) {
            const gfsmp::LevelLayout &a = reused.level[l], &b = single.level[l];
            bad |= a.rows != b.rows || a.pairs != b.pairs || a.adj != b.adj || a.pi != b.pi || a.inv != b.inv ||
                   a.cons_ptr != b.cons_ptr || a.cons_slab != b.cons_slab || a.rowscale != b.rowscale || a.rsum != b.rsum;
        }
        bad |= reused.x != single.x;
        std::printf("round %u: %zu molecules, level-3 rows %lld %s\n", round, nV.size(), (long long)reused.level[3].rows,
                    bad ? "MISMATCH" : "ok");
    }
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
