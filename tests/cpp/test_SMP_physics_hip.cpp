// test_SMP_physics_hip.cpp -- the `_physics` / `_pairgraphs` model drop-ins, driven like the reference's
// tests/test_SMP_omega_physics.cpp and tests/test_SMP_omega_pairgraphs.cpp: the four hand-built molecules (CH4, NH3, H2O, C2H4;
// one-hot C,H,N,O features), nLevels 2, nChanels 16, max_receptive_field 4, max_nVertices 10; physics target = number of atoms,
// pairgraphs = all 16 ordered pairs with target = difference of the atom counts (test_SMP_omega_pairgraphs.cpp:176-186).
// Known answers: the REAL classes, constructed after srand(7), report for three BatchLearn(..., 1e-3) calls the (before, after)
// losses below (tests/golden/make_golden.py -> smp_physics.npz: trainphys / trainpair).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "SMP_physics_hip.h"

struct Molecule {  // public fields of GraphFlow/DenseGraph.h
    int nVertices, nFeatures;
    int **adj;
    double **feature;
    Molecule(int V, int F) : nVertices(V), nFeatures(F) {
        adj = new int *[V];
        feature = new double *[V];
        for (int i = 0; i < V; ++i) {
            adj[i] = new int[V]();
            feature[i] = new double[F]();
        }
    }
};

static Molecule *build(const char *labels, int nEdges, const int (*edges)[2]) {
    const int V = (int)std::strlen(labels);
    Molecule *m = new Molecule(V, 4);
    for (int e = 0; e < nEdges; ++e) m->adj[edges[e][0]][edges[e][1]] = m->adj[edges[e][1]][edges[e][0]] = 1;
    for (int v = 0; v < V; ++v) m->feature[v][std::strchr("CHNO", labels[v]) - "CHNO"] = 1.0;
    return m;
}

static int close_to(const char *what, double got, double ref, double tol) {
    const double rel = std::fabs(got - ref) / std::fmax(1.0, std::fabs(ref));
    std::printf("%-44s %-14.8f reference %-14.8f rel %.2e %s\n", what, got, ref, rel, rel <= tol ? "" : "  <-- FAIL");
    return rel > tol;
}

int main() {
    static const int e1[][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}}, e2[][2] = {{0, 1}, {0, 2}, {0, 3}}, e3[][2] = {{0, 1}, {0, 2}},
                     e4[][2] = {{0, 1}, {0, 2}, {0, 3}, {3, 4}, {3, 5}};
    Molecule *mol[4] = {build("CHHHH", 4, e1), build("NHHH", 3, e2), build("OHH", 2, e3), build("CHHCHH", 5, e4)};
    double target[4];
    for (int i = 0; i < 4; ++i) target[i] = mol[i]->nVertices;
    int bad = 0;
    char name[96];
    {   // SMP_omega_physics
        srand(7);
        SMP_omega_physics_hip net(10, 4, 2, 16, 4);
        net.init_multi_threads(8);
        static const double ref[3][2] = {{43.00789267, 42.69926275}, {42.69926275, 41.85516785}, {41.85516785, 39.44387171}};
        for (int it = 0; it < 3; ++it) {
            std::pair<double, double> r = net.BatchLearn(4, mol, target, 1e-3);
            std::snprintf(name, sizeof name, "SMP_omega_physics BatchLearn %d before", it);
            bad |= close_to(name, r.first, ref[it][0], 1e-5);
            std::snprintf(name, sizeof name, "SMP_omega_physics BatchLearn %d after", it);
            bad |= close_to(name, r.second, ref[it][1], 5e-5);
        }
        double y[4], loss = 0.0;
        net.Threaded_Predict(4, mol, y);
        for (int i = 0; i < 4; ++i) {
            bad |= close_to("Predict == Threaded_Predict", net.Predict(mol[i]), y[i], 1e-6);
            loss += 0.5 * (y[i] - target[i]) * (y[i] - target[i]);
        }
        bad |= close_to("getLoss == sum 0.5 (y-t)^2", net.getLoss(4, mol, target), loss, 1e-5);
        const char *path = "/tmp/gf_smp_physics_hip_ckpt.txt";
        net.save_model(path);
        srand(99);
        SMP_omega_physics_hip other(10, 4, 2, 16, 4);
        other.load_model(path);
        bad |= close_to("loaded model predicts alike", other.Predict(mol[3]), y[3], 1e-4);  // 6 printed digits
        srand(5);
        SMP_beta_physics_hip beta(10, 2, 16, 4);
        srand(5);
        SMP_omega_physics_hip uncapped(10, 10, 2, 16, 4);
        for (int i = 0; i < 4; ++i) bad |= close_to("SMP_beta_physics == uncapped SMP_omega_physics", beta.Predict(mol[i]), uncapped.Predict(mol[i]), 0.0);
        // "does it learn" (tests/test_SMP_omega_physics.cpp): Threaded_BatchLearn epochs at 1e-3
        double first = net.getLoss(4, mol, target);
        for (int epoch = 0; epoch < 400; ++epoch) net.Threaded_BatchLearn(4, mol, target, 1e-3);
        const double last = net.getLoss(4, mol, target);
        std::printf("SMP_omega_physics loss %.4f -> %.6f after 400 epochs\n", first, last);
        bad |= !(last < 0.05 * first);
    }
    {   // SMP_omega_pairgraphs on all ordered pairs
        Molecule *g1[16], *g2[16];
        double t[16];
        for (int i = 0, c = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j, ++c) {
                g1[c] = mol[i];
                g2[c] = mol[j];
                t[c] = target[i] - target[j];
            }
        srand(7);
        SMP_omega_pairgraphs_hip net(10, 10, 4, 2, 16, 4, 4);
        static const double ref[3][2] = {{19.99999626, 19.99888786}, {19.99888786, 19.97672295}, {19.97672295, 19.89087374}};
        for (int it = 0; it < 3; ++it) {
            std::pair<double, double> r = net.BatchLearn(16, g1, g2, t, 1e-3);
            std::snprintf(name, sizeof name, "SMP_omega_pairgraphs BatchLearn %d before", it);
            bad |= close_to(name, r.first, ref[it][0], 1e-5);
            std::snprintf(name, sizeof name, "SMP_omega_pairgraphs BatchLearn %d after", it);
            bad |= close_to(name, r.second, ref[it][1], 5e-5);
        }
        double y[16];
        net.Threaded_Predict(16, g1, g2, y);
        bad |= close_to("pair Predict == Threaded_Predict", net.Predict(g1[7], g2[7]), y[7], 1e-6);
        SMP_beta_pairgraphs_hip beta(10, 10, 2, 16, 4, 4);
        bad |= !(beta.Predict(mol[0], mol[3]) == beta.Predict(mol[0], mol[3]));
        // SMP_sigma_pairgraphs: train mode draws slice masks (different losses from call to call), test mode is deterministic
        srand(3);
        SMP_sigma_pairgraphs_hip sigma(10, 10, 4, 2, 16, 4, 4, 9);
        const double l1 = sigma.getLoss(16, g1, g2, t), l2 = sigma.getLoss(16, g1, g2, t);
        sigma.setTestMode();
        const double l3 = sigma.getLoss(16, g1, g2, t), l4 = sigma.getLoss(16, g1, g2, t);
        std::printf("SMP_sigma_pairgraphs train-mode losses %.6f %.6f, test-mode %.6f %.6f\n", l1, l2, l3, l4);
        bad |= (l1 == l2) || (l3 != l4);
        sigma.setTrainMode();
        for (int epoch = 0; epoch < 3; ++epoch) sigma.Threaded_BatchLearn(16, g1, g2, t, 1e-3);
    }
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
