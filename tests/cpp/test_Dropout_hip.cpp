// test_Dropout_hip.cpp -- RisiContraction_18_dropout_hip through the executor, against the fp64 oracle.
// Known answer: after srand(2024) the REAL RisiContraction_18_dropout::forward (RisiContraction_18_dropout.h:113-125)
// with nKept = 7 keeps slices {2, 6, 12, 14, 15, 16, 17} (captured by tests/golden/make_golden.py, fixture
// drop_train_s2024_k7); the drop-in class must draw the same mask from the same seed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gf_executor.h"

extern "C" {
int gfo_r18_dropout_forward(const int *use, int train, int nKept, const double *P, const double *A, double *Out, int N, int C);
int gfo_r18_dropout_backward(const int *use, const double *G, const double *A, double *dP, int N, int C);
}

static int check(const char *what, const std::vector<double> &got, const std::vector<double> &ref) {
    double scale = 1, mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) scale = std::fmax(scale, std::fabs(ref[i]));
    for (size_t i = 0; i < ref.size(); ++i) mx = std::fmax(mx, std::fabs(got[i] - ref[i]) / scale);
    std::printf("%-40s max rel %.3e %s\n", what, mx, mx <= 1e-5 ? "" : "  <-- FAIL");
    return mx > 1e-5;
}

int main() {
    const int N = 6, C = 5, nKept = 7;
    srand(11);
    std::vector<Tensor3D *> t(N);
    std::vector<double> P((size_t)N * N * N * C), A((size_t)N * N), d0(P.size());
    const size_t per = (size_t)N * N * C;
    for (int a = 0; a < N; ++a) {
        t[a] = new Tensor3D(N, N, C);
        for (size_t i = 0; i < per; ++i) {
            t[a]->value[i] = P[a * per + i] = (rand() % 200 - 100) / 64.0;
            t[a]->gradient[i] = d0[a * per + i] = rand() % 4;
        }
    }
    Matrix adj(N, N);
    for (int i = 0; i < N * N; ++i) adj.value[i] = A[i] = (rand() % 3 == 0) ? 0.0 : (rand() % 5) / 2.0;
    RisiContraction_18_dropout_hip op(N, C);
    op.setParameter(N, C);
    for (int a = 0; a < N; ++a) op.add_tensor(t[a]);
    op.set_adjacency(&adj);
    op.setContractions(nKept);
    op.setTrainMode();
    GraphFlowExec graph;
    graph.add(&op, gftags::RISICONTRACTION_18_DROPOUT_HIP);

    srand(2024);
    graph.forward();
    static const int expect[18] = {0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 1, 1, 1, 1};
    int bad = 0, use[18];
    for (int k = 0; k < 18; ++k) {
        use[k] = op.use[k] ? 1 : 0;
        bad |= use[k] != expect[k];
    }
    std::printf("mask drawn after srand(2024): %s\n", bad ? "DIFFERS from the reference's  <-- FAIL" : "same as the reference's");
    std::vector<double> ref((size_t)N * N * 18 * C), got(ref.size()), G(ref.size());
    gfo_r18_dropout_forward(use, 1, nKept, &P[0], &A[0], &ref[0], N, C);
    for (size_t i = 0; i < got.size(); ++i) got[i] = op.value[i];
    bad |= check("dropout forward (train)", got, ref);
    for (size_t i = 0; i < G.size(); ++i) op.gradient[i] = G[i] = (rand() % 200 - 100) / 100.0;
    graph.backward();
    std::vector<double> dref = d0, dgot(d0.size());
    gfo_r18_dropout_backward(use, &G[0], &A[0], &dref[0], N, C);
    for (int a = 0; a < N; ++a)
        for (size_t i = 0; i < per; ++i) dgot[a * per + i] = t[a]->gradient[i];
    bad |= check("dropout backward (train, +=)", dgot, dref);

    op.setTestMode();
    graph.forward();
    for (int k = 0; k < 18; ++k) use[k] = 1;
    gfo_r18_dropout_forward(use, 0, nKept, &P[0], &A[0], &ref[0], N, C);
    for (size_t i = 0; i < got.size(); ++i) got[i] = op.value[i];
    bad |= check("dropout forward (test, x nKept/18)", got, ref);
    for (int a = 0; a < N; ++a) delete t[a];
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
