// test_Mixers_hip.cpp -- Entity-style mixers through the tag-dispatched executor, against the fp64 oracle.
// Part 1 follows the reference's tests/test_MatMul_gpu.cu (A = 1600 x 720, B = 720 x 40, integer entries,
// srand(123456789), forward + backward, sum of absolute errors) but FAILS above 1e-5 relative error.
// Part 2 runs the promotion chain of one SMP vertex -- MatTensorMul -> TensorMatMul -> StackTensor3D ->
// RisiContraction_18 -> MatMul (K-projection) -- as ONE graph: forward() in insertion order, backward() in reverse,
// every gradient checked against the same chain evaluated with the oracle.
#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gf_executor.h"

extern "C" {
void gfo_matmul_forward(const double *A, const double *B, double *C, int M, int K, int N);
void gfo_matmul_backward(const double *dC, const double *A, const double *B, double *dA, double *dB, int M, int K, int N);
void gfo_mattensormul_forward(const double *X, const double *F, double *Out, int R, int Kd, int J, int D);
void gfo_mattensormul_backward(const double *G, const double *X, const double *F, double *dX, double *dF, int R, int Kd, int J, int D);
void gfo_tensormatmul_forward(const double *F, const double *Y, double *Out, int R, int Kd, int J, int D);
void gfo_tensormatmul_backward(const double *G, const double *F, const double *Y, double *dF, double *dY, int R, int Kd, int J, int D);
void gfo_custommatmultensor_forward(const double *W, const double *T, double *Out, int rows, int V, int Kout);
void gfo_custommatmultensor_backward(const double *G, const double *W, const double *T, double *dW, double *dT, int rows, int V, int Kout);
int gfo_contract_forward(int K, const double *P, const double *A, double *Out, int N, int C);
int gfo_contract_backward(int K, const double *G, const double *A, double *dP, int N, int C);
}

static double now_ms() {
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return tp.tv_sec * 1e3 + tp.tv_usec * 1e-3;
}

template <class V>
static std::vector<double> vals(const V *v, bool grad = false) {
    std::vector<double> o(v->size);
    for (int i = 0; i < v->size; ++i) o[i] = grad ? v->gradient[i] : v->value[i];
    return o;
}

static int check(const char *what, const std::vector<double> &got, const std::vector<double> &ref) {
    double scale = 1, err = 0, mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) scale = std::fmax(scale, std::fabs(ref[i]));
    for (size_t i = 0; i < ref.size(); ++i) {
        const double d = std::fabs(got[i] - ref[i]);
        err += d;
        mx = std::fmax(mx, d / scale);
    }
    std::printf("%-34s absolute error %-12g max rel %.3e %s\n", what, err, mx, mx <= 1e-5 ? "" : "  <-- FAIL");
    return mx > 1e-5 || got.size() != ref.size();
}

static int part1_matmul() {
    const int M = 1600, K = 720, N = 40;
    srand(123456789);
    Matrix A(M, K), B(K, N);
    for (int i = 0; i < A.size; ++i) A.value[i] = rand() % 100;
    for (int i = 0; i < B.size; ++i) B.value[i] = rand() % 100;
    for (int i = 0; i < A.size; ++i) A.gradient[i] = rand() % 5;  // pins `+=`
    for (int i = 0; i < B.size; ++i) B.gradient[i] = rand() % 5;
    MatMul_hip obj(&A, &B);
    double t0 = now_ms();
    obj.forward();
    std::printf("GPU forward time: %.2f ms\n", now_ms() - t0);
    std::vector<double> a = vals(&A), b = vals(&B), ref((size_t)M * N);
    t0 = now_ms();
    gfo_matmul_forward(&a[0], &b[0], &ref[0], M, K, N);
    std::printf("CPU forward time: %.2f ms\n", now_ms() - t0);
    int bad = check("MatMul forward", vals(&obj), ref);
    std::vector<double> g((size_t)M * N), da = vals(&A, true), db = vals(&B, true);
    for (int i = 0; i < obj.size; ++i) {
        bad |= obj.gradient[i] != 0;
        obj.gradient[i] = g[i] = rand() % 100;
    }
    t0 = now_ms();
    obj.backward();
    std::printf("GPU backward time: %.2f ms\n", now_ms() - t0);
    t0 = now_ms();
    gfo_matmul_backward(&g[0], &a[0], &b[0], &da[0], &db[0], M, K, N);
    std::printf("CPU backward time: %.2f ms\n", now_ms() - t0);
    bad |= check("MatMul backward (first)", vals(&A, true), da);
    bad |= check("MatMul backward (second)", vals(&B, true), db);
    return bad;
}

// one SMP vertex at level l: s = |phi_l(v)| neighbours, each with a previous-level tensor of size sw x sw x C
static int part2_vertex_chain() {
    const int s = 6, sw = 4, C = 8;
    srand(7);
    GraphFlowExec graph;
    std::vector<Tensor3D *> f_prev(s);
    std::vector<Matrix *> X(s), XT(s);
    std::vector<MatTensorMul_hip *> left(s);
    std::vector<TensorMatMul_hip *> quad(s);
    Matrix adj(s, s), Kw(18 * C, C);
    for (int i = 0; i < adj.size; ++i) adj.value[i] = (rand() % 3) ? 1.0 : 0.0;
    for (int i = 0; i < Kw.size; ++i) Kw.value[i] = (rand() % 200 - 100) / 100.0;
    graph.add(&adj, gftags::MATRIX);
    graph.add(&Kw, gftags::MATRIX);
    for (int w = 0; w < s; ++w) {
        f_prev[w] = new Tensor3D(sw, sw, C);
        for (int i = 0; i < f_prev[w]->size; ++i) f_prev[w]->value[i] = (rand() % 200 - 100) / 100.0;
        X[w] = new Matrix(s, sw);  // 0/1 selection matrix, as SMP_omega.h:461-474 builds it
        XT[w] = new Matrix(sw, s);
        for (int i = 0; i < X[w]->size; ++i) X[w]->value[i] = 0;
        for (int k = 0; k < sw; ++k) X[w]->value[X[w]->index((k + w) % s, k)] = 1;
        for (int i = 0; i < s; ++i)
            for (int k = 0; k < sw; ++k) XT[w]->value[XT[w]->index(k, i)] = X[w]->value[X[w]->index(i, k)];
        graph.add(f_prev[w], gftags::TENSOR3D);
        graph.add(X[w], gftags::MATRIX);
        graph.add(XT[w], gftags::MATRIX);
    }
    RisiContraction_18_hip contract(s, C);
    contract.setParameter(s, C);
    for (int w = 0; w < s; ++w) {
        left[w] = new MatTensorMul_hip(X[w], f_prev[w]);
        quad[w] = new TensorMatMul_hip(left[w], XT[w]);
        graph.add(left[w], gftags::MATTENSORMUL_HIP);
        graph.add(quad[w], gftags::TENSORMATMUL_HIP);
        contract.add_tensor(quad[w]);
    }
    contract.set_adjacency(&adj);
    graph.add(&contract, gftags::RISICONTRACTION_18_HIP);
    // Reshape2D(s*s, 18C) is a pure relabelling of the same row-major buffer: view the contraction as a Matrix
    Matrix view(s * s, 18 * C);
    MatMul_hip project(s * s, C);
    // (the view shares no storage in this container model, so copy through a tiny identity op below)
    graph.forward();
    for (int i = 0; i < view.size; ++i) view.value[i] = contract.value[i];
    for (int i = 0; i < view.size; ++i) view.gradient[i] = 0;
    project.setParameter(&view, &Kw);
    project.forward();

    // oracle chain
    std::vector<double> P((size_t)s * s * s * C), A = vals(&adj), kw = vals(&Kw);
    std::vector<std::vector<double> > T1(s), x(s), xt(s), fp(s);
    for (int w = 0; w < s; ++w) {
        x[w] = vals(X[w]);
        xt[w] = vals(XT[w]);
        fp[w] = vals(f_prev[w]);
        T1[w].resize((size_t)s * sw * C);
        gfo_mattensormul_forward(&x[w][0], &fp[w][0], &T1[w][0], s, sw, sw, C);
        gfo_tensormatmul_forward(&T1[w][0], &xt[w][0], &P[(size_t)w * s * s * C], s, sw, s, C);
    }
    std::vector<double> Q((size_t)s * s * 18 * C), Y((size_t)s * s * C);
    gfo_contract_forward(18, &P[0], &A[0], &Q[0], s, C);
    gfo_matmul_forward(&Q[0], &kw[0], &Y[0], s * s, 18 * C, C);
    int bad = check("chain forward (K-projection out)", vals(&project), Y);

    // backward: seed dY, run project.backward() then the graph in reverse
    std::vector<double> dY((size_t)s * s * C), dQ(Q.size(), 0.0), dKw(kw.size(), 0.0), dPo(P.size(), 0.0);
    for (int i = 0; i < project.size; ++i) project.gradient[i] = dY[i] = (rand() % 200 - 100) / 100.0;
    project.backward();
    for (int i = 0; i < view.size; ++i) contract.gradient[i] = view.gradient[i];
    graph.backward();
    gfo_matmul_backward(&dY[0], &Q[0], &kw[0], &dQ[0], &dKw[0], s * s, 18 * C, C);
    gfo_contract_backward(18, &dQ[0], &A[0], &dPo[0], s, C);
    bad |= check("chain backward (K weights)", vals(&Kw, true), dKw);
    for (int w = 0; w < s; ++w) {
        std::vector<double> dT1(T1[w].size(), 0.0), dF(fp[w].size(), 0.0);
        gfo_tensormatmul_backward(&dPo[(size_t)w * s * s * C], &T1[w][0], &xt[w][0], &dT1[0], NULL, s, sw, s, C);
        gfo_mattensormul_backward(&dT1[0], &x[w][0], &fp[w][0], NULL, &dF[0], s, sw, sw, C);
        char name[64];
        std::snprintf(name, sizeof name, "chain backward (f_prev[%d])", w);
        bad |= check(name, vals(f_prev[w], true), dF);
    }
    // StackTensor3D_hip: forward copy is exact, backward adds
    StackTensor3D_hip stack(s, s, s, C);
    for (int w = 0; w < s; ++w) stack.add_tensor(quad[w]);
    stack.forward();
    for (size_t i = 0; i < P.size(); ++i) bad |= std::fabs(stack.value[i] - quad[i / ((size_t)s * s * C)]->value[i % ((size_t)s * s * C)]) != 0;
    std::printf("StackTensor3D_hip forward exact: %s\n", bad ? "NO" : "yes");
    for (int w = 0; w < s; ++w) {
        delete left[w]; delete quad[w]; delete X[w]; delete XT[w]; delete f_prev[w];
    }
    return bad;
}

// CustomMatMulTensor_hip as SMP_2D_ver6-8 use it: the K-projection applied straight to the contraction output
// viewed as a Tensor3D [N][N][18C] (no Reshape2D), through the executor.
static int part3_channel_mix() {
    const int N = 9, V = 18 * 12, Kout = 12;
    srand(99);
    Matrix W(Kout, V);
    Tensor3D T(N, N, V);
    for (int i = 0; i < W.size; ++i) W.value[i] = (rand() % 200 - 100) / 50.0;
    for (int i = 0; i < T.size; ++i) T.value[i] = (rand() % 200 - 100) / 50.0;
    CustomMatMulTensor_hip mix(&W, &T);
    GraphFlowExec graph;
    graph.add(&W, gftags::MATRIX);
    graph.add(&T, gftags::TENSOR3D);
    graph.add(&mix, gftags::CUSTOMMATMULTENSOR_HIP);
    graph.forward();  // a container's forward() zeroes its gradient (Matrix.h:58-62), so pin `+=` after it
    for (int i = 0; i < W.size; ++i) W.gradient[i] = rand() % 3;
    for (int i = 0; i < T.size; ++i) T.gradient[i] = rand() % 3;
    std::vector<double> w = vals(&W), t = vals(&T), dw = vals(&W, true), dt = vals(&T, true);
    int bad = mix.nRows != N || mix.nColumns != N || mix.nDepth != Kout;
    std::vector<double> ref((size_t)N * N * Kout), g(ref.size());
    gfo_custommatmultensor_forward(&w[0], &t[0], &ref[0], N * N, V, Kout);
    bad |= check("CustomMatMulTensor forward", vals(&mix), ref);
    for (int i = 0; i < mix.size; ++i) mix.gradient[i] = g[i] = (rand() % 200 - 100) / 100.0;
    graph.backward();
    gfo_custommatmultensor_backward(&g[0], &w[0], &t[0], &dw[0], &dt[0], N * N, V, Kout);
    bad |= check("CustomMatMulTensor backward (first)", vals(&W, true), dw);
    bad |= check("CustomMatMulTensor backward (second)", vals(&T, true), dt);
    return bad;
}

int main() {
    int bad = part1_matmul();
    bad |= part2_vertex_chain();
    bad |= part3_channel_mix();
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
