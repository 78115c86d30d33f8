// test_RisiContraction_hip.cpp -- op-level CPU-oracle-vs-GPU parity + timing, in the style of the reference's
// tests/test_RisiContraction_18_gpu.cu (N and nChanels from argv, srand(123456789), symmetric integer tensors,
// symmetric 0/1 adjacency with unit diagonal, "sum of absolute errors" printed for forward and backward).
// Differences: the GPU op is RisiContraction_K_hip driven through the tag-dispatched executor in BOTH binding
// styles (add_tensor and pre-stacked Tensor4D), the ground truth is the fp64 oracle (oracle/gf_oracle.c -- the
// reference itself cannot travel to the GPU box), and the program FAILS (exit 1) above a relative error of 1e-5.
//
// usage: test_RisiContraction_hip N nChanels [K=18]
#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gf_executor.h"

extern "C" {
int gfo_contract_forward(int K, const double *P, const double *A, double *Out, int N, int C);
int gfo_contract_backward(int K, const double *G, const double *A, double *dP, int N, int C);
void gfo_r4_forward(const double *P, double *Out, int N, int C);
void gfo_r4_backward(const double *G, double *dP, int N, int C);
}

static double now_ms() {
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return tp.tv_sec * 1e3 + tp.tv_usec * 1e-3;
}

template <int K>
int run(int N, int C) {
    srand(123456789);
    std::vector<Tensor3D *> tensors(N);
    for (int i = 0; i < N; ++i) {
        tensors[i] = new Tensor3D(N, N, C);
        for (int ch = 0; ch < C; ++ch)
            for (int r = 0; r < N; ++r)
                for (int c = r; c < N; ++c) {
                    const int v = rand() % 10;
                    tensors[i]->value[tensors[i]->index(r, c, ch)] = v;
                    tensors[i]->value[tensors[i]->index(c, r, ch)] = v;
                }
        for (int j = 0; j < tensors[i]->size; ++j) tensors[i]->gradient[j] = (rand() % 7) - 3;  // pins the `+=`
    }
    Matrix adj(N, N);
    for (int i = 0; i < N; ++i) {
        adj.value[adj.index(i, i)] = 1;
        for (int j = i + 1; j < N; ++j) {
            const int v = rand() % 2;
            adj.value[adj.index(i, j)] = v;
            adj.value[adj.index(j, i)] = v;
        }
    }

    // style 1: add_tensor / set_adjacency (RisiContraction_18.h)
    RisiContraction_hip<K> op(N, C);
    op.setParameter(N, C);
    for (int i = 0; i < N; ++i) op.add_tensor(tensors[i]);
    if (K != 4) op.set_adjacency(&adj);
    GraphFlowExec graph;
    for (int i = 0; i < N; ++i) graph.add(tensors[i], gftags::TENSOR3D);
    graph.add(&adj, gftags::MATRIX);
    const int tag = K == 4 ? gftags::RISICONTRACTION_4_HIP : K == 10 ? gftags::RISICONTRACTION_10_HIP
                  : K == 18 ? gftags::RISICONTRACTION_18_HIP : gftags::RISICONTRACTION_50_HIP;
    graph.add(&op, tag);

    // ground truth (fp64 oracle) on the stacked copy
    const size_t per = (size_t)N * N * C, nP = per * N, nO = (size_t)N * N * K * C;
    std::vector<double> P(nP), A(N * N), ref_out(nO), G(nO), ref_dP(nP), dP0(nP);
    for (int a = 0; a < N; ++a)
        for (size_t i = 0; i < per; ++i) {
            P[a * per + i] = tensors[a]->value[i];
            dP0[a * per + i] = tensors[a]->gradient[i];
        }
    for (int i = 0; i < N * N; ++i) A[i] = adj.value[i];

    // the executor's forward zeroes the leaves' gradients (Vector::forward) before the op runs
    double t0 = now_ms();
    graph.forward();
    double t1 = now_ms();
    std::printf("GPU forward time (host pointers, incl. PCIe): %.3f ms\n", t1 - t0);
    t0 = now_ms();
    if (K == 4) gfo_r4_forward(&P[0], &ref_out[0], N, C);
    else gfo_contract_forward(K, &P[0], &A[0], &ref_out[0], N, C);
    t1 = now_ms();
    std::printf("CPU forward time (oracle): %.3f ms\n", t1 - t0);

    double err = 0, scale = 1;
    for (size_t i = 0; i < nO; ++i) {
        err += std::fabs(ref_out[i] - op.value[i]);
        scale = std::fmax(scale, std::fabs(ref_out[i]));
    }
    double maxrel = 0;
    for (size_t i = 0; i < nO; ++i) maxrel = std::fmax(maxrel, std::fabs(ref_out[i] - op.value[i]) / scale);
    std::printf("Forward absolute error: %g   (max rel %.3e)\n", err, maxrel);
    int bad = maxrel > 1e-5;
    for (int i = 0; i < op.size; ++i) bad |= (op.gradient[i] != 0);  // forward() must zero own gradient

    // leaves' gradients were zeroed by graph.forward(); seed them again to pin the accumulate contract
    for (int a = 0; a < N; ++a)
        for (size_t i = 0; i < per; ++i) tensors[a]->gradient[i] = dP0[a * per + i];
    for (size_t i = 0; i < nO; ++i) {
        G[i] = rand() % 100;
        op.gradient[i] = G[i];
    }
    t0 = now_ms();
    graph.backward();
    t1 = now_ms();
    std::printf("GPU backward time (host pointers, incl. PCIe): %.3f ms\n", t1 - t0);
    ref_dP = dP0;
    t0 = now_ms();
    if (K == 4) gfo_r4_backward(&G[0], &ref_dP[0], N, C);
    else gfo_contract_backward(K, &G[0], &A[0], &ref_dP[0], N, C);
    t1 = now_ms();
    std::printf("CPU backward time (oracle): %.3f ms\n", t1 - t0);
    err = 0, scale = 1, maxrel = 0;
    for (size_t i = 0; i < nP; ++i) scale = std::fmax(scale, std::fabs(ref_dP[i]));
    for (int a = 0; a < N; ++a)
        for (size_t i = 0; i < per; ++i) {
            const double d = std::fabs(ref_dP[a * per + i] - tensors[a]->gradient[i]);
            err += d;
            maxrel = std::fmax(maxrel, d / scale);
        }
    std::printf("Backward absolute error: %g   (max rel %.3e)\n", err, maxrel);
    bad |= maxrel > 1e-5;

    // style 2: one pre-stacked Tensor4D + adjacency (RisiContraction_18_gpu.h:920) must give the same bits
    Tensor4D stack(N, N, N, C);
    for (size_t i = 0; i < nP; ++i) {
        stack.value[i] = P[i];
        stack.gradient[i] = dP0[i];
    }
    RisiContraction_hip<K> op2(N, C);
    op2.setParameter(&stack, K == 4 ? NULL : &adj);
    op2.forward();
    for (size_t i = 0; i < nO; ++i) {
        bad |= (op2.value[i] != op.value[i]);
        op2.gradient[i] = G[i];
    }
    op2.backward();
    for (int a = 0; a < N; ++a)
        for (size_t i = 0; i < per; ++i) bad |= (stack.gradient[a * per + i] != tensors[a]->gradient[i]);
    std::printf("Tensor4D binding identical to add_tensor binding: %s\n", bad ? "NO / FAILED" : "yes");

    // set_gpu_stream is per op (RisiContraction_18_gpu.h:947-955): op2 moves to a context of its own, `op` stays where it was;
    // results do not depend on the stream, and turn_off_gpu_stream brings op2 back
    op2.set_gpu_stream(NULL);
    op2.forward();
    op.forward();
    for (size_t i = 0; i < nO; ++i) bad |= (op2.value[i] != op.value[i]);
    op2.turn_off_gpu_stream();
    op2.forward();
    for (size_t i = 0; i < nO; ++i) bad |= (op2.value[i] != op.value[i]);
    std::printf("per-op stream context: %s\n", bad ? "FAILED" : "ok");

    // the reference's other two constructors: RisiContraction_18_gpu(Tensor4D*, Matrix*) (RisiContraction_18_gpu.h:879-918, the
    // caller shape of tests/test_RisiContraction_18_gpu.cu) and RisiContraction_18(max rows, max columns, max depth) (:24-26)
    if (K != 4) {
        RisiContraction_hip<K> op3(&stack, &adj);
        op3.forward();
        for (size_t i = 0; i < nO; ++i) bad |= (op3.value[i] != op.value[i]);
        RisiContraction_hip<K> op4(N + 1, N + 1, K * (C + 1));
        op4.setParameter(&stack, &adj);
        op4.forward();
        for (size_t i = 0; i < nO; ++i) bad |= (op4.value[i] != op.value[i]);
        std::printf("(Tensor4D*, Matrix*) and 3-int constructors: %s\n", bad ? "FAILED" : "ok");
    }

    for (int i = 0; i < N; ++i) delete tensors[i];
    std::printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s N nChanels [K]\n", argv[0]);
        return 2;
    }
    const int N = std::atoi(argv[1]), C = std::atoi(argv[2]), K = argc > 3 ? std::atoi(argv[3]) : 18;
    std::printf("-------------------------------------------------------\nN = %d\nnChanels = %d\nK = %d\n", N, C, K);
    switch (K) {
        case 4: return run<4>(N, C);
        case 10: return run<10>(N, C);
        case 18: return run<18>(N, C);
        case 50: return run<50>(N, C);
    }
    std::fprintf(stderr, "K must be 4, 10, 18 or 50\n");
    return 2;
}
