// ref_shim.cpp -- C-ABI shim around the REAL reference headers (fp64 GraphFlow/ variant).
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile with
//     g++ -std=c++11 -O2 -DNDEBUG -pthread -I$(GF_REFERENCE)/GraphFlow ref_shim.cpp
// into oracle/_ref/libgf_ref.so (git-ignored; it is a build of GPL-3 reference code and never
// enters history).  No reference source is copied here: this file only #includes the headers
// where they lie under /root/reference and marshals plain buffers in and out of the reference's
// own op objects.  Used to (a) generate tests/golden/*.npz, (b) validate oracle/gf_oracle.c,
// (c) serve as the "reference"-kind CPU baseline in bench.py.
//
// -DNDEBUG is required because RisiContraction_18_thread's constructors assert(DEPRECATED == false)
// (RisiContraction_18_thread.h:27,33,784).
#include <cstddef>
#include <cstdlib>
#include <vector>

#include "Matrix.h"
#include "Tensor3D.h"
#include "Tensor4D.h"
#include "RisiContraction_4.h"
#include "RisiContraction_10.h"
#include "RisiContraction_18.h"
#include "RisiContraction_18_thread.h"
#include "RisiContraction_18_dropout.h"
#include "RisiContraction_50.h"
#include "MatMul.h"
#include "MatTensorMul.h"
#include "TensorMatMul.h"
#include "StackTensor3D.h"
#include "CustomMatMulTensor.h"

namespace {

struct Neighbourhood {
    std::vector<Tensor3D *> t;
    Matrix *adj;
    int N, C;
    Neighbourhood(const double *P, const double *dP, const double *A, int N_, int C_) : adj(NULL), N(N_), C(C_) {
        const size_t per = (size_t)N * N * C;
        for (int a = 0; a < N; ++a) {
            Tensor3D *x = new Tensor3D(N, N, C);
            for (size_t i = 0; i < per; ++i) {
                x->value[i] = P ? P[a * per + i] : 0.0;
                x->gradient[i] = dP ? dP[a * per + i] : 0.0;
            }
            t.push_back(x);
        }
        if (A) {
            adj = new Matrix(N, N);
            for (int i = 0; i < N * N; ++i) adj->value[i] = A[i];
        }
    }
    void grads_out(double *dP) const {
        const size_t per = (size_t)N * N * C;
        for (int a = 0; a < N; ++a)
            for (size_t i = 0; i < per; ++i) dP[a * per + i] = t[a]->gradient[i];
    }
    ~Neighbourhood() {
        for (size_t i = 0; i < t.size(); ++i) delete t[i];
        delete adj;
    }
};

template <class Op>
void bind(Op &op, Neighbourhood &nb, bool with_adj) {
    op.clear();
    for (int a = 0; a < nb.N; ++a) op.add_tensor(nb.t[a]);
    (void)with_adj;
}

template <class Op>
void contract_fwd(const double *P, const double *A, double *Out, int N, int C) {
    Neighbourhood nb(P, NULL, A, N, C);
    Op op(N, C);
    bind(op, nb, true);
    op.set_adjacency(nb.adj);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
}

template <class Op>
void contract_bwd(const double *G, const double *A, double *dP, int N, int C) {
    Neighbourhood nb(NULL, dP, A, N, C);
    Op op(N, C);
    bind(op, nb, true);
    op.set_adjacency(nb.adj);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    nb.grads_out(dP);
}

}  // namespace

extern "C" {

void ref_r18_forward(const double *P, const double *A, double *Out, int N, int C) {
    contract_fwd<RisiContraction_18>(P, A, Out, N, C);
}
void ref_r18_backward(const double *G, const double *A, double *dP, int N, int C) {
    contract_bwd<RisiContraction_18>(G, A, dP, N, C);
}
void ref_r18_thread_forward(const double *P, const double *A, double *Out, int N, int C) {
    contract_fwd<RisiContraction_18_thread>(P, A, Out, N, C);
}
// RisiContraction_18_dropout: srand(seed) then one forward (train mode draws the kept slices with rand(), :113-125),
// optionally followed by backward.  use_out[18] receives the mask the reference drew.
void ref_r18_dropout(unsigned seed, int nKept, int train, const double *P, const double *A, const double *G, double *Out,
                     double *dP, int *use_out, int N, int C) {
    Neighbourhood nb(P, dP, A, N, C);
    RisiContraction_18_dropout op(N, C);
    bind(op, nb, true);
    op.set_adjacency(nb.adj);
    op.setContractions(nKept);
    op.setMode(train != 0);
    srand(seed);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
    for (int k = 0; k < 18; ++k) use_out[k] = op.use[k] ? 1 : 0;
    if (train && G && dP) {
        for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
        op.backward();
        nb.grads_out(dP);
    }
}
void ref_r10_forward(const double *P, const double *A, double *Out, int N, int C) {
    contract_fwd<RisiContraction_10>(P, A, Out, N, C);
}
void ref_r10_backward(const double *G, const double *A, double *dP, int N, int C) {
    contract_bwd<RisiContraction_10>(G, A, dP, N, C);
}
void ref_r50_forward(const double *P, const double *A, double *Out, int N, int C) {
    contract_fwd<RisiContraction_50>(P, A, Out, N, C);
}
void ref_r50_backward(const double *G, const double *A, double *dP, int N, int C) {
    contract_bwd<RisiContraction_50>(G, A, dP, N, C);
}

void ref_r4_forward(const double *P, double *Out, int N, int C) {
    Neighbourhood nb(P, NULL, NULL, N, C);
    RisiContraction_4 op(N, C);
    bind(op, nb, false);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
}
void ref_r4_backward(const double *G, double *dP, int N, int C) {
    Neighbourhood nb(NULL, dP, NULL, N, C);
    RisiContraction_4 op(N, C);
    bind(op, nb, false);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    nb.grads_out(dP);
}

// Timed entry for the CPU baseline: forward + backward of RisiContraction_18 on one graph,
// objects built outside the timed calls is not possible through a flat ABI, so the caller times
// this whole function; construction is O(N^3 C) copies, negligible next to the O(nnz N^3 C) loops.
void ref_r18_fwd_bwd(const double *P, const double *A, const double *G, double *Out, double *dP, int N, int C) {
    Neighbourhood nb(P, dP, A, N, C);
    RisiContraction_18 op(N, C);
    bind(op, nb, true);
    op.set_adjacency(nb.adj);
    op.forward();
    for (int i = 0; i < op.size; ++i) {
        Out[i] = op.value[i];
        op.gradient[i] = G[i];
    }
    op.backward();
    nb.grads_out(dP);
}

static void fill(Vector *v, const double *val, const double *grad) {
    for (int i = 0; i < v->size; ++i) {
        v->value[i] = val ? val[i] : 0.0;
        v->gradient[i] = grad ? grad[i] : 0.0;
    }
}

void ref_matmul_forward(const double *A, const double *B, double *C, int M, int K, int N) {
    Matrix a(M, K), b(K, N);
    fill(&a, A, NULL);
    fill(&b, B, NULL);
    MatMul op(&a, &b);
    op.forward();
    for (int i = 0; i < op.size; ++i) C[i] = op.value[i];
}
void ref_matmul_backward(const double *dC, const double *A, const double *B, double *dA, double *dB, int M, int K,
                         int N) {
    Matrix a(M, K), b(K, N);
    fill(&a, A, dA);
    fill(&b, B, dB);
    MatMul op(&a, &b);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = dC[i];
    op.backward();
    for (int i = 0; i < a.size; ++i) dA[i] = a.gradient[i];
    for (int i = 0; i < b.size; ++i) dB[i] = b.gradient[i];
}

void ref_mattensormul_forward(const double *X, const double *F, double *Out, int R, int Kd, int J, int D) {
    Matrix x(R, Kd);
    Tensor3D f(Kd, J, D);
    fill(&x, X, NULL);
    fill(&f, F, NULL);
    MatTensorMul op(&x, &f);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
}
void ref_mattensormul_backward(const double *G, const double *X, const double *F, double *dX, double *dF, int R,
                               int Kd, int J, int D) {
    Matrix x(R, Kd);
    Tensor3D f(Kd, J, D);
    fill(&x, X, dX);
    fill(&f, F, dF);
    MatTensorMul op(&x, &f);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    for (int i = 0; i < x.size; ++i) dX[i] = x.gradient[i];
    for (int i = 0; i < f.size; ++i) dF[i] = f.gradient[i];
}

void ref_tensormatmul_forward(const double *F, const double *Y, double *Out, int R, int Kd, int J, int D) {
    Tensor3D f(R, Kd, D);
    Matrix y(Kd, J);
    fill(&f, F, NULL);
    fill(&y, Y, NULL);
    TensorMatMul op(&f, &y);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
}
void ref_tensormatmul_backward(const double *G, const double *F, const double *Y, double *dF, double *dY, int R,
                               int Kd, int J, int D) {
    Tensor3D f(R, Kd, D);
    Matrix y(Kd, J);
    fill(&f, F, dF);
    fill(&y, Y, dY);
    TensorMatMul op(&f, &y);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    for (int i = 0; i < f.size; ++i) dF[i] = f.gradient[i];
    for (int i = 0; i < y.size; ++i) dY[i] = y.gradient[i];
}

void ref_custommatmultensor_forward(const double *W, const double *T, double *Out, int I, int J, int V, int Kout) {
    Matrix w(Kout, V);
    Tensor3D t(I, J, V);
    fill(&w, W, NULL);
    fill(&t, T, NULL);
    CustomMatMulTensor op(&w, &t);
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
}
void ref_custommatmultensor_backward(const double *G, const double *W, const double *T, double *dW, double *dT, int I, int J,
                                     int V, int Kout) {
    Matrix w(Kout, V);
    Tensor3D t(I, J, V);
    fill(&w, W, dW);
    fill(&t, T, dT);
    CustomMatMulTensor op(&w, &t);
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    for (int i = 0; i < w.size; ++i) dW[i] = w.gradient[i];
    for (int i = 0; i < t.size; ++i) dT[i] = t.gradient[i];
}

// StackTensor3D round trip on nRows tensors of [nCols][n1][n2] given contiguously.
void ref_stack_forward(const double *T, double *Out, int nRows, int nCols, int n1, int n2) {
    const size_t per = (size_t)nCols * n1 * n2;
    std::vector<Tensor3D *> t;
    StackTensor3D op(nRows, nCols, n1, n2);
    for (int r = 0; r < nRows; ++r) {
        t.push_back(new Tensor3D(nCols, n1, n2));
        fill(t[r], T + r * per, NULL);
        op.add_tensor(t[r]);
    }
    op.forward();
    for (int i = 0; i < op.size; ++i) Out[i] = op.value[i];
    for (int r = 0; r < nRows; ++r) delete t[r];
}
void ref_stack_backward(const double *G, double *dT, int nRows, int nCols, int n1, int n2) {
    const size_t per = (size_t)nCols * n1 * n2;
    std::vector<Tensor3D *> t;
    StackTensor3D op(nRows, nCols, n1, n2);
    for (int r = 0; r < nRows; ++r) {
        t.push_back(new Tensor3D(nCols, n1, n2));
        fill(t[r], NULL, dT + r * per);
        op.add_tensor(t[r]);
    }
    for (int i = 0; i < op.size; ++i) op.gradient[i] = G[i];
    op.backward();
    for (int r = 0; r < nRows; ++r) {
        for (size_t i = 0; i < per; ++i) dT[r * per + i] = t[r]->gradient[i];
        delete t[r];
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// SMP_omega (the caller of the ops above), driven with DUMPED parameters so goldens do not depend on rand():
// the flat parameter / gradient layout is the registration order of SMP_omega.h:289-295 (H, K_1, b_1, ..., W).
// ---------------------------------------------------------------------------------------------------------------
#include "SMP_omega.h"

extern "C" int ref_smp_omega_run_acts(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                 int has_WL, int V, const int *adj, const double *feature, double target,
                                 const double *params, double *graph_feature, double *predict, double *loss,
                                 double *grads, int *phi /* [L+1][V][max_rf+1], slot 0 = size */,
                                 double *reduced_adj /* [L+1][V][max_rf*max_rf] */,
                                 const double *coulomb /* NULL, or V x V: the use_coulomb constructor (:91-113) */,
                                 double *acts /* NULL, or level[l]->f[v]->value ([s][s][C]) back to back in (l, v) order */);
extern "C" int ref_smp_omega_run(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                 int has_WL, int V, const int *adj, const double *feature, double target,
                                 const double *params, double *graph_feature, double *predict, double *loss,
                                 double *grads, int *phi, double *reduced_adj, const double *coulomb) {
    return ref_smp_omega_run_acts(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth, has_WL, V, adj, feature, target, params,
                                  graph_feature, predict, loss, grads, phi, reduced_adj, coulomb, NULL);
}
extern "C" int ref_smp_omega_run_acts(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                 int has_WL, int V, const int *adj, const double *feature, double target,
                                 const double *params, double *graph_feature, double *predict, double *loss,
                                 double *grads, int *phi, double *reduced_adj, const double *coulomb, double *acts) {
    // heap-allocated and intentionally leaked: ~SMP_omega / ~DenseGraph free memory the executor's destructor also
    // frees (SURVEY.md 8b "Ownership"), which is fatal in a long-lived process
    SMP_omega &net = coulomb ? *new SMP_omega(true, max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth, has_WL != 0)
                             : *new SMP_omega(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth, has_WL != 0);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) net.sgd->params[i]->value[j] = params[off++];
    DenseGraph &g = *new DenseGraph(V, nFeatures);
    for (int i = 0; i < V; ++i) {
        for (int j = 0; j < V; ++j) g.adj[i][j] = adj[i * V + j];
        if (coulomb)
            for (int j = 0; j < V; ++j) g.coulomb[i][j] = coulomb[i * V + j];
        for (int f = 0; f < nFeatures; ++f) g.feature[i][f] = feature[i * nFeatures + f];
    }
    net.complete_computation_graph(&g);
    net.target->value[0] = target;
    net.graph->forward();
    net.graph->backward();
    for (int f = 0; f < nChanels; ++f) graph_feature[f] = net.graph_feature->value[f];
    *predict = net.predict->value[0];
    *loss = net.sql->getLoss();
    off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) grads[off++] = net.sgd->params[i]->gradient[j];
    for (int l = 0; l <= nLevels; ++l)
        for (int v = 0; v < V; ++v) {
            int *p = phi + ((size_t)l * V + v) * (max_rf + 1);
            const std::vector<int> &f = net.level[l]->phi[v];
            p[0] = (int)f.size();
            for (size_t i = 0; i < f.size(); ++i) p[1 + i] = f[i];
            if (l > 0 && reduced_adj) {
                double *a = reduced_adj + ((size_t)l * V + v) * max_rf * max_rf;
                for (int i = 0; i < net.level[l]->adj[v]->size; ++i) a[i] = net.level[l]->adj[v]->value[i];
            }
        }
    if (acts) {
        size_t ao = 0;
        for (int l = 0; l <= nLevels; ++l)
            for (int v = 0; v < V; ++v) {
                const size_t s = net.level[l]->phi[v].size(), n = s * s * nChanels;
                for (size_t i = 0; i < n; ++i) acts[ao + i] = net.level[l]->f[v]->value[i];
                ao += n;
            }
    }
    return (int)off;
}

// SMP_beta (GraphFlow/SMP_beta.h): the same model without the receptive-field cap.  One molecule, dumped parameters.
#include "SMP_beta.h"
extern "C" int ref_smp_beta_run(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, int has_WL, int V,
                                const int *adj, const double *feature, double target, const double *params,
                                double *graph_feature, double *predict, double *loss, double *grads) {
    SMP_beta &net = *new SMP_beta(max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL != 0);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) net.sgd->params[i]->value[j] = params[off++];
    DenseGraph &g = *new DenseGraph(V, nFeatures);
    for (int i = 0; i < V; ++i) {
        for (int j = 0; j < V; ++j) g.adj[i][j] = adj[i * V + j];
        for (int f = 0; f < nFeatures; ++f) g.feature[i][f] = feature[i * nFeatures + f];
    }
    net.complete_computation_graph(&g);
    net.target->value[0] = target;
    net.graph->forward();
    net.graph->backward();
    for (int f = 0; f < nChanels; ++f) graph_feature[f] = net.graph_feature->value[f];
    *predict = net.predict->value[0];
    *loss = net.sql->getLoss();
    off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) grads[off++] = net.sgd->params[i]->gradient[j];
    return (int)off;
}

// SMP_2D_ver6 / ver7 / ver8 (RisiContraction_10 / _50 / _18 + CustomMatMulTensor, no receptive-field cap).
// (every one of these headers defines a global `const int INF`: give each its own name)
#define INF INF_2d_ver6
#include "SMP_2D_ver6.h"
#undef INF
#define INF INF_2d_ver7
#include "SMP_2D_ver7.h"
#undef INF
#define INF INF_2d_ver8
#include "SMP_2D_ver8.h"
#undef INF
namespace {
template <class Net>
int smp2d_run(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, int has_WL, int V, const int *adj,
              const double *feature, double target, const double *params, double *graph_feature, double *predict, double *loss,
              double *grads, int *phi, int phi_stride) {
    Net &net = *new Net(max_nVertices, nLevels, nChanels, nFeatures, nDepth, 0.9, has_WL != 0);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) net.sgd->params[i]->value[j] = params[off++];
    DenseGraph &g = *new DenseGraph(V, nFeatures);
    for (int i = 0; i < V; ++i) {
        for (int j = 0; j < V; ++j) g.adj[i][j] = adj[i * V + j];
        for (int f = 0; f < nFeatures; ++f) g.feature[i][f] = feature[i * nFeatures + f];
    }
    net.complete_computation_graph(&g);
    net.target->value[0] = target;
    net.graph->forward();
    net.graph->backward();
    for (int f = 0; f < nChanels; ++f) graph_feature[f] = net.graph_feature->value[f];
    *predict = net.predict->value[0];
    *loss = net.sql->getLoss();
    off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) grads[off++] = net.sgd->params[i]->gradient[j];
    for (int l = 0; l <= nLevels; ++l)
        for (int v = 0; v < V; ++v) {
            int *p = phi + ((size_t)l * V + v) * phi_stride;
            const std::vector<int> &f = net.level[l]->phi[v];
            p[0] = (int)f.size();
            for (size_t i = 0; i < f.size(); ++i) p[1 + i] = f[i];
        }
    return (int)off;
}
}  // namespace
namespace {
template <class Net>
int smp2d_batchlearn(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, double momentum_param, int nMol,
                     const int *nV, const int *adj, const double *feature, const double *targets, int seed, double *params0_out,
                     int nIter, double learning_rate, double *losses, double *params_out) {
    srand((unsigned)seed);
    Net &net = *new Net(max_nVertices, nLevels, nChanels, nFeatures, nDepth, momentum_param);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) params0_out[off++] = net.sgd->params[i]->value[j];
    std::vector<DenseGraph *> mol(nMol);
    std::vector<double> tgt(targets, targets + nMol);
    size_t ao = 0, fo = 0;
    for (int m = 0; m < nMol; ++m) {
        const int V = nV[m];
        mol[m] = new DenseGraph(V, nFeatures);
        for (int i = 0; i < V; ++i) {
            for (int j = 0; j < V; ++j) mol[m]->adj[i][j] = adj[ao + (size_t)i * V + j];
            for (int f = 0; f < nFeatures; ++f) mol[m]->feature[i][f] = feature[fo + (size_t)i * nFeatures + f];
        }
        ao += (size_t)V * V;
        fo += (size_t)V * nFeatures;
    }
    for (int it = 0; it < nIter; ++it) {
        std::pair<double, double> r = net.BatchLearn(nMol, &mol[0], &tgt[0], learning_rate);
        losses[2 * it] = r.first;
        losses[2 * it + 1] = r.second;
    }
    off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) params_out[off++] = net.sgd->params[i]->value[j];
    return (int)off;
}
}  // namespace
// nIter x the REAL SMP_2D_ver{6,7,8}::BatchLearn (Momentum optimiser) from the weights its constructor draws after srand(seed)
extern "C" int ref_smp_2d_batchlearn(int version, int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth,
                                     double momentum_param, int nMol, const int *nV, const int *adj, const double *feature,
                                     const double *targets, int seed, double *params0_out, int nIter, double learning_rate,
                                     double *losses, double *params_out) {
#define GF_RUN(N) smp2d_batchlearn<N>(max_nVertices, nLevels, nChanels, nFeatures, nDepth, momentum_param, nMol, nV, adj, feature, \
                                      targets, seed, params0_out, nIter, learning_rate, losses, params_out)
    if (version == 6) return GF_RUN(SMP_2D_ver6);
    if (version == 7) return GF_RUN(SMP_2D_ver7);
    if (version == 8) return GF_RUN(SMP_2D_ver8);
#undef GF_RUN
    return -1;
}

extern "C" int ref_smp_2d_run(int version, int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, int has_WL,
                              int V, const int *adj, const double *feature, double target, const double *params,
                              double *graph_feature, double *predict, double *loss, double *grads, int *phi, int phi_stride) {
#define GF_RUN(N) smp2d_run<N>(max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL, V, adj, feature, target, params, \
                               graph_feature, predict, loss, grads, phi, phi_stride)
    if (version == 6) return GF_RUN(SMP_2D_ver6);
    if (version == 7) return GF_RUN(SMP_2D_ver7);
    if (version == 8) return GF_RUN(SMP_2D_ver8);
#undef GF_RUN
    return -1;
}

// Text checkpoint written by the reference's own SMP_omega::save_model (SMP_omega.h:1033-1042) for given parameters.
extern "C" int ref_smp_omega_save_model(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                        const double *params, const char *path) {
    SMP_omega &net = *new SMP_omega(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) net.sgd->params[i]->value[j] = params[off++];
    net.save_model(path);
    return (int)off;
}

// nIter calls of the REAL SMP_omega::BatchLearn(nBatch, molecule, target, learning_rate) (SMP_omega.h:798-825: loss before,
// summed gradients, Adam::Learn(alpha, nBatch), loss after) from given parameters; returns the parameters afterwards.
// seed >= 0: ignore `params`, srand(seed) before constructing -- the constructor's own weights_initialization
// (SMP_omega.h:334-338, GraphFlow.h:1297-1306) then draws the initial weights, which are returned in params0_out.
extern "C" int ref_smp_omega_batchlearn(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                        int nMol, const int *nV, const int *adj, const double *feature, const double *targets,
                                        const double *params, int seed, double *params0_out, int nIter, double learning_rate,
                                        double *losses /* [nIter][2] */, double *params_out) {
    if (seed >= 0) srand((unsigned)seed);
    SMP_omega &net = *new SMP_omega(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth);
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) {
            if (seed < 0) net.sgd->params[i]->value[j] = params[off];
            if (params0_out) params0_out[off] = net.sgd->params[i]->value[j];
            ++off;
        }
    std::vector<DenseGraph *> mol(nMol);
    std::vector<double> tgt(targets, targets + nMol);
    size_t ao = 0, fo = 0;
    for (int m = 0; m < nMol; ++m) {
        const int V = nV[m];
        mol[m] = new DenseGraph(V, nFeatures);
        for (int i = 0; i < V; ++i) {
            for (int j = 0; j < V; ++j) mol[m]->adj[i][j] = adj[ao + (size_t)i * V + j];
            for (int f = 0; f < nFeatures; ++f) mol[m]->feature[i][f] = feature[fo + (size_t)i * nFeatures + f];
        }
        ao += (size_t)V * V;
        fo += (size_t)V * nFeatures;
    }
    for (int it = 0; it < nIter; ++it) {
        std::pair<double, double> r = net.BatchLearn(nMol, &mol[0], &tgt[0], learning_rate);
        losses[2 * it] = r.first;
        losses[2 * it + 1] = r.second;
    }
    off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) params_out[off++] = net.sgd->params[i]->value[j];
    return (int)off;
}

// CPU baseline for the SMP_omega workload: one model instance, nMol molecules, the per-molecule body of
// SMP_omega::BatchLearn's gradient loop (complete_computation_graph + forward + backward, SMP_omega.h:810-818);
// returns the seconds spent in that loop (construction and allocation are outside the clock).
#include <sys/time.h>
extern "C" double ref_smp_omega_time(int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int nDepth,
                                     int nMol, const int *nV, const int *adj, const double *feature,
                                     const double *targets) {
    SMP_omega &net = *new SMP_omega(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth);
    std::vector<DenseGraph *> mol(nMol);
    size_t ao = 0, fo = 0;
    for (int m = 0; m < nMol; ++m) {
        const int V = nV[m];
        mol[m] = new DenseGraph(V, nFeatures);
        for (int i = 0; i < V; ++i) {
            for (int j = 0; j < V; ++j) mol[m]->adj[i][j] = adj[ao + (size_t)i * V + j];
            for (int f = 0; f < nFeatures; ++f) mol[m]->feature[i][f] = feature[fo + (size_t)i * nFeatures + f];
        }
        ao += (size_t)V * V;
        fo += (size_t)V * nFeatures;
    }
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    for (int m = 0; m < nMol; ++m) {
        net.complete_computation_graph(mol[m]);
        net.target->value[0] = targets[m];
        net.graph->forward();
        net.graph->backward();
    }
    gettimeofday(&t1, NULL);
    return (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
}

// The all-cores CPU baseline of SURVEY 8(d)(iii): the REAL SMP_omega::Threaded_BatchLearn (SMP_omega.h:750-792) with
// init_multi_threads(nThreads) -- one model clone per thread, the batch in waves of nThreads molecules.  Returns the seconds
// of the call (clone construction outside the clock); includes the Adam step, which is negligible.
extern "C" double ref_smp_omega_threaded_time(int nThreads, int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures,
                                              int nDepth, int nMol, const int *nV, const int *adj, const double *feature,
                                              const double *targets) {
    SMP_omega &net = *new SMP_omega(max_nVertices, max_rf, nLevels, nChanels, nFeatures, nDepth);
    net.init_multi_threads(nThreads);
    std::vector<DenseGraph *> mol(nMol);
    std::vector<double> tgt(targets, targets + nMol);
    size_t ao = 0, fo = 0;
    for (int m = 0; m < nMol; ++m) {
        const int V = nV[m];
        mol[m] = new DenseGraph(V, nFeatures);
        for (int i = 0; i < V; ++i) {
            for (int j = 0; j < V; ++j) mol[m]->adj[i][j] = adj[ao + (size_t)i * V + j];
            for (int f = 0; f < nFeatures; ++f) mol[m]->feature[i][f] = feature[fo + (size_t)i * nFeatures + f];
        }
        ao += (size_t)V * V;
        fo += (size_t)V * nFeatures;
    }
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    net.Threaded_BatchLearn(nMol, &mol[0], &tgt[0], 1e-3);
    gettimeofday(&t1, NULL);
    return (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
}

// ---------------------------------------------------------------------------------------------------------------
// The `_physics` and `_pairgraphs` drivers: raw features, halving channels, every level read out, an MLP head.
// One sample, dumped parameters in the class's own registration order (sgd->params), forward + backward.
// ---------------------------------------------------------------------------------------------------------------
#include "SMP_omega_physics.h"
#include "SMP_beta_physics.h"
#include "SMP_omega_pairgraphs.h"
#include "SMP_beta_pairgraphs.h"
#include "SMP_sigma_pairgraphs.h"

namespace {
DenseGraph *make_graph(int V, int F, const int *adj, const double *feature) {
    DenseGraph *g = new DenseGraph(V, F);
    for (int i = 0; i < V; ++i) {
        for (int j = 0; j < V; ++j) g->adj[i][j] = adj[i * V + j];
        for (int f = 0; f < F; ++f) g->feature[i][f] = feature[i * F + f];
    }
    return g;
}
template <class Net>
int set_params(Net &net, const double *params) {
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) net.sgd->params[i]->value[j] = params[off++];
    return (int)off;
}
template <class Net>
int get_grads(Net &net, double *grads) {
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) grads[off++] = net.sgd->params[i]->gradient[j];
    return (int)off;
}
template <class Level>
void dump_phi(Level **level, int nLevels, int V, int stride, int *phi) {
    for (int l = 0; l <= nLevels; ++l)
        for (int v = 0; v < V; ++v) {
            int *p = phi + ((size_t)l * V + v) * stride;
            const std::vector<int> &f = level[l]->phi[v];
            p[0] = (int)f.size();
            for (size_t i = 0; i < f.size(); ++i) p[1 + i] = f[i];
        }
}
template <class Net>
int physics_run(Net &net, int nLevels, int nFeatures, int V, const int *adj, const double *feature, double target, const double *params,
                double *predict, double *loss, double *grads, int *phi, int phi_stride, double *graph_feature) {
    set_params(net, params);
    DenseGraph *g = make_graph(V, nFeatures, adj, feature);
    net.complete_computation_graph(g);
    net.target->value[0] = target;
    net.graph->forward();
    net.graph->backward();
    *predict = net.predict->value[0];
    *loss = net.sql->getLoss();
    if (graph_feature)
        for (int i = 0; i < net.graph_feature->size; ++i) graph_feature[i] = net.graph_feature->value[i];
    dump_phi(net.level, nLevels, V, phi_stride, phi);
    return get_grads(net, grads);
}
template <class Net>
int pair_run(Net &net, int nLevels, int F1, int F2, int V1, const int *adj1, const double *feat1, int V2, const int *adj2,
             const double *feat2, double target, const double *params, double *predict, double *loss, double *grads, int *phi1,
             int *phi2, int phi_stride, double *graph_feature) {
    set_params(net, params);
    DenseGraph *g1 = make_graph(V1, F1, adj1, feat1), *g2 = make_graph(V2, F2, adj2, feat2);
    net.complete_computation_graph(g1, g2);
    net.target->value[0] = target;
    net.graph->forward();
    net.graph->backward();
    *predict = net.predict->value[0];
    *loss = net.sql->getLoss();
    if (graph_feature)
        for (int i = 0; i < net.graph_feature->size; ++i) graph_feature[i] = net.graph_feature->value[i];
    dump_phi(net.level_1, nLevels, V1, phi_stride, phi1);
    dump_phi(net.level_2, nLevels, V2, phi_stride, phi2);
    return get_grads(net, grads);
}
}  // namespace

// beta != 0: SMP_beta_physics (no receptive-field cap).  Returns the parameter count.
extern "C" int ref_smp_physics_run(int beta, int max_nVertices, int max_rf, int nLevels, int nChanels, int nFeatures, int V, const int *adj,
                                   const double *feature, double target, const double *params, double *predict, double *loss,
                                   double *grads, int *phi, int phi_stride, double *graph_feature) {
    if (beta) {
        SMP_beta_physics &net = *new SMP_beta_physics(max_nVertices, nLevels, nChanels, nFeatures);
        return physics_run(net, nLevels, nFeatures, V, adj, feature, target, params, predict, loss, grads, phi, phi_stride, graph_feature);
    }
    SMP_omega_physics &net = *new SMP_omega_physics(max_nVertices, max_rf, nLevels, nChanels, nFeatures);
    return physics_run(net, nLevels, nFeatures, V, adj, feature, target, params, predict, loss, grads, phi, phi_stride, graph_feature);
}

// kind 0: SMP_omega_pairgraphs, 1: SMP_beta_pairgraphs, 2: SMP_sigma_pairgraphs (RisiContraction_18_dropout; nKept slices kept;
// train != 0: the masks are drawn with rand() after srand(seed), test mode otherwise).  Returns the parameter count.
extern "C" int ref_smp_pairgraphs_run(int kind, int maxV1, int maxV2, int max_rf, int nLevels, int nChanels, int F1, int F2, int V1,
                                      const int *adj1, const double *feat1, int V2, const int *adj2, const double *feat2, double target,
                                      const double *params, int nKept, int train, int seed, double *predict, double *loss, double *grads,
                                      int *phi1, int *phi2, int phi_stride, double *graph_feature) {
    if (kind == 1) {
        SMP_beta_pairgraphs &net = *new SMP_beta_pairgraphs(maxV1, maxV2, nLevels, nChanels, F1, F2);
        return pair_run(net, nLevels, F1, F2, V1, adj1, feat1, V2, adj2, feat2, target, params, predict, loss, grads, phi1, phi2, phi_stride, graph_feature);
    }
    if (kind == 2) {
        SMP_sigma_pairgraphs &net = *new SMP_sigma_pairgraphs(maxV1, maxV2, max_rf, nLevels, nChanels, F1, F2, nKept);
        net.setMode(train != 0);
        srand((unsigned)seed);
        return pair_run(net, nLevels, F1, F2, V1, adj1, feat1, V2, adj2, feat2, target, params, predict, loss, grads, phi1, phi2, phi_stride, graph_feature);
    }
    SMP_omega_pairgraphs &net = *new SMP_omega_pairgraphs(maxV1, maxV2, max_rf, nLevels, nChanels, F1, F2);
    return pair_run(net, nLevels, F1, F2, V1, adj1, feat1, V2, adj2, feat2, target, params, predict, loss, grads, phi1, phi2, phi_stride, graph_feature);
}

// nIter x the REAL BatchLearn of the `_physics` (kind 0 omega, 1 beta) or `_pairgraphs` (kind 10 omega, 11 beta) class from the
// weights its constructor draws after srand(seed) (weights_initialization); returns params0, the (before, after) losses and the
// parameters afterwards.  Graphs back to back as in ref_smp_omega_batchlearn; the second set is ignored for the physics kinds.
namespace {
std::vector<DenseGraph *> unpack_graphs(int nMol, int F, const int *nV, const int *adj, const double *feature) {
    std::vector<DenseGraph *> g(nMol);
    size_t ao = 0, fo = 0;
    for (int m = 0; m < nMol; ++m) {
        g[m] = make_graph(nV[m], F, adj + ao, feature + fo);
        ao += (size_t)nV[m] * nV[m];
        fo += (size_t)nV[m] * F;
    }
    return g;
}
template <class Net>
int dump_params(Net &net, double *out) {
    size_t off = 0;
    for (size_t i = 0; i < net.sgd->params.size(); ++i)
        for (int j = 0; j < net.sgd->params[i]->size; ++j) out[off++] = net.sgd->params[i]->value[j];
    return (int)off;
}
}  // namespace

extern "C" int ref_smp_model_batchlearn(int kind, int maxV, int max_rf, int nLevels, int nChanels, int F1, int F2, int nMol, const int *nV1,
                                        const int *adj1, const double *feat1, const int *nV2, const int *adj2, const double *feat2,
                                        const double *targets, int seed, int nIter, double learning_rate, double *params0,
                                        double *losses /* [nIter][2] */, double *params_out) {
    srand((unsigned)seed);
    std::vector<DenseGraph *> g1 = unpack_graphs(nMol, F1, nV1, adj1, feat1), g2;
    std::vector<double> tgt(targets, targets + nMol);
    int n = 0;
#define GF_RUN(NET, CALL)                                                   \
    {                                                                       \
        NET;                                                                \
        dump_params(net, params0);                                          \
        for (int it = 0; it < nIter; ++it) {                                \
            std::pair<double, double> r = CALL;                             \
            losses[2 * it] = r.first;                                       \
            losses[2 * it + 1] = r.second;                                  \
        }                                                                   \
        n = dump_params(net, params_out);                                   \
    }
    if (kind == 0) GF_RUN(SMP_omega_physics &net = *new SMP_omega_physics(maxV, max_rf, nLevels, nChanels, F1), net.BatchLearn(nMol, &g1[0], &tgt[0], learning_rate))
    if (kind == 1) GF_RUN(SMP_beta_physics &net = *new SMP_beta_physics(maxV, nLevels, nChanels, F1), net.BatchLearn(nMol, &g1[0], &tgt[0], learning_rate))
    if (kind >= 10) g2 = unpack_graphs(nMol, F2, nV2, adj2, feat2);
    if (kind == 10) GF_RUN(SMP_omega_pairgraphs &net = *new SMP_omega_pairgraphs(maxV, maxV, max_rf, nLevels, nChanels, F1, F2), net.BatchLearn(nMol, &g1[0], &g2[0], &tgt[0], learning_rate))
    if (kind == 11) GF_RUN(SMP_beta_pairgraphs &net = *new SMP_beta_pairgraphs(maxV, maxV, nLevels, nChanels, F1, F2), net.BatchLearn(nMol, &g1[0], &g2[0], &tgt[0], learning_rate))
#undef GF_RUN
    return n;
}
