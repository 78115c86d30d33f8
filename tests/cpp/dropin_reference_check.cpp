// dropin_reference_check.cpp -- compile-only proof that the HIP op classes drop into the REAL reference tree:
// the reference's own containers are included first (-I<reference>/GraphFlow), then our op header, and the op is
// used exactly like GraphFlow_gpu/SMP_omega_gpu.h uses RisiContraction_18_gpu.  Built only where the reference
// is mounted (tests/test_host_cpp.py); nothing from the reference is copied.
#include "Matrix.h"
#include "Tensor3D.h"
#include "Tensor4D.h"
#include "StackTensor3D.h"
#include "DenseGraph.h"

#include "Mixers_hip.h"
#include "RisiContraction_hip.h"
#include "gf_executor.h"
#include "SMP_omega_hip.h"

// compiled, never called here (needs a GPU): the model-level drop-in takes the reference's own DenseGraph
double model_level_dropin(DenseGraph **molecules, double *targets, int nBatch) {
    SMP_omega_hip net(29, 29, 3, 64, molecules[0]->nFeatures, 5);
    std::pair<double, double> loss = net.BatchLearn(nBatch, molecules, targets, 1e-3);
    return loss.second + net.Predict(molecules[0]) + net.Feature(molecules[0])[0];
}

int main() {
    const int N = 4, C = 8;
    Tensor3D *t[N];
    StackTensor3D stack(N, N, N, C);
    for (int i = 0; i < N; ++i) {
        t[i] = new Tensor3D(N, N, C);
        stack.add_tensor(t[i]);
    }
    Matrix adj(N, N);
    RisiContraction_18_hip contract(N, C);
    contract.setParameter(&stack, &adj);          // GPU-op style binding on the reference's StackTensor3D
    RisiContraction_18_hip contract2(N, C);
    contract2.setParameter(N, C);                 // CPU-op style binding on the reference's Tensor3D
    for (int i = 0; i < N; ++i) contract2.add_tensor(t[i]);
    contract2.set_adjacency(&adj);
    Matrix W(C, 18 * C);
    Tensor3D view(N, N, 18 * C);
    CustomMatMulTensor_hip mix(&W, &view);        // SMP_2D_ver6-8 style channel mix on the reference's containers
    GraphFlowExec g;
    g.add(&mix, gftags::CUSTOMMATMULTENSOR_HIP);
    g.clear();
    g.add(&adj, gftags::MATRIX);
    g.add(&contract, gftags::RISICONTRACTION_18_HIP);
    return (int)g.size() - 2;
}
