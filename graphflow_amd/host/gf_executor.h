// gf_executor.h -- tag-dispatched graph executor with the reference's observable behaviour
// (GraphFlow/GraphFlow.h:176-1337): add(Entity*, tag) appends; forward() runs every node's forward() in insertion
// order; backward() runs backward() in reverse order; the executor does not own its nodes.
//
// The reference dispatches through two ~500-line `if (TAG == type)` chains of C-style downcasts.  Here a tag is
// bound once to a pair of thunks (forward, backward) for the concrete class, so adding an op is one
// bind<Class>(TAG) line instead of two if-blocks.  Tag values of the ops on the SMP_beta/omega path are the
// reference's own (GraphFlow.h:98-174); the HIP ops get new tags next to the CUDA ones
// (GraphFlow_gpu/GraphFlow.h:179-180 uses 100 and 101).
#ifndef GF_EXECUTOR_H_INCLUDED
#define GF_EXECUTOR_H_INCLUDED

#include <cstdio>
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>

#include "Mixers_hip.h"
#include "RisiContraction_hip.h"
#include "gf_containers.h"

namespace gftags {
// reference values (GraphFlow.h:98-174)
const int VECTOR = 1, MATRIX = 2, TENSOR3D = 3, TENSOR4D = 4;
// HIP ops (new)
const int RISICONTRACTION_4_HIP = 110, RISICONTRACTION_10_HIP = 111, RISICONTRACTION_18_HIP = 112,
          RISICONTRACTION_50_HIP = 113, MATMUL_HIP = 114, MATTENSORMUL_HIP = 115, TENSORMATMUL_HIP = 116,
          STACKTENSOR3D_HIP = 117, CUSTOMMATMULTENSOR_HIP = 118, RISICONTRACTION_18_DROPOUT_HIP = 119;
}  // namespace gftags

class GraphFlowExec {
public:
    typedef void (*thunk)(Entity *);

    GraphFlowExec() {
        bind<Vector>(gftags::VECTOR);
        bind<Matrix>(gftags::MATRIX);
        bind<Tensor3D>(gftags::TENSOR3D);
        bind<Tensor4D>(gftags::TENSOR4D);
        bind<RisiContraction_4_hip>(gftags::RISICONTRACTION_4_HIP);
        bind<RisiContraction_10_hip>(gftags::RISICONTRACTION_10_HIP);
        bind<RisiContraction_18_hip>(gftags::RISICONTRACTION_18_HIP);
        bind<RisiContraction_50_hip>(gftags::RISICONTRACTION_50_HIP);
        bind<MatMul_hip>(gftags::MATMUL_HIP);
        bind<MatTensorMul_hip>(gftags::MATTENSORMUL_HIP);
        bind<TensorMatMul_hip>(gftags::TENSORMATMUL_HIP);
        bind<StackTensor3D_hip>(gftags::STACKTENSOR3D_HIP);
        bind<CustomMatMulTensor_hip>(gftags::CUSTOMMATMULTENSOR_HIP);
        bind<RisiContraction_18_dropout_hip>(gftags::RISICONTRACTION_18_DROPOUT_HIP);
    }

    // Teach the executor a (tag -> class) pair.  The class needs public non-virtual forward()/backward().
    template <class Op>
    void bind(int tag) {
        table[tag] = std::make_pair(&call_forward<Op>, &call_backward<Op>);
    }

    void add(Entity *e, int tag) {
        std::map<int, std::pair<thunk, thunk> >::const_iterator it = table.find(tag);
        if (it == table.end()) {
            std::fprintf(stderr, "GraphFlowExec::add: no op bound to tag %d\n", tag);
            std::abort();
        }
        Node n = {e, it->second.first, it->second.second, tag};
        topology.push_back(n);
    }
    void clear() { topology.clear(); }
    size_t size() const { return topology.size(); }

    void forward() {
        for (size_t i = 0; i < topology.size(); ++i) topology[i].fwd(topology[i].e);
    }
    void backward() {
        for (size_t i = topology.size(); i-- > 0;) topology[i].bwd(topology[i].e);
    }

private:
    struct Node {
        Entity *e;
        thunk fwd, bwd;
        int tag;
    };
    template <class Op>
    static void call_forward(Entity *e) {
        static_cast<Op *>(e)->forward();
    }
    template <class Op>
    static void call_backward(Entity *e) {
        static_cast<Op *>(e)->backward();
    }
    std::map<int, std::pair<thunk, thunk> > table;
    std::vector<Node> topology;
};

#endif
