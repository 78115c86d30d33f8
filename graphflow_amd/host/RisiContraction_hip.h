// RisiContraction_hip.h -- drop-in Entity-style ops RisiContraction_{4,10,18,50}_hip.
//
// Same contract as the reference ops (GraphFlow/RisiContraction_18.h:25-63 for the add_tensor interface,
// GraphFlow_gpu/RisiContraction_18_gpu.h:879-955 for the Tensor4D + stream interface):
//   ctor(max N, max channels) allocates value/gradient for [N][N][K*C]; setParameter() rebinds logical dims;
//   clear()/add_tensor()/set_adjacency() bind non-owning input pointers (asserting shapes);
//   forward()  overwrites value, then zeroes own gradient;
//   backward() accumulates into the inputs' gradient (`+=`), reads own gradient; the adjacency gets no gradient.
// The arithmetic happens in libgf_hip.so on the GPU (host-pointer mode of the C ABI); nothing is computed here.
#ifndef RISICONTRACTION_HIP_H_INCLUDED
#define RISICONTRACTION_HIP_H_INCLUDED

#include <cassert>
#include <vector>

#include <cstdlib>

#include "gf_containers.h"
#include "gf_runtime.h"

template <int K>
class RisiContraction_hip : public Tensor3D {
public:
    static const int nContractions = K;

    RisiContraction_hip(int max_N, int max_nChanels)
        : Tensor3D(max_N, max_N, K * max_nChanels), N(max_N), nChanels(max_nChanels), adj(NULL), stacked(NULL), ctx(NULL), own_ctx(NULL) {}
    // RisiContraction_18(int max_nRows, int max_nColumns, int max_nDepth) (RisiContraction_18.h:24-26): a buffer for the largest
    // output the op will hold; N and nChanels come with the first setParameter().
    RisiContraction_hip(int max_nRows, int max_nColumns, int max_nDepth)
        : Tensor3D(max_nRows, max_nColumns, max_nDepth), N(0), nChanels(0), adj(NULL), stacked(NULL), ctx(NULL), own_ctx(NULL) {}
    // RisiContraction_18_gpu(Tensor4D *tensor, Matrix *adj) (RisiContraction_18_gpu.h:879-918): bound to one pre-stacked input
    // from the start (the reference allocates its device mirrors here; ours are the context's staging buffers)
    RisiContraction_hip(Tensor4D *tensor, Matrix *a)
        : Tensor3D(tensor->nRows, tensor->nRows, K * tensor->nChanels2), N(tensor->nRows), nChanels(tensor->nChanels2), adj(a),
          stacked(tensor), ctx(NULL), own_ctx(NULL) {
        assert(tensor->nColumns == N && tensor->nChanels1 == N);
        if (K != 4) assert(a != NULL && a->nRows == N && a->nColumns == N);
    }
    ~RisiContraction_hip() {
        if (own_ctx) gf_ctx_destroy(own_ctx);
    }

    // --- CPU-op style binding (RisiContraction_18.h:36-63) ---
    void setParameter(int N_, int nChanels_) {
        N = N_;
        nChanels = nChanels_;
        Tensor3D::setParameter(N, N, K * nChanels);
        tensors.clear();
        stacked = NULL;
    }
    void clear() {
        tensors.clear();
        stacked = NULL;
    }
    void add_tensor(Tensor3D *t) {
        assert(t->nRows == N && t->nColumns == N && t->nDepth == nChanels);
        tensors.push_back(t);
    }
    void set_adjacency(Matrix *a) {
        assert(a->nRows == N && a->nColumns == N);
        adj = a;
    }

    // --- GPU-op style binding: one pre-stacked Tensor4D [N][N][N][C] (RisiContraction_18_gpu.h:920-936) ---
    void setParameter(Tensor4D *tensor, Matrix *a) {
        N = tensor->nRows;
        nChanels = tensor->nChanels2;
        assert(tensor->nColumns == N && tensor->nChanels1 == N);
        if (K != 4) assert(a != NULL && a->nRows == N && a->nColumns == N);
        Tensor3D::setParameter(N, N, K * nChanels);
        tensors.clear();
        stacked = tensor;
        adj = a;
    }

    // RisiContraction_18_gpu::set_gpu_stream / turn_off_gpu_stream (:947-955).  `stream` is a hipStream_t.
    // The stream belongs to THIS op, as in the reference (a per-object cudaStream_t): the op gets a context of its own the
    // first time a stream is set on it, so the other ops of the thread keep running on the thread's default context.
    void set_gpu_stream(void *stream) {
        if (!own_ctx) {
            const char *dev = std::getenv("GF_DEVICE");
            gf_status st = gf_ctx_create(&own_ctx, dev ? std::atoi(dev) : 0, stream);
            if (st != GF_OK) gfhost::die(NULL, "gf_ctx_create", st);
            ctx = own_ctx;
            return;
        }
        gf_status st = gf_ctx_set_stream(own_ctx, stream);
        if (st != GF_OK) gfhost::die(own_ctx, "gf_ctx_set_stream", st);
        ctx = own_ctx;
    }
    void turn_off_gpu_stream() {   // back to the thread's default context (the device's default stream)
        if (ctx == own_ctx) ctx = NULL;
    }
    void set_context(gf_ctx *c) { ctx = c; }

    void forward() {
        gather(false);
        gf_status st = gfhost::contract_forward_host(context(), K, &vptr[0], K == 4 ? NULL : adj->value, value, N, nChanels);
        if (st != GF_OK) gfhost::die(context(), "RisiContraction_hip::forward", st);
        for (int i = 0; i < size; ++i) gradient[i] = 0;
    }

    void backward() {
        gather(true);
        gf_status st = gfhost::contract_backward_host(context(), K, gradient, K == 4 ? NULL : adj->value, &gptr[0], N, nChanels);
        if (st != GF_OK) gfhost::die(context(), "RisiContraction_hip::backward", st);
    }

    int N;
    int nChanels;
    std::vector<Tensor3D *> tensors;
    Matrix *adj;

protected:
    gf_ctx *context() { return ctx ? ctx : gfhost::default_context(); }
    void gather(bool grads) {
        vptr.resize(N);
        gptr.resize(N);
        if (stacked) {
            const size_t per = (size_t)N * N * nChanels;
            for (int a = 0; a < N; ++a) {
                vptr[a] = stacked->value + a * per;
                gptr[a] = stacked->gradient + a * per;
            }
        } else {
            assert((int)tensors.size() == N);
            for (int a = 0; a < N; ++a) {
                vptr[a] = tensors[a]->value;
                gptr[a] = tensors[a]->gradient;
            }
        }
        if (K != 4) assert(adj != NULL);
        (void)grads;
    }
    Tensor4D *stacked;
    gf_ctx *ctx;
    gf_ctx *own_ctx;   // created by set_gpu_stream, destroyed with the op
    std::vector<const gf_real *> vptr;
    std::vector<gf_real *> gptr;
};

typedef RisiContraction_hip<4> RisiContraction_4_hip;
typedef RisiContraction_hip<10> RisiContraction_10_hip;
typedef RisiContraction_hip<18> RisiContraction_18_hip;
typedef RisiContraction_hip<50> RisiContraction_50_hip;

// RisiContraction_18_dropout (GraphFlow/RisiContraction_18_dropout.h:22-803): slice dropout.  Train mode draws nKept of
// the 18 slices with rand() -- the same draw loop, so the same srand() seed selects the same slices as the reference
// (:113-125) -- drops the others in forward and backward; test mode uses all slices scaled by nKept/18 (:465-471).
class RisiContraction_18_dropout_hip : public RisiContraction_hip<18> {
public:
    RisiContraction_18_dropout_hip(int max_N, int max_nChanels) : RisiContraction_hip<18>(max_N, max_nChanels), mode(true), nKept(0) {
        for (int i = 0; i < nContractions; ++i) use[i] = false;
    }
    void setContractions(int nKept_) {
        assert(nKept_ > 0 && nKept_ <= nContractions);
        nKept = nKept_;
    }
    void setTrainMode() { mode = true; }
    void setTestMode() { mode = false; }
    void setMode(bool m) { mode = m; }

    void forward() {
        assert(nKept > 0);
        if (mode) {
            for (int i = 0; i < nContractions; ++i) use[i] = false;
            for (int i = 0; i < nKept; ++i)
                for (;;) {
                    const int j = rand() % nContractions;
                    if (!use[j]) {
                        use[j] = true;
                        break;
                    }
                }
        } else {
            for (int i = 0; i < nContractions; ++i) use[i] = true;
        }
        gather(false);
        gf_status st = gfhost::contract18_dropout_forward_host(context(), mask(), mode ? 1.0 : (double)nKept / nContractions,
                                                               &vptr[0], adj->value, value, N, nChanels);
        if (st != GF_OK) gfhost::die(context(), "RisiContraction_18_dropout_hip::forward", st);
        for (int i = 0; i < size; ++i) gradient[i] = 0;
    }

    void backward() {
        assert(nKept > 0);
        assert(mode == true);  // only for train mode (RisiContraction_18_dropout.h:484)
        gather(true);
        gf_status st = gfhost::contract18_dropout_backward_host(context(), mask(), gradient, adj->value, &gptr[0], N, nChanels);
        if (st != GF_OK) gfhost::die(context(), "RisiContraction_18_dropout_hip::backward", st);
    }

    bool mode;  // true = train
    bool use[18];
    int nKept;

private:
    unsigned mask() const {
        unsigned m = 0;
        for (int k = 0; k < nContractions; ++k)
            if (use[k]) m |= 1u << k;
        return m;
    }
};

#endif
