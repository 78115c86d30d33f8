// Mixers_hip.h -- drop-in Entity-style ops MatMul_hip, MatTensorMul_hip, TensorMatMul_hip, CustomMatMulTensor_hip,
// StackTensor3D_hip.
//
// Same contracts as GraphFlow/MatMul.h:21-89, MatTensorMul.h:22-92, TensorMatMul.h:22-91, StackTensor3D.h:25-98
// (and GraphFlow_gpu/MatMul_gpu.h:113-505 for the GPU flavour): ctor with maximum dims allocates; setParameter()
// rebinds the two inputs and the logical dims; forward() overwrites value then zeroes own gradient; backward()
// accumulates into BOTH inputs' gradients.  All arithmetic runs in libgf_hip.so (host-pointer mode of the C ABI).
#ifndef MIXERS_HIP_H_INCLUDED
#define MIXERS_HIP_H_INCLUDED

#include <cassert>
#include <vector>

#include "gf_containers.h"
#include "gf_runtime.h"

namespace gfhost {
template <class V>
inline void zero_gradient(V *v) {
    for (int i = 0; i < v->size; ++i) v->gradient[i] = 0;
}
inline void must(gf_status st, const char *where) {
    if (st != GF_OK) die(default_context(), where, st);
}
}  // namespace gfhost

// C = first * second
class MatMul_hip : public Matrix {
public:
    MatMul_hip(int max_first_nRows, int /*max_first_nColumns*/, int /*max_second_nRows*/, int max_second_nColumns)
        : Matrix(max_first_nRows, max_second_nColumns), first(NULL), second(NULL) {}
    MatMul_hip(int max_nRows, int max_nColumns) : Matrix(max_nRows, max_nColumns), first(NULL), second(NULL) {}
    MatMul_hip(Matrix *a, Matrix *b) : Matrix(a->nRows, b->nColumns), first(NULL), second(NULL) { setParameter(a, b); }

    void setParameter(Matrix *a, Matrix *b) {
        assert(a->nColumns == b->nRows);
        first = a;
        second = b;
        Matrix::setParameter(a->nRows, b->nColumns);
    }
    void forward() {
        gfhost::must(gfhost::matmul_forward_host(gfhost::default_context(), first->value, second->value, value, nRows,
                                                 first->nColumns, nColumns), "MatMul_hip::forward");
        gfhost::zero_gradient(this);
    }
    void backward() {
        gfhost::must(gfhost::matmul_backward_host(gfhost::default_context(), gradient, first->value, second->value,
                                                  first->gradient, second->gradient, nRows, first->nColumns, nColumns),
                     "MatMul_hip::backward");
    }
    Matrix *first, *second;
};

// Out[i,j,d] = sum_k first[i,k] * second[k,j,d]
class MatTensorMul_hip : public Tensor3D {
public:
    MatTensorMul_hip(int max_nRows, int max_nColumns, int max_nDepth)
        : Tensor3D(max_nRows, max_nColumns, max_nDepth), first(NULL), second(NULL) {}
    MatTensorMul_hip(Matrix *a, Tensor3D *b) : Tensor3D(a->nRows, b->nColumns, b->nDepth), first(NULL), second(NULL) {
        setParameter(a, b);
    }
    void setParameter(Matrix *a, Tensor3D *b) {
        assert(a->nColumns == b->nRows);
        first = a;
        second = b;
        Tensor3D::setParameter(a->nRows, b->nColumns, b->nDepth);
    }
    void forward() {
        gfhost::must(gfhost::mattensormul_forward_host(gfhost::default_context(), first->value, second->value, value, nRows,
                                                       first->nColumns, nColumns, nDepth), "MatTensorMul_hip::forward");
        gfhost::zero_gradient(this);
    }
    void backward() {
        gfhost::must(gfhost::mattensormul_backward_host(gfhost::default_context(), gradient, first->value, second->value,
                                                        first->gradient, second->gradient, nRows, first->nColumns,
                                                        nColumns, nDepth), "MatTensorMul_hip::backward");
    }
    Matrix *first;
    Tensor3D *second;
};

// Out[i,j,d] = sum_k first[i,k,d] * second[k,j]
class TensorMatMul_hip : public Tensor3D {
public:
    TensorMatMul_hip(int max_nRows, int max_nColumns, int max_nDepth)
        : Tensor3D(max_nRows, max_nColumns, max_nDepth), first(NULL), second(NULL) {}
    TensorMatMul_hip(Tensor3D *a, Matrix *b) : Tensor3D(a->nRows, b->nColumns, a->nDepth), first(NULL), second(NULL) {
        setParameter(a, b);
    }
    void setParameter(Tensor3D *a, Matrix *b) {
        assert(a->nColumns == b->nRows);
        first = a;
        second = b;
        Tensor3D::setParameter(a->nRows, b->nColumns, a->nDepth);
    }
    void forward() {
        gfhost::must(gfhost::tensormatmul_forward_host(gfhost::default_context(), first->value, second->value, value, nRows,
                                                       first->nColumns, nColumns, nDepth), "TensorMatMul_hip::forward");
        gfhost::zero_gradient(this);
    }
    void backward() {
        gfhost::must(gfhost::tensormatmul_backward_host(gfhost::default_context(), gradient, first->value, second->value,
                                                        first->gradient, second->gradient, nRows, first->nColumns,
                                                        nColumns, nDepth), "TensorMatMul_hip::backward");
    }
    Tensor3D *first;
    Matrix *second;
};

// Out[i,j,k] = sum_v first[k,v] * second[i,j,v]   (channel mix; GraphFlow/CustomMatMulTensor.h:22-91)
class CustomMatMulTensor_hip : public Tensor3D {
public:
    CustomMatMulTensor_hip(int max_nRows, int max_nColumns, int max_nDepth)
        : Tensor3D(max_nRows, max_nColumns, max_nDepth), first(NULL), second(NULL) {}
    CustomMatMulTensor_hip(Matrix *a, Tensor3D *b) : Tensor3D(b->nRows, b->nColumns, a->nRows), first(NULL), second(NULL) {
        setParameter(a, b);
    }
    void setParameter(Matrix *a, Tensor3D *b) {
        assert(a->nColumns == b->nDepth);
        first = a;
        second = b;
        Tensor3D::setParameter(b->nRows, b->nColumns, a->nRows);
    }
    void forward() {
        gfhost::must(gfhost::custommatmultensor_forward_host(gfhost::default_context(), first->value, second->value, value,
                                                             (long long)nRows * nColumns, second->nDepth, nDepth),
                     "CustomMatMulTensor_hip::forward");
        gfhost::zero_gradient(this);
    }
    void backward() {
        gfhost::must(gfhost::custommatmultensor_backward_host(gfhost::default_context(), gradient, first->value,
                                                              second->value, first->gradient, second->gradient,
                                                              (long long)nRows * nColumns, second->nDepth, nDepth),
                     "CustomMatMulTensor_hip::backward");
    }
    Matrix *first;
    Tensor3D *second;
};

// value[row][col][c1][c2] = tensors[row]->value[col][c1][c2].  Only needed to feed ops that want one contiguous
// Tensor4D; RisiContraction_*_hip accepts add_tensor() directly and does not need it.
class StackTensor3D_hip : public Tensor4D {
public:
    StackTensor3D_hip(int rows, int cols, int c1, int c2) : Tensor4D(rows, cols, c1, c2) {}
    void setParameter(int rows, int cols, int c1, int c2) {
        nRows = rows;
        nColumns = cols;
        nChanels1 = c1;
        nChanels2 = c2;
        size = rows * cols * c1 * c2;
        tensors.clear();
    }
    void add_tensor(Tensor3D *t) {
        assert(t->nRows == nColumns && t->nColumns == nChanels1 && t->nDepth == nChanels2);
        tensors.push_back(t);
    }
    void clear() { tensors.clear(); }
    void forward() {
        assert((int)tensors.size() == nRows);
        std::vector<const gf_real *> p(nRows);
        for (int r = 0; r < nRows; ++r) p[r] = tensors[r]->value;
        gfhost::must(gfhost::stack_forward_host(gfhost::default_context(), &p[0], value, nRows,
                                                (size_t)nColumns * nChanels1 * nChanels2), "StackTensor3D_hip::forward");
        gfhost::zero_gradient(this);
    }
    void backward() {
        assert((int)tensors.size() == nRows);
        std::vector<gf_real *> p(nRows);
        for (int r = 0; r < nRows; ++r) p[r] = tensors[r]->gradient;
        gfhost::must(gfhost::stack_backward_host(gfhost::default_context(), gradient, &p[0], nRows,
                                                 (size_t)nColumns * nChanels1 * nChanels2), "StackTensor3D_hip::backward");
    }
    std::vector<Tensor3D *> tensors;
};

#endif
