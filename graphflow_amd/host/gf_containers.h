// gf_containers.h -- value containers with the reference's data ABI.
//
// The op classes in this directory only touch the PUBLIC FIELDS of the reference containers
// (GraphFlow/Vector.h:22-44, Matrix.h:23-36, Tensor3D.h:23-44, Tensor4D.h:23-33):
//     size, value, gradient, nRows, nColumns, nDepth, nChanels1, nChanels2, index(...)
// so they compile unchanged against the real GraphFlow headers.  When those headers are NOT on the include path
// (the GPU box, our own tests) this file supplies equivalent containers.  If a reference header was already
// included (its include guard is defined) the matching definition below is skipped.
//
// GF_REAL selects the scalar: double (GraphFlow/, the default) or float (GraphFlow_32bit/).
#ifndef GF_CONTAINERS_H_INCLUDED
#define GF_CONTAINERS_H_INCLUDED

#include <cstddef>

#ifndef GF_REAL
#define GF_REAL double
#endif
typedef GF_REAL gf_real;

#ifndef __ENTITY_H_INCLUDED__
#define __ENTITY_H_INCLUDED__
class Entity {};  // opaque handle type of the executor (GraphFlow/Entity.h:10-16)
#endif

#ifndef __VECTOR_H_INCLUDED__
#define __VECTOR_H_INCLUDED__
// Owns `value` and `gradient`, allocated once for the maximum size; setParameter() on the derived types only
// changes the logical extent.  forward() = "zero my gradient", backward() = nothing: that is what makes a plain
// container usable as a parameter / input node of the executor.
class Vector : public Entity {
public:
    explicit Vector(int n) : size(n), value(new gf_real[n > 0 ? n : 1]), gradient(new gf_real[n > 0 ? n : 1]) {}
    ~Vector() {
        delete[] value;
        delete[] gradient;
    }
    void forward() {
        for (int i = 0; i < size; ++i) gradient[i] = 0;
    }
    void backward() {}

    int size;
    gf_real *value;
    gf_real *gradient;

private:
    Vector(const Vector &);
    Vector &operator=(const Vector &);
};
#endif

#ifndef __MATRIX_H_INCLUDED__
#define __MATRIX_H_INCLUDED__
class Matrix : public Vector {
public:
    Matrix(int rows, int cols) : Vector(rows * cols), nRows(rows), nColumns(cols) {}
    void setParameter(int rows, int cols) {
        nRows = rows;
        nColumns = cols;
        size = rows * cols;
    }
    int index(int r, int c) const { return r * nColumns + c; }
    int nRows, nColumns;
};
#endif

#ifndef __TENSOR3D_H_INCLUDED__
#define __TENSOR3D_H_INCLUDED__
class Tensor3D : public Vector {
public:
    Tensor3D(int rows, int cols, int depth) : Vector(rows * cols * depth), nRows(rows), nColumns(cols), nDepth(depth) {}
    void setParameter(int rows, int cols, int depth) {
        nRows = rows;
        nColumns = cols;
        nDepth = depth;
        size = rows * cols * depth;
    }
    int index(int r, int c, int d) const { return (r * nColumns + c) * nDepth + d; }
    int nRows, nColumns, nDepth;
};
#endif

#ifndef __TENSOR4D_H_INCLUDED__
#define __TENSOR4D_H_INCLUDED__
class Tensor4D : public Vector {
public:
    Tensor4D(int rows, int cols, int c1, int c2)
        : Vector(rows * cols * c1 * c2), nRows(rows), nColumns(cols), nChanels1(c1), nChanels2(c2) {}
    int index(int r, int c, int k1, int k2) const { return ((r * nColumns + c) * nChanels1 + k1) * nChanels2 + k2; }
    int nRows, nColumns, nChanels1, nChanels2;
};
#endif

#endif  // GF_CONTAINERS_H_INCLUDED
