/*
 * smp_port.c -- CPU port (fp64) of one SMP_omega training-step body: the op DAG that
 * SMP_omega::complete_computation_graph builds for a molecule (GraphFlow/SMP_omega.h:607-692), walked forward and
 * backward op by op with the reference-structured loop nests of gf_oracle.c (dense 0/1 selection matrices through
 * MatTensorMul / TensorMatMul, the nnz-gated RisiContraction_18 loops, naive ijk MatMul), plus the batch-parallel
 * driver that mirrors Threaded_BatchLearn (:750-792: one worker per host thread, molecules dealt round-robin, the
 * workers' gradients added serially by the master).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as gf_oracle.c): it is (a) a second, independent restatement of
 * the SMP step -- tests/test_smp_cpu.py holds it against the numpy restatement oracle/smp_oracle.py and against the
 * real-reference goldens -- and (b) the "port"-kind CPU baseline bench.py times on the GPU box, where the reference
 * itself cannot travel.  tools/port_vs_reference.py records, in the build container, how its run time compares with the
 * real reference's (profiles/r02_port_vs_reference.json).
 *
 * Graph preparation (hop distances, WL features, ranking, receptive fields: SMP_omega.h:358-537) is the caller's: it
 * is < 0.1 % of a molecule's cost and oracle/smp_oracle.py already restates it; the timings exclude it.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* op loops of gf_oracle.c */
void gfo_r18_loops_forward(const double *P, const double *A, double *Out, int N, int C);
void gfo_r18_loops_backward(const double *G, const double *A, double *dP, int N, int C);
void gfo_matmul_forward(const double *A, const double *B, double *Cm, int M, int K, int N);
void gfo_matmul_backward(const double *dC, const double *A, const double *B, double *dA, double *dB, int M, int K, int N);
void gfo_mattensormul_forward(const double *X, const double *F, double *Out, int R, int Kd, int J, int D);
void gfo_mattensormul_backward(const double *G, const double *X, const double *F, double *dX, double *dF, int R, int Kd, int J, int D);
void gfo_tensormatmul_forward(const double *F, const double *Y, double *Out, int R, int Kd, int J, int D);
void gfo_tensormatmul_backward(const double *G, const double *F, const double *Y, double *dF, double *dY, int R, int Kd, int J, int D);

#define ALPHA 0.01 /* LeakyReLU3D.h:41 */

static double *zalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static double slope_of(double f, int ext, double tol, long long *overrides, long long *conflicts) {
    const int own = f > 0 ? 1 : -1;
    if (ext == 0 || ext == own) return own > 0 ? 1.0 : ALPHA;
    const double z = f > 0 ? f : -f / ALPHA;
    if (z <= tol) {
        ++*overrides;
        return ext > 0 ? 1.0 : ALPHA;
    }
    ++*conflicts;
    return own > 0 ? 1.0 : ALPHA;
}

typedef struct {
    int s;          /* receptive-field size */
    const int *fld; /* the field: vertex ids */
    double *A;      /* reduced adjacency [s][s] (SMP_omega.h:556-581) */
    double **X;     /* per neighbour a: selection matrix [s][s_w] (:461-474) */
    double **Xt;    /* its transpose [s_w][s] (:549-550) */
    double **T1;    /* per neighbour a: X f_{l-1}[w] = [s][s_w][C]  (MatTensorMul value, kept for TensorMatMul::backward) */
    double *Q;      /* contraction output [s][s][18][C] */
    double *f;      /* level output [s][s][C] (post LeakyReLU; its sign is the sign of the pre-activation) */
    double *df;     /* its gradient */
} port_node;

/* One molecule through the DAG.  phi: [L+1][V][cap+1] ints, slot 0 = size.  Parameters / gradients in registration order
 * H[C][FD], (K_l[18C][C], b_l[C]) l = 1..L, W[C] (SMP_omega.h:289-295); grads is ACCUMULATED into (sum_gradients, :808-820).
 * act_level >= 0 && act_out: copy f[act_level][act_vertex] ([s][s][C]) out (parity checks of the per-level activations).
 * Returns 0, or -1 on allocation failure. */
/* LeakyReLU's kink and fp32.  A pre-activation z within the fp32 rounding error of its own sum (|z| <= kink_tol * max|z| of
 * its level) has no well-defined slope for an fp32 implementation: a different summation order lands it on either side of 0,
 * and either one-sided derivative is a valid subgradient there.  With ext_sign given (one signed char per activation element,
 * nodes in (level, vertex) order, each [s][s][C]: the sign the implementation under test saw) the reverse sweep takes THAT
 * slope at such elements and its own everywhere else; *n_override counts the elements where this changed the slope and
 * *n_conflict the elements OUTSIDE the tolerance whose external sign disagrees (a real forward error: must be 0).
 * acts_out (optional): every f[l][v] back to back in the same order. */
int gfo_smp_molecule_ex(int V, int FD, int L, int C, int cap, const double *x, const int *phi, const int *adj,
                        const double *coulomb, const double *params, double target, int want_grads, double *graph_feature,
                        double *predict, double *loss, double *grads, int act_level, int act_vertex, double *act_out,
                        const signed char *ext_sign, double kink_tol, long long *n_override, long long *n_conflict,
                        double *acts_out);

int gfo_smp_molecule(int V, int FD, int L, int C, int cap, const double *x, const int *phi, const int *adj,
                     const double *coulomb, const double *params, double target, int want_grads, double *graph_feature,
                     double *predict, double *loss, double *grads, int act_level, int act_vertex, double *act_out) {
    return gfo_smp_molecule_ex(V, FD, L, C, cap, x, phi, adj, coulomb, params, target, want_grads, graph_feature, predict, loss,
                               grads, act_level, act_vertex, act_out, NULL, 0.0, NULL, NULL, NULL);
}

int gfo_smp_molecule_ex(int V, int FD, int L, int C, int cap, const double *x, const int *phi, const int *adj,
                        const double *coulomb, const double *params, double target, int want_grads, double *graph_feature,
                        double *predict, double *loss, double *grads, int act_level, int act_vertex, double *act_out,
                        const signed char *ext_sign, double kink_tol, long long *n_override, long long *n_conflict,
                        double *acts_out) {
    const size_t nH = (size_t)C * FD, nK = (size_t)18 * C * C;
    const double *H = params;
    const double *W = params + nH + (size_t)L * (nK + C);
#define KL(l) (params + nH + (size_t)((l)-1) * (nK + C))
#define BL(l) (KL(l) + nK)
#define PHI(l, v) (phi + ((size_t)(l)*V + (v)) * (cap + 1))
    port_node *nodes = (port_node *)calloc((size_t)(L + 1) * V, sizeof(port_node));
    if (!nodes) return -1;
#define ND(l, v) nodes[(size_t)(l)*V + (v)]
    int rc = 0;

    /* level 0 (:617-626): MatMul(H, x_v) -> Reshape3D(1,1,C) -> LeakyReLU3D */
    for (int v = 0; v < V; ++v) {
        port_node *n = &ND(0, v);
        n->s = 1;
        n->fld = PHI(0, v) + 1;
        n->f = zalloc(C);
        n->df = zalloc(C);
        gfo_matmul_forward(H, x + (size_t)v * FD, n->f, C, FD, 1);
        for (int c = 0; c < C; ++c) n->f[c] = n->f[c] > 0 ? n->f[c] : ALPHA * n->f[c];
    }
    for (int l = 1; l <= L; ++l)
        for (int v = 0; v < V; ++v) {
            port_node *n = &ND(l, v);
            const int s = PHI(l, v)[0];
            n->s = s;
            n->fld = PHI(l, v) + 1;
            n->A = zalloc((size_t)s * s);
            for (int i = 0; i < s; ++i)
                for (int j = 0; j < s; ++j) {
                    const int vi = n->fld[i], vj = n->fld[j];
                    n->A[(size_t)i * s + j] = coulomb ? coulomb[(size_t)vi * V + vj] : (vi == vj ? 1.0 : (double)adj[(size_t)vi * V + vj]);
                }
            n->X = (double **)calloc(s, sizeof(double *));
            n->Xt = (double **)calloc(s, sizeof(double *));
            n->T1 = (double **)calloc(s, sizeof(double *));
            double *P = zalloc((size_t)s * s * s * C); /* StackTensor3D-less stack: tensors[a] = P + a s^2 C */
            for (int a = 0; a < s; ++a) {
                const port_node *src = &ND(l - 1, n->fld[a]);
                const int sw = src->s;
                double *X = n->X[a] = zalloc((size_t)s * sw), *Xt = n->Xt[a] = zalloc((size_t)sw * s);
                for (int i = 0; i < s; ++i)
                    for (int k = 0; k < sw; ++k)
                        if (n->fld[i] == src->fld[k]) X[(size_t)i * sw + k] = Xt[(size_t)k * s + i] = 1.0;
                n->T1[a] = zalloc((size_t)s * sw * C);
                gfo_mattensormul_forward(X, src->f, n->T1[a], s, sw, sw, C);                 /* :641-642 */
                gfo_tensormatmul_forward(n->T1[a], Xt, P + (size_t)a * s * s * C, s, sw, s, C); /* :644-645 */
            }
            n->Q = zalloc((size_t)s * s * 18 * C);
            gfo_r18_loops_forward(P, n->A, n->Q, s, C); /* :650-651 */
            free(P);
            n->f = zalloc((size_t)s * s * C);
            n->df = zalloc((size_t)s * s * C);
            gfo_matmul_forward(n->Q, KL(l), n->f, s * s, 18 * C, C); /* Reshape2D + MatMul, :654-657 */
            const double *b = BL(l);
            for (size_t i = 0; i < (size_t)s * s; ++i)
                for (int c = 0; c < C; ++c) {
                    const double z = n->f[i * C + c] + b[c]; /* VectorAddTensor, :663 */
                    n->f[i * C + c] = z > 0 ? z : ALPHA * z; /* LeakyReLU3D, :667 */
                }
        }
    /* element offsets of the nodes in (level, vertex) order, the largest |z| of every level, optional dump */
    size_t *node_off = (size_t *)calloc((size_t)(L + 1) * V + 1, sizeof(size_t));
    double *zmax = zalloc(L + 1);
    {
        size_t o = 0;
        for (int l = 0; l <= L; ++l)
            for (int v = 0; v < V; ++v) {
                const port_node *n = &ND(l, v);
                const size_t cnt = (size_t)n->s * n->s * C;
                node_off[(size_t)l * V + v] = o;
                for (size_t i = 0; i < cnt; ++i) {
                    const double z = n->f[i] > 0 ? n->f[i] : -n->f[i] / ALPHA;
                    if (z > zmax[l]) zmax[l] = z;
                }
                if (acts_out) memcpy(acts_out + o, n->f, sizeof(double) * cnt);
                o += cnt;
            }
    }
    long long overrides = 0, conflicts = 0;
/* slope of element i of node (l, v): the node's own sign, or the external one inside the kink tolerance */
#define SLOPE(l, v, n, i) slope_of((n)->f[i], ext_sign ? ext_sign[node_off[(size_t)(l)*V + (v)] + (i)] : 0, kink_tol * zmax[l], &overrides, &conflicts)
    if (act_out && act_level >= 0 && act_level <= L && act_vertex >= 0 && act_vertex < V) {
        const port_node *n = &ND(act_level, act_vertex);
        memcpy(act_out, n->f, sizeof(double) * (size_t)n->s * n->s * C);
    }
    /* readout (:676-692): ShrinkTensor -> LeakyReLU -> SumVectors -> InnerProduct -> SquaredLoss */
    double *sh = zalloc((size_t)V * C), *g = zalloc(C);
    for (int v = 0; v < V; ++v) {
        const port_node *n = &ND(L, v);
        for (size_t i = 0; i < (size_t)n->s * n->s; ++i)
            for (int c = 0; c < C; ++c) sh[(size_t)v * C + c] += n->f[i * C + c];
        for (int c = 0; c < C; ++c) {
            const double z = sh[(size_t)v * C + c];
            g[c] += z > 0 ? z : ALPHA * z;
        }
    }
    double y = 0.0;
    for (int c = 0; c < C; ++c) y += g[c] * W[c];
    if (graph_feature) memcpy(graph_feature, g, sizeof(double) * C);
    if (predict) *predict = y;
    if (loss) *loss = 0.5 * (y - target) * (y - target);

    if (want_grads && grads) { /* reverse sweep (GraphFlow.h:729-1266: reverse insertion order, every op `+=`) */
        double *dH = grads, *dW = grads + nH + (size_t)L * (nK + C);
        const double dy = y - target; /* SquaredLoss.h:55-61 */
        for (int c = 0; c < C; ++c) dW[c] += dy * g[c];
        for (int v = V - 1; v >= 0; --v) {
            port_node *n = &ND(L, v);
            for (int c = 0; c < C; ++c) {
                const double d = dy * W[c] * (sh[(size_t)v * C + c] > 0 ? 1.0 : ALPHA);
                for (size_t i = 0; i < (size_t)n->s * n->s; ++i) n->df[i * C + c] += d; /* ShrinkTensor.h:52-61 */
            }
        }
        for (int l = L; l >= 1; --l)
            for (int v = V - 1; v >= 0; --v) {
                port_node *n = &ND(l, v);
                const int s = n->s;
                double *dK = grads + nH + (size_t)(l - 1) * (nK + C), *db = dK + nK;
                double *dz = zalloc((size_t)s * s * C);
                for (size_t i = 0; i < (size_t)s * s; ++i)
                    for (int c = 0; c < C; ++c) {
                        const double d = n->df[i * C + c] * SLOPE(l, v, n, i * C + c);
                        dz[i * C + c] = d;
                        db[c] += d; /* VectorAddTensor.h:61-72 */
                    }
                double *dQ = zalloc((size_t)s * s * 18 * C);
                gfo_matmul_backward(dz, n->Q, KL(l), dQ, dK, s * s, 18 * C, C); /* MatMul.h:69-82 */
                free(dz);
                double *dP = zalloc((size_t)s * s * s * C);
                gfo_r18_loops_backward(dQ, n->A, dP, s, C); /* RisiContraction_18.h:333-560 */
                free(dQ);
                for (int a = s - 1; a >= 0; --a) {
                    port_node *src = &ND(l - 1, n->fld[a]);
                    const int sw = src->s;
                    double *dT1 = zalloc((size_t)s * sw * C), *dXt = zalloc((size_t)sw * s), *dX = zalloc((size_t)s * sw);
                    /* the reference also fills the selection matrices' gradients (never read): same work here */
                    gfo_tensormatmul_backward(dP + (size_t)a * s * s * C, n->T1[a], n->Xt[a], dT1, dXt, s, sw, s, C);
                    gfo_mattensormul_backward(dT1, n->X[a], src->f, dX, src->df, s, sw, sw, C);
                    free(dT1);
                    free(dXt);
                    free(dX);
                }
                free(dP);
            }
        for (int v = V - 1; v >= 0; --v) {
            port_node *n = &ND(0, v);
            double dz0[1024], *dzp = C <= 1024 ? dz0 : zalloc(C);
            for (int c = 0; c < C; ++c) dzp[c] = n->df[c] * SLOPE(0, v, n, c);
            double *dx = zalloc(FD);
            gfo_matmul_backward(dzp, H, x + (size_t)v * FD, dH, dx, C, FD, 1);
            free(dx);
            if (dzp != dz0) free(dzp);
        }
    }
    if (n_override) *n_override = overrides;
    if (n_conflict) *n_conflict = conflicts;
    free(node_off);
    free(zmax);
    free(sh);
    free(g);
    for (int l = 0; l <= L; ++l)
        for (int v = 0; v < V; ++v) {
            port_node *n = &ND(l, v);
            if (n->X)
                for (int a = 0; a < n->s; ++a) {
                    free(n->X[a]);
                    free(n->Xt[a]);
                    free(n->T1[a]);
                }
            free(n->X);
            free(n->Xt);
            free(n->T1);
            free(n->A);
            free(n->Q);
            free(n->f);
            free(n->df);
        }
    free(nodes);
    return rc;
#undef KL
#undef BL
#undef PHI
#undef ND
}

/* ---- batch-parallel driver: the shape of SMP_omega::Threaded_BatchLearn (SMP_omega.h:750-792) --------------------------
 * One model clone per worker thread (:115-129).  Molecules are given back to back: x [sum V][FD] with molecule m at
 * x_off[m] (doubles), its fields at phi_off[m] (ints, [L+1][V][cap+1]), its adjacency at adj_off[m] (ints, [V][V]).   */
typedef struct {
    int m, FD, L, C, cap;
    const int *nV, *phi, *adj;
    const long long *x_off, *phi_off, *adj_off;
    const double *x, *params, *targets;
    double *predict, *loss, *grads;
    int rc;
} port_job;

static void *port_worker(void *arg) { /* compute_gradient_job (:742-748) */
    port_job *j = (port_job *)arg;
    const int m = j->m;
    j->rc = gfo_smp_molecule(j->nV[m], j->FD, j->L, j->C, j->cap, j->x + j->x_off[m], j->phi + j->phi_off[m],
                             j->adj + j->adj_off[m], NULL, j->params, j->targets[m], 1, NULL, &j->predict[m], &j->loss[m],
                             j->grads, -1, -1, NULL);
    return NULL;
}

/* The batch goes through in waves of nThreads molecules: one std::thread per molecule of the wave (:768-771), joined
 * (:773-775), then the wave's gradients added to the master's one clone after the other (:777-779).  nThreads == 1 is the
 * serial gradient loop of BatchLearn (:808-818). */
int gfo_smp_batch(int nMol, const int *nV, int FD, int L, int C, int cap, const double *x, const long long *x_off,
                  const int *phi, const long long *phi_off, const int *adj, const long long *adj_off, const double *params,
                  const double *targets, int nThreads, double *predict, double *loss, double *grads) {
    const size_t np = (size_t)C * FD + (size_t)L * ((size_t)18 * C * C + C) + C;
    if (nThreads < 1) nThreads = 1;
    pthread_t *th = (pthread_t *)calloc(nThreads, sizeof(pthread_t));
    port_job *jobs = (port_job *)calloc(nThreads, sizeof(port_job));
    double **clone = (double **)calloc(nThreads, sizeof(double *));
    for (int t = 0; t < nThreads; ++t) clone[t] = zalloc(np);
    int rc = 0;
    for (int start = 0; start < nMol; start += nThreads) {
        const int nRuns = (nMol - start < nThreads) ? nMol - start : nThreads;
        for (int t = 0; t < nRuns; ++t) {
            port_job j = {start + t, FD, L, C, cap, nV, phi, adj, x_off, phi_off, adj_off, x, params, targets, predict, loss, clone[t], 0};
            memset(clone[t], 0, sizeof(double) * np); /* a parameter's forward() zeroes its gradient (Vector.h:28-32) */
            jobs[t] = j;
            if (nThreads == 1)
                port_worker(&jobs[t]);
            else
                pthread_create(&th[t], NULL, port_worker, &jobs[t]);
        }
        for (int t = 0; t < nRuns; ++t) {
            if (nThreads > 1) pthread_join(th[t], NULL);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
        for (int t = 0; t < nRuns; ++t)
            for (size_t i = 0; i < np; ++i) grads[i] += clone[t][i];
    }
    for (int t = 0; t < nThreads; ++t) free(clone[t]);
    free(clone);
    free(th);
    free(jobs);
    return rc;
}

/* ---- batch-parallel RisiContraction_18: the "best-effort CPU" of SURVEY 8(d)(iii) for the pure-op workload ------------------
 * nGraphs independent graphs, forward + backward each (the nnz-gated single-thread loops), one graph per host thread at a
 * time -- the batch-parallel shape Threaded_BatchLearn gives the SMP model, applied to the op benchmark.  All graphs share
 * the same P / A / G buffers (read-only); each thread owns its Out / dP.                                                   */
typedef struct {
    const double *P, *A, *G;
    int N, C, count;
} r18_batch_job;

static void *r18_batch_worker(void *arg) {
    r18_batch_job *j = (r18_batch_job *)arg;
    const size_t nOut = (size_t)j->N * j->N * 18 * j->C, nP = (size_t)j->N * j->N * j->N * j->C;
    double *Out = zalloc(nOut), *dP = zalloc(nP);
    for (int i = 0; i < j->count; ++i) {
        gfo_r18_loops_forward(j->P, j->A, Out, j->N, j->C);
        gfo_r18_loops_backward(j->G, j->A, dP, j->N, j->C);
    }
    free(Out);
    free(dP);
    return NULL;
}

int gfo_r18_batch_threads(const double *P, const double *A, const double *G, int N, int C, int nGraphs, int nThreads) {
    if (nThreads < 1) nThreads = 1;
    if (nThreads > nGraphs) nThreads = nGraphs;
    pthread_t *th = (pthread_t *)calloc(nThreads, sizeof(pthread_t));
    r18_batch_job *jobs = (r18_batch_job *)calloc(nThreads, sizeof(r18_batch_job));
    for (int t = 0; t < nThreads; ++t) {
        r18_batch_job j = {P, A, G, N, C, nGraphs / nThreads + (t < nGraphs % nThreads ? 1 : 0)};
        jobs[t] = j;
        pthread_create(&th[t], NULL, r18_batch_worker, &jobs[t]);
    }
    for (int t = 0; t < nThreads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}
