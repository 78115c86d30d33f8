/*
 * gf_oracle.c -- CPU restatement (fp64) of GraphFlow's second-order CCN/SMP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped path is the HIP
 * library in graphflow_amd/csrc and never links or calls anything in here.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * golden vectors captured from the real reference (oracle/ref_shim.cpp compiled against
 * /root/reference/GraphFlow, generator tests/golden/make_golden.py) and, when
 * oracle/_ref/libgf_ref.so is present, against the reference itself on random inputs.
 *
 * Layouts (all row-major, fp64 like the reference's GraphFlow/ variant):
 *   P   [N][N][N][C]   P[a][b][c][f] = tensors[a]->value[(b*N+c)*C+f]   (RisiContraction_18.h:48-53)
 *   A   [N][N]         adj->value[d*N+e]                                   (Matrix.h:34-36)
 *   Out [N][N][K][C]   value[(x*N+y)*K*C + k*C + f]                        (RisiContraction_18.h:29,103)
 *
 * Two independent statements of the contractions are given on purpose:
 *   (1) a table-driven "spec" form (gfo_contract_*): every case is (kept pair, tie classes)
 *       over the five indices a,b,c,d,e -- follows the case comments of
 *       RisiContraction_50.h:94-430, RisiContraction_10.h:94-142, RisiContraction_18.h:98-322;
 *   (2) loop-nest forms of RisiContraction_18 with the reference's loop order, A>0 gate and
 *       nnz skipping (gfo_r18_loops_*: RisiContraction_18.h:73-331, 333-560) and the 6-way
 *       case-group split of RisiContraction_18_thread.h:79-408 (gfo_r18_thread_forward).
 *       These are what bench.py times as the CPU baseline.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define IDX_A 0
#define IDX_B 1
#define IDX_C 2
#define IDX_D 3
#define IDX_E 4

/* One contraction case: output indexed by (keep0, keep1); cls[i] = equivalence class of index i
 * (indices in the same class are tied, i.e. forced equal); every class that does not contain a
 * kept index is summed over. */
typedef struct {
    int keep0, keep1;
    int cls[5];
    int ncls;
} gfo_case;

static void case_init(gfo_case *k, int keep0, int keep1) {
    k->keep0 = keep0;
    k->keep1 = keep1;
    for (int i = 0; i < 5; ++i) k->cls[i] = i;
    k->ncls = 5;
}

static void case_tie(gfo_case *k, int i, int j) {
    int from = k->cls[j], to = k->cls[i];
    if (from == to) return;
    for (int t = 0; t < 5; ++t)
        if (k->cls[t] == from) k->cls[t] = to;
}

static void case_compact(gfo_case *k) {
    int map[5] = {-1, -1, -1, -1, -1}, n = 0;
    for (int i = 0; i < 5; ++i) {
        if (map[k->cls[i]] < 0) map[k->cls[i]] = n++;
    }
    for (int i = 0; i < 5; ++i) k->cls[i] = map[k->cls[i]];
    k->ncls = n;
}

/* The 50 cases in the order of RisiContraction_50.h:94-430:
 *   1..10   kept pair in lexicographic order over {a,b,c,d,e}, nothing tied ("1+1+1")
 *   11..40  per kept pair, the three lexicographic ways of tying two of the other three ("1+2")
 *   41..50  per kept pair, all three others tied ("3")                                         */
static void build_50(gfo_case *out) {
    int n = 0;
    int pairs[10][2], np = 0;
    for (int i = 0; i < 5; ++i)
        for (int j = i + 1; j < 5; ++j) {
            pairs[np][0] = i;
            pairs[np][1] = j;
            ++np;
        }
    for (int p = 0; p < 10; ++p) {
        case_init(&out[n], pairs[p][0], pairs[p][1]);
        case_compact(&out[n]);
        ++n;
    }
    for (int p = 0; p < 10; ++p) {
        int rest[3], nr = 0;
        for (int i = 0; i < 5; ++i)
            if (i != pairs[p][0] && i != pairs[p][1]) rest[nr++] = i;
        static const int tw[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int t = 0; t < 3; ++t) {
            case_init(&out[n], pairs[p][0], pairs[p][1]);
            case_tie(&out[n], rest[tw[t][0]], rest[tw[t][1]]);
            case_compact(&out[n]);
            ++n;
        }
    }
    for (int p = 0; p < 10; ++p) {
        int rest[3], nr = 0;
        for (int i = 0; i < 5; ++i)
            if (i != pairs[p][0] && i != pairs[p][1]) rest[nr++] = i;
        case_init(&out[n], pairs[p][0], pairs[p][1]);
        case_tie(&out[n], rest[0], rest[1]);
        case_tie(&out[n], rest[0], rest[2]);
        case_compact(&out[n]);
        ++n;
    }
}

/* "(k/50)" labels in the comments of RisiContraction_18.h:98-322. */
static const int R18_OF_50[18] = {1, 3, 5, 6, 10, 11, 13, 17, 18, 23, 26, 27, 28, 38, 40, 43, 46, 50};

int gfo_r18_case_of_50(int k) { return (k >= 0 && k < 18) ? R18_OF_50[k] : -1; }

static int family_cases(int K, gfo_case *cases) {
    gfo_case all[50];
    build_50(all);
    if (K == 50) {
        memcpy(cases, all, sizeof(all));
        return 50;
    }
    if (K == 10) { /* RisiContraction_10.h:94-142 = the ten "1+1+1" cases */
        memcpy(cases, all, 10 * sizeof(gfo_case));
        return 10;
    }
    if (K == 18) {
        for (int k = 0; k < 18; ++k) cases[k] = all[R18_OF_50[k] - 1];
        return 18;
    }
    return 0;
}

/* Enumerate every assignment of the ncls classes (an odometer over N^ncls tuples) and either
 * accumulate Out (forward) or scatter into dP (backward: dP += G * A, RisiContraction_50.h:76-78). */
static void run_case(const gfo_case *k, int kslot, int K, int N, int C, const double *A,
                     const double *P, double *Out, const double *G, double *dP) {
    int val[5] = {0, 0, 0, 0, 0};
    const size_t NC = (size_t)N * C;
    for (;;) {
        const int a = val[k->cls[IDX_A]], b = val[k->cls[IDX_B]], c = val[k->cls[IDX_C]];
        const int d = val[k->cls[IDX_D]], e = val[k->cls[IDX_E]];
        const int ids[5] = {a, b, c, d, e};
        const int x = ids[k->keep0], y = ids[k->keep1];
        const double adj = A[(size_t)d * N + e];
        const size_t pofs = ((size_t)a * N + b) * NC + (size_t)c * C;
        const size_t oofs = ((size_t)x * N + y) * K * C + (size_t)kslot * C;
        if (Out) {
            for (int f = 0; f < C; ++f) Out[oofs + f] += P[pofs + f] * adj;
        } else {
            for (int f = 0; f < C; ++f) dP[pofs + f] += G[oofs + f] * adj;
        }
        int i = k->ncls - 1;
        while (i >= 0 && ++val[i] == N) val[i--] = 0;
        if (i < 0) break;
    }
}

/* K in {10,18,50}.  gate != 0 applies RisiContraction_18's `if (adj_value > 0)` (RisiContraction_18.h:90,345),
 * i.e. A is replaced by A+ = A*[A>0]; _10 and _50 have no gate (RisiContraction_50.h:83-97). */
static double *gated_copy(const double *A, int N, int gate) {
    double *Ag = (double *)malloc(sizeof(double) * (size_t)N * N);
    for (int i = 0; i < N * N; ++i) Ag[i] = (gate && !(A[i] > 0)) ? 0.0 : A[i];
    return Ag;
}

int gfo_contract_forward(int K, const double *P, const double *A, double *Out, int N, int C) {
    gfo_case cases[50];
    const int n = family_cases(K, cases);
    if (n == 0) return -1;
    double *Ag = gated_copy(A, N, K == 18);
    memset(Out, 0, sizeof(double) * (size_t)N * N * K * C); /* forward() zeroes value first (:76-78) */
    for (int k = 0; k < n; ++k) run_case(&cases[k], k, K, N, C, Ag, P, Out, NULL, NULL);
    free(Ag);
    return 0;
}

/* dP is accumulated into (`+=`), as every backward() in the reference does. */
int gfo_contract_backward(int K, const double *G, const double *A, double *dP, int N, int C) {
    gfo_case cases[50];
    const int n = family_cases(K, cases);
    if (n == 0) return -1;
    double *Ag = gated_copy(A, N, K == 18);
    for (int k = 0; k < n; ++k) run_case(&cases[k], k, K, N, C, Ag, NULL, NULL, G, dP);
    free(Ag);
    return 0;
}

/* RisiContraction_4 (RisiContraction_4.h:68-125 forward, :127-173 backward): no adjacency.
 *   k=0 (a,b) sum c | k=1 (b,c) sum a | k=2 (a,c) with a==b | k=3 (a,b) with b==c          */
void gfo_r4_forward(const double *P, double *Out, int N, int C) {
    const size_t NC = (size_t)N * C;
    memset(Out, 0, sizeof(double) * (size_t)N * N * 4 * C);
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b)
            for (int c = 0; c < N; ++c)
                for (int f = 0; f < C; ++f) {
                    const double v = P[((size_t)a * N + b) * NC + (size_t)c * C + f];
                    Out[((size_t)a * N + b) * 4 * C + 0 * C + f] += v;
                    Out[((size_t)b * N + c) * 4 * C + 1 * C + f] += v;
                    if (a == b) Out[((size_t)a * N + c) * 4 * C + 2 * C + f] += v;
                    if (b == c) Out[((size_t)a * N + b) * 4 * C + 3 * C + f] += v;
                }
}

void gfo_r4_backward(const double *G, double *dP, int N, int C) {
    const size_t NC = (size_t)N * C;
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b)
            for (int c = 0; c < N; ++c)
                for (int f = 0; f < C; ++f) {
                    double g = G[((size_t)a * N + b) * 4 * C + 0 * C + f] +
                               G[((size_t)b * N + c) * 4 * C + 1 * C + f];
                    if (a == b) g += G[((size_t)a * N + c) * 4 * C + 2 * C + f];
                    if (b == c) g += G[((size_t)a * N + b) * 4 * C + 3 * C + f];
                    dP[((size_t)a * N + b) * NC + (size_t)c * C + f] += g;
                }
}

/* ------------------------------------------------------------------------------------------
 * Loop-nest form of RisiContraction_18 -- same loop order, gate and nnz skipping as the
 * reference (forward RisiContraction_18.h:73-331, backward :333-560).  This is the single-thread
 * CPU baseline.  `dir` = 0 forward (Out += P*A), 1 backward (dP += G*A).
 * ------------------------------------------------------------------------------------------ */
#define PIX(a, b, c, f) ((((size_t)(a) * N + (b)) * N + (c)) * C + (f))
#define OIX(x, y, k, f) ((((size_t)(x) * N + (y)) * 18 + (k)) * C + (f))
#define ACC(o, p)                          \
    do {                                   \
        if (dir == 0)                      \
            Out[o] += P[p] * w;            \
        else                               \
            dP[p] += G[o] * w;             \
    } while (0)

static void r18_loops(int dir, const double *P, const double *A, double *Out, const double *G,
                      double *dP, int N, int C) {
    int a, b, c, d, e, f;
    for (d = 0; d < N; ++d) {
        for (e = 0; e < N; ++e) {
            const double w = A[(size_t)d * N + e];
            if (!(w > 0)) continue;
            /* cases 1-5: the N^3*C block executed once per non-zero of A (:91-124) */
            for (f = 0; f < C; ++f)
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b)
                        for (c = 0; c < N; ++c) {
                            const size_t p = PIX(a, b, c, f);
                            ACC(OIX(a, b, 0, f), p);
                            ACC(OIX(a, d, 1, f), p);
                            ACC(OIX(b, c, 2, f), p);
                            ACC(OIX(b, d, 3, f), p);
                            ACC(OIX(d, e, 4, f), p);
                        }
            for (f = 0; f < C; ++f) /* case 6: c == d (:126-139) */
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b) ACC(OIX(a, b, 5, f), PIX(a, b, d, f));
            if (d == e) /* case 7: d == e (:141-157) */
                for (f = 0; f < C; ++f)
                    for (a = 0; a < N; ++a)
                        for (b = 0; b < N; ++b)
                            for (c = 0; c < N; ++c) ACC(OIX(a, b, 6, f), PIX(a, b, c, f));
            for (f = 0; f < C; ++f) /* case 8: b == c (:159-171) */
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b) ACC(OIX(a, d, 7, f), PIX(a, b, b, f));
            for (f = 0; f < C; ++f) /* case 9: b == e (:173-186) */
                for (a = 0; a < N; ++a)
                    for (c = 0; c < N; ++c) ACC(OIX(a, d, 8, f), PIX(a, e, c, f));
            for (f = 0; f < C; ++f) /* case 10: a == d (:188-202) */
                for (b = 0; b < N; ++b)
                    for (c = 0; c < N; ++c) ACC(OIX(b, c, 9, f), PIX(d, b, c, f));
            for (f = 0; f < C; ++f) /* case 11: a == c (:204-217) */
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b) ACC(OIX(b, d, 10, f), PIX(a, b, a, f));
            for (f = 0; f < C; ++f) /* case 12: a == e (:219-232) */
                for (b = 0; b < N; ++b)
                    for (c = 0; c < N; ++c) ACC(OIX(b, d, 11, f), PIX(e, b, c, f));
            for (f = 0; f < C; ++f) /* case 13: c == e (:234-247) */
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b) ACC(OIX(b, d, 12, f), PIX(a, b, e, f));
            for (f = 0; f < C; ++f) /* case 14: a == b (:249-262) */
                for (a = 0; a < N; ++a)
                    for (c = 0; c < N; ++c) ACC(OIX(d, e, 13, f), PIX(a, a, c, f));
            for (f = 0; f < C; ++f) /* case 15: b == c (:264-277) */
                for (a = 0; a < N; ++a)
                    for (b = 0; b < N; ++b) ACC(OIX(d, e, 14, f), PIX(a, b, b, f));
            for (f = 0; f < C; ++f) /* case 16: b == c == e (:279-295) */
                for (a = 0; a < N; ++a) ACC(OIX(a, d, 15, f), PIX(a, e, e, f));
            for (f = 0; f < C; ++f) /* case 17: a == c == e (:297-309) */
                for (b = 0; b < N; ++b) ACC(OIX(b, d, 16, f), PIX(e, b, e, f));
            for (f = 0; f < C; ++f) /* case 18: a == b == c (:311-323) */
                for (a = 0; a < N; ++a) ACC(OIX(d, e, 17, f), PIX(a, a, a, f));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * RisiContraction_50 as the reference runs it (RisiContraction_50.h:73-441 forward, :443-802 backward): per channel ONE loop nest
 * over (a, b, c, d, e) with all 50 predicated updates inside, value_at = P[a][b][c][f] * A[d][e] (:65-67), no adjacency gate.
 * The CPU baseline of cfg5 (bench.py); bit-identical to the spec form above on the goldens (tests/test_oracle_golden.py). */
void gfo_r50_loops_forward(const double *P, const double *A, double *Out, int N, int C) {
    const size_t NC = (size_t)N * C;
    memset(Out, 0, sizeof(double) * (size_t)N * N * 50 * C);
    for (int f = 0; f < C; ++f)
        for (int a = 0; a < N; ++a)
            for (int b = 0; b < N; ++b)
                for (int c = 0; c < N; ++c)
                    for (int d = 0; d < N; ++d)
                        for (int e = 0; e < N; ++e) {
                            const double v = P[((size_t)a * N + b) * NC + (size_t)c * C + f] * A[(size_t)d * N + e];
#define X(k, i, j) Out[((size_t)(i) * N + (j)) * 50 * C + (size_t)(k) * C + f] += v
#include "gf_oracle_r50_cases.inc"
#undef X
                        }
}

void gfo_r50_loops_backward(const double *G, const double *A, double *dP, int N, int C) {
    const size_t NC = (size_t)N * C;
    for (int f = 0; f < C; ++f)
        for (int a = 0; a < N; ++a)
            for (int b = 0; b < N; ++b)
                for (int c = 0; c < N; ++c)
                    for (int d = 0; d < N; ++d)
                        for (int e = 0; e < N; ++e) {
                            const double w = A[(size_t)d * N + e];
                            double *dst = &dP[((size_t)a * N + b) * NC + (size_t)c * C + f];
#define X(k, i, j) *dst += G[((size_t)(i) * N + (j)) * 50 * C + (size_t)(k) * C + f] * w
#include "gf_oracle_r50_cases.inc"
#undef X
                        }
}

void gfo_r18_loops_forward(const double *P, const double *A, double *Out, int N, int C) {
    memset(Out, 0, sizeof(double) * (size_t)N * N * 18 * C);
    r18_loops(0, P, A, Out, NULL, NULL, N, C);
}

void gfo_r18_loops_backward(const double *G, const double *A, double *dP, int N, int C) {
    r18_loops(1, NULL, A, NULL, G, dP, N, C);
}

/* ------------------------------------------------------------------------------------------
 * 6-thread forward of RisiContraction_18_thread (RisiContraction_18_thread.h:79-408, 743-765):
 * case groups {1-3},{4-6},{7-9},{10-12},{13-15},{16-18} on one thread each, NO A>0 gate and no
 * nnz skipping (every case walks its full index space through value_at, :70-72).  The outputs of
 * the six groups are disjoint slices, so the forward is race-free.  (The reference's threaded
 * backward is racy -- SURVEY.md section 0-7 -- and is deliberately not restated.)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const double *P, *A;
    double *Out;
    int N, C, k0, k1;
} r18_job;

/* value_at (:70-72): the product is formed anew at every use and the neighbour tensor is reached through the op's
 * pointer table (tensors[a] -> value), as in the reference */
#define VAL(a, b, c, d, e, f) (T[a][((b) * N + (c)) * C + (f)] * A[(d) * N + (e)])
#define TIX(x, y, k, f) (((x) * N + (y)) * (18 * C) + (k) * C + (f)) /* Tensor3D::index with int arithmetic, as the reference */

static void *r18_thread_job(void *arg) {
    r18_job *j = (r18_job *)arg;
    const double *P = j->P, *A = j->A;
    double *Out = j->Out;
    const int N = j->N, C = j->C;
    const double *T[64]; /* (N <= 64: the thread variant is only ever timed at the cfg2 shape) */
    for (int a = 0; a < N && a < 64; ++a) T[a] = P + (size_t)a * N * N * C;
    if (j->k0 == 0 && N <= 64) { /* job 0 (:79-113): cases 1-3 share ONE f-a-b-c-d-e nest, three updates per iteration */
        for (int f = 0; f < C; ++f)
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b)
                    for (int c = 0; c < N; ++c)
                        for (int d = 0; d < N; ++d)
                            for (int e = 0; e < N; ++e) {
                                Out[TIX(a, b, 0, f)] += VAL(a, b, c, d, e, f);
                                Out[TIX(a, d, 1, f)] += VAL(a, b, c, d, e, f);
                                Out[TIX(b, c, 2, f)] += VAL(a, b, c, d, e, f);
                            }
        return NULL;
    }
    if (j->k0 == 3 && N <= 64) { /* job 1 (:116-165): cases 4-5 in one N^5 nest, case 6 (c == d) in an N^4 nest of its own */
        for (int f = 0; f < C; ++f)
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b)
                    for (int c = 0; c < N; ++c)
                        for (int d = 0; d < N; ++d)
                            for (int e = 0; e < N; ++e) {
                                Out[TIX(b, d, 3, f)] += VAL(a, b, c, d, e, f);
                                Out[TIX(d, e, 4, f)] += VAL(a, b, c, d, e, f);
                            }
        for (int f = 0; f < C; ++f)
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b)
                    for (int c = 0; c < N; ++c)
                        for (int e = 0; e < N; ++e) Out[TIX(a, b, 5, f)] += VAL(a, b, c, c, e, f);
        return NULL;
    }
    /* jobs 2-5 (:168-408): N^4 nests and smaller, off the critical path -- evaluated through the case table */
    gfo_case cases[18];
    family_cases(18, cases);
    for (int k = j->k0; k < j->k1; ++k)
        run_case(&cases[k], k, 18, j->N, j->C, j->A, j->P, j->Out, NULL, NULL);
    return NULL;
}

void gfo_r18_thread_forward(const double *P, const double *A, double *Out, int N, int C) {
    pthread_t th[6];
    r18_job jobs[6];
    memset(Out, 0, sizeof(double) * (size_t)N * N * 18 * C);
    for (int t = 0; t < 6; ++t) {
        jobs[t].P = P;
        jobs[t].A = A; /* ungated on purpose */
        jobs[t].Out = Out;
        jobs[t].N = N;
        jobs[t].C = C;
        jobs[t].k0 = 3 * t;
        jobs[t].k1 = 3 * t + 3;
        pthread_create(&th[t], NULL, r18_thread_job, &jobs[t]);
    }
    for (int t = 0; t < 6; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------------
 * Feature mixers.
 * ------------------------------------------------------------------------------------------ */

/* MatMul (MatMul.h:48-67 forward, :69-82 backward): C[M][N] = A[M][K] * B[K][N]. */
void gfo_matmul_forward(const double *A, const double *B, double *Cm, int M, int K, int N) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += A[(size_t)i * K + k] * B[(size_t)k * N + j];
            Cm[(size_t)i * N + j] = s;
        }
}

void gfo_matmul_backward(const double *dC, const double *A, const double *B, double *dA, double *dB,
                         int M, int K, int N) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            const double g = dC[(size_t)i * N + j];
            for (int k = 0; k < K; ++k) {
                dA[(size_t)i * K + k] += g * B[(size_t)k * N + j];
                dB[(size_t)k * N + j] += g * A[(size_t)i * K + k];
            }
        }
}

/* MatTensorMul (MatTensorMul.h:47-68, :70-85): Out[R][J][D] = sum_k X[R][Kd] * F[Kd][J][D]. */
void gfo_mattensormul_forward(const double *X, const double *F, double *Out, int R, int Kd, int J, int D) {
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < J; ++j)
            for (int d = 0; d < D; ++d) {
                double s = 0.0;
                for (int k = 0; k < Kd; ++k) s += X[(size_t)i * Kd + k] * F[((size_t)k * J + j) * D + d];
                Out[((size_t)i * J + j) * D + d] = s;
            }
}

void gfo_mattensormul_backward(const double *G, const double *X, const double *F, double *dX, double *dF,
                               int R, int Kd, int J, int D) {
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < J; ++j)
            for (int d = 0; d < D; ++d) {
                const double g = G[((size_t)i * J + j) * D + d];
                for (int k = 0; k < Kd; ++k) {
                    if (dX) dX[(size_t)i * Kd + k] += g * F[((size_t)k * J + j) * D + d];
                    dF[((size_t)k * J + j) * D + d] += g * X[(size_t)i * Kd + k];
                }
            }
}

/* TensorMatMul (TensorMatMul.h:46-67, :69-84): Out[R][J][D] = sum_k F[R][Kd][D] * Y[Kd][J]. */
void gfo_tensormatmul_forward(const double *F, const double *Y, double *Out, int R, int Kd, int J, int D) {
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < J; ++j)
            for (int d = 0; d < D; ++d) {
                double s = 0.0;
                for (int k = 0; k < Kd; ++k) s += F[((size_t)i * Kd + k) * D + d] * Y[(size_t)k * J + j];
                Out[((size_t)i * J + j) * D + d] = s;
            }
}

void gfo_tensormatmul_backward(const double *G, const double *F, const double *Y, double *dF, double *dY,
                               int R, int Kd, int J, int D) {
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < J; ++j)
            for (int d = 0; d < D; ++d) {
                const double g = G[((size_t)i * J + j) * D + d];
                for (int k = 0; k < Kd; ++k) {
                    dF[((size_t)i * Kd + k) * D + d] += g * Y[(size_t)k * J + j];
                    if (dY) dY[(size_t)k * J + j] += g * F[((size_t)i * Kd + k) * D + d];
                }
            }
}

/* RisiContraction_18_dropout (RisiContraction_18_dropout.h:106-477, :479-783): slices with use[k] == 0 are skipped in
 * forward (stay 0) and in backward; in test mode (train == 0) every slice is used and the value is scaled by nKept/18
 * (:465-471).  The mask itself is drawn by the caller (the reference draws it with rand(), :113-125). */
int gfo_r18_dropout_forward(const int *use, int train, int nKept, const double *P, const double *A, double *Out, int N, int C) {
    if (gfo_contract_forward(18, P, A, Out, N, C) != 0) return -1;
    for (size_t r = 0; r < (size_t)N * N; ++r)
        for (int k = 0; k < 18; ++k)
            for (int f = 0; f < C; ++f) {
                double *o = &Out[(r * 18 + k) * C + f];
                if (train) {
                    if (!use[k]) *o = 0.0;
                } else {
                    *o *= (double)nKept / 18.0;
                }
            }
    return 0;
}

int gfo_r18_dropout_backward(const int *use, const double *G, const double *A, double *dP, int N, int C) {
    const size_t n = (size_t)N * N * 18 * C;
    double *Gm = (double *)malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; ++i) Gm[i] = use[(i / C) % 18] ? G[i] : 0.0;
    const int rc = gfo_contract_backward(18, Gm, A, dP, N, C);
    free(Gm);
    return rc;
}

/* Mask draw of RisiContraction_18_dropout::forward (:113-125) given the caller's rand() stream. */
void gfo_r18_dropout_draw(int nKept, int *use, int (*next_rand)(void)) {
    for (int i = 0; i < 18; ++i) use[i] = 0;
    for (int i = 0; i < nKept; ++i)
        for (;;) {
            const int j = next_rand() % 18;
            if (!use[j]) {
                use[j] = 1;
                break;
            }
        }
}

/* CustomMatMulTensor (CustomMatMulTensor.h:47-68, :70-85): channel mix Out[i][j][k] = sum_v W[k][v] * T[i][j][v]
 * over rows = nRows*nColumns positions; W is [Kout][V]. */
void gfo_custommatmultensor_forward(const double *W, const double *T, double *Out, int rows, int V, int Kout) {
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < Kout; ++k) {
            double s = 0.0;
            for (int v = 0; v < V; ++v) s += W[(size_t)k * V + v] * T[(size_t)r * V + v];
            Out[(size_t)r * Kout + k] = s;
        }
}

void gfo_custommatmultensor_backward(const double *G, const double *W, const double *T, double *dW, double *dT, int rows,
                                     int V, int Kout) {
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < Kout; ++k) {
            const double g = G[(size_t)r * Kout + k];
            for (int v = 0; v < V; ++v) {
                dW[(size_t)k * V + v] += g * T[(size_t)r * V + v];
                dT[(size_t)r * V + v] += g * W[(size_t)k * V + v];
            }
        }
}

/* StackTensor3D (StackTensor3D.h:54-73, :75-90): copy nRows tensors [nCols][n1][n2] into one
 * contiguous [nRows][nCols][n1][n2]; backward scatter-adds the stacked gradient back. */
void gfo_stack_forward(const double *const *tensors, double *Out, int nRows, size_t per) {
    for (int r = 0; r < nRows; ++r) memcpy(Out + (size_t)r * per, tensors[r], sizeof(double) * per);
}

void gfo_stack_backward(const double *G, double *const *dT, int nRows, size_t per) {
    for (int r = 0; r < nRows; ++r)
        for (size_t i = 0; i < per; ++i) dT[r][i] += G[(size_t)r * per + i];
}
