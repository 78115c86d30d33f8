// smp_prep.cpp -- see smp_prep.h.  Pure host C++; every routine cites the reference lines whose RESULT it must
// reproduce exactly (tie-breaking included: the reference's exchange sorts are not stable, so they are restated
// swap for swap rather than replaced by std::sort).
#include "smp_prep.h"

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>

namespace gfsmp {

static void *default_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
static void default_free(void *p) { std::free(p); }
void *(*table_alloc)(size_t) = default_alloc;
void (*table_free)(void *) = default_free;

namespace {

constexpr int kMaxVertices = 4096;  // per molecule, for the stack-resident vertex tables of the batch builder

// fn(i) for i in [0, n) on up to GF_PREP_THREADS (default: hardware concurrency, at most 32) host threads; every
// iteration writes disjoint memory, so the result does not depend on the thread count or on the schedule.  Iterations
// are handed out in small blocks from a shared counter: molecules differ a lot in cost (O(V^3) each).
// Worker threads are kept between calls (a training loop prepares a batch per step: spawning 31 threads for each of the
// seven parallel sections cost ~3 ms of a 24 ms preparation).  One job at a time; the caller takes part in it.
class WorkerPool {
public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (size_t t = 0; t < threads_.size(); ++t) threads_[t].join();
    }
    // runs job() on `helpers` pool threads and on the calling thread; returns when all of them have finished
    void run(int helpers, const std::function<void()> &job) {
        std::lock_guard<std::mutex> one_job(run_);
        {
            std::lock_guard<std::mutex> lk(m_);
            while ((int)threads_.size() < helpers) {
                const int id = (int)threads_.size();
                threads_.emplace_back([this, id]() { loop(id); });
            }
            job_ = &job;
            want_ = helpers;
            pending_ = helpers;
            ++generation_;
        }
        wake_.notify_all();
        job();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int id) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void()> *job;
            {
                std::unique_lock<std::mutex> lk(m_);
                wake_.wait(lk, [&]() { return stop_ || (generation_ != seen && id < want_); });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            (*job)();
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex run_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    const std::function<void()> *job_ = nullptr;
    unsigned generation_ = 0;
    int want_ = 0, pending_ = 0;
    bool stop_ = false;
};
// One pool per CALLING thread: a loop that prepares two batches at once on two host threads (the graph preparation of a batch
// takes longer than its device step since round 2) gets two sets of workers instead of queueing on one.
WorkerPool &pool() {
    static thread_local WorkerPool p;
    return p;
}

template <typename Fn>
void parallel_for(int n, const Fn &fn) {
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *e = std::getenv("GF_PREP_THREADS")) nt = std::atoi(e);
    nt = std::max(1, std::min(nt, std::getenv("GF_PREP_THREADS") ? 128 : 32));
    if (n < 64 || nt == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    const int block = std::max(1, std::min(16, n / (nt * 8)));
    std::atomic<int> next(0);
    const std::function<void()> worker = [&]() {
        for (;;) {
            const int lo = next.fetch_add(block);
            if (lo >= n) return;
            const int hi = std::min(n, lo + block);
            for (int i = lo; i < hi; ++i) fn(i);
        }
    };
    pool().run(nt - 1, worker);
}

// a handful of independent tasks (one per level), each on its own thread
template <typename Fn>
void parallel_tasks(int n, const Fn &fn) {
    if (std::getenv("GF_PREP_THREADS") && std::atoi(std::getenv("GF_PREP_THREADS")) <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next(0);
    const std::function<void()> worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            fn(i);
        }
    };
    pool().run(n - 1, worker);
}

const int kInf = 1000000000;  // SMP_omega.h:1064

// SMP_omega.h:358-380.  The initialisation visits (i,j) in row-major order and, on an edge, writes BOTH [i][j] and
// [j][i]; a later visit of (j,i) overwrites [j][i] again.  Restated literally so asymmetric inputs behave the same.
void hop_distances(int V, const int *adj, std::vector<int> *out) {
    std::vector<int> &sp = *out;
    sp.assign((size_t)V * V, 0);
    for (int i = 0; i < V; ++i)
        for (int j = 0; j < V; ++j) {
            sp[i * V + j] = (i == j) ? 0 : kInf;
            if (i != j && adj[i * V + j] > 0) {
                sp[i * V + j] = 1;
                sp[j * V + i] = 1;
            }
        }
    for (int k = 0; k < V; ++k)
        for (int i = 0; i < V; ++i)
            for (int j = 0; j < V; ++j) sp[i * V + j] = std::min(sp[i * V + j], sp[i * V + k] + sp[k * V + j]);
}

// SMP_omega.h:382-404: histogram[v][d*F + f] = sum of the raw feature f over vertices at hop distance exactly d.
void wl_features(const Config &cfg, int V, const double *feature, const std::vector<int> &sp, std::vector<double> *out) {
    const int F = cfg.nFeatures, FD = cfg.fdim();
    out->assign((size_t)V * FD, 0.0);
    for (int v = 0; v < V; ++v)
        for (int d = 0; d <= cfg.nDepth; ++d)
            for (int u = 0; u < V; ++u)
                if (sp[u * V + v] == d)
                    for (int f = 0; f < F; ++f) (*out)[(size_t)v * FD + d * F + f] += feature[(size_t)u * F + f];
}

// SMP_omega.h:406-434: lexicographic compare of the histograms, exchange sort into DESCENDING order, rank = position.
int compare_wl(const std::vector<double> &wl, int FD, int u, int v) {
    for (int f = 0; f < FD; ++f) {
        if (wl[(size_t)u * FD + f] < wl[(size_t)v * FD + f]) return -1;
        if (wl[(size_t)u * FD + f] > wl[(size_t)v * FD + f]) return 1;
    }
    return 0;
}

void rank_vertices(int V, int FD, const std::vector<double> &wl, std::vector<int> *rank) {
    std::vector<int> order(V);
    for (int v = 0; v < V; ++v) order[v] = v;
    for (int i = 0; i < V; ++i)
        for (int j = i + 1; j < V; ++j)
            if (compare_wl(wl, FD, order[i], order[j]) < 0) std::swap(order[i], order[j]);
    rank->assign(V, 0);
    for (int i = 0; i < V; ++i) (*rank)[order[i]] = i;
}

// SMP_omega.h:476-507: order by (hop distance from v, rank) with the reference's exchange sort, then drop whole
// farthest hop shells until the field fits.
// use_rank = false: the `_physics` variant of limit_receptive_field (SMP_omega_physics.h:436-450) swaps on distance only.
void cap_field(int v, int V, int cap, const std::vector<int> &sp, const std::vector<int> &rank, std::vector<int> *field, bool use_rank) {
    std::vector<int> &A = *field;
    for (size_t i = 0; i < A.size(); ++i)
        for (size_t j = i + 1; j < A.size(); ++j) {
            const int di = sp[v * V + A[i]], dj = sp[v * V + A[j]];
            if (di > dj) {
                std::swap(A[i], A[j]);
            } else if (use_rank && di == dj && rank[A[i]] > rank[A[j]]) {
                std::swap(A[i], A[j]);
            }
        }
    while ((int)A.size() > cap) {
        const int d = sp[v * V + A.back()];
        while (!A.empty() && sp[v * V + A.back()] == d) A.pop_back();
    }
}

}  // namespace

void prepare_molecule(const Config &cfg, int V, const int *adj, const double *feature, Molecule *out) {
    out->V = V;
    hop_distances(V, adj, &out->hops);
    wl_features(cfg, V, feature, out->hops, &out->wl);
    rank_vertices(V, cfg.fdim(), out->wl, &out->rank);
    // receptive fields, SMP_omega.h:509-537
    out->phi.resize(cfg.nLevels + 1);
    for (int l = 0; l <= cfg.nLevels; ++l) {  // reused storage: clear, keep capacity
        out->phi[l].resize(V);
        for (int v = 0; v < V; ++v) out->phi[l][v].clear();
    }
    for (int v = 0; v < V; ++v) out->phi[0][v].assign(1, v);
    for (int l = 1; l <= cfg.nLevels; ++l)
        for (int v = 0; v < V; ++v) {
            std::vector<int> &field = out->phi[l][v];
            for (int u = 0; u < V; ++u) {
                if (out->hops[u * V + v] > 1) continue;
                const std::vector<int> &B = out->phi[l - 1][u];  // union_set (:436-449): append unseen, keep order
                for (size_t i = 0; i < B.size(); ++i)
                    if (std::find(field.begin(), field.end(), B[i]) == field.end()) field.push_back(B[i]);
            }
            if ((int)field.size() > cfg.max_receptive_field)
                cap_field(v, V, cfg.max_receptive_field, out->hops, out->rank, &field, !cfg.physics);
            if (cfg.has_WL_ordering && !cfg.physics)  // sort() at :451-459: exchange sort by ascending rank
                for (size_t i = 0; i < field.size(); ++i)
                    for (size_t j = i + 1; j < field.size(); ++j)
                        if (out->rank[field[i]] > out->rank[field[j]]) std::swap(field[i], field[j]);
        }
}

void build_batch(const Config &cfg, int nMol, const int *nVertices, const int *adj, const double *feature,
                 const double *coulomb, BatchLayout *out) {
    const int L = cfg.nLevels, FD = cfg.fdim(), F = cfg.nFeatures;
    const std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    out->nMol = nMol;
    out->mols.resize(nMol);  // (the molecules of the previous batch keep their vectors: no allocator traffic in steady state)
    out->mol_first_vertex.assign(nMol + 1, 0);
    std::vector<size_t> adj_off(nMol + 1, 0);
    for (int m = 0; m < nMol; ++m) {
        out->mol_first_vertex[m + 1] = out->mol_first_vertex[m] + nVertices[m];
        adj_off[m + 1] = adj_off[m] + (size_t)nVertices[m] * nVertices[m];
    }
    const int totalV = out->mol_first_vertex[nMol];
    out->x.assign((size_t)totalV * FD, 0.f);
    if (out->device_tables) {
        out->mol_nv.assign(nVertices, nVertices + nMol);
        out->mol_adj_off.resize((size_t)nMol + 1);
        out->max_vertices = 1;
        for (int m = 0; m <= nMol; ++m) out->mol_adj_off[(size_t)m] = (int64_t)adj_off[m];
        for (int m = 0; m < nMol; ++m) out->max_vertices = std::max(out->max_vertices, nVertices[m]);
        out->mol_adj.assign(adj, adj + adj_off[nMol]);
        if (coulomb)
            out->mol_coul.assign(coulomb, coulomb + adj_off[nMol]);
        else
            out->mol_coul.clear();
    }
    parallel_for(nMol, [&](int m) {
        const int V = nVertices[m], v0 = out->mol_first_vertex[m];
        prepare_molecule(cfg, V, adj + adj_off[m], feature + (size_t)v0 * F, &out->mols[m]);
        for (size_t i = 0; i < (size_t)V * FD; ++i) out->x[(size_t)v0 * FD + i] = (float)out->mols[m].wl[i];
    });

    const bool timing = std::getenv("GF_PREP_TIMING") != nullptr;
    const std::chrono::steady_clock::time_point t_mol = std::chrono::steady_clock::now();
    // keep the vectors of a previous batch (their capacity): re-faulting ~50 MB of fresh pages costs more than the work
    out->level.resize(L + 1);
    for (int l = 0; l <= L; ++l) out->level[l].buckets.clear();
    // node numbering per level: level 0 in (molecule, vertex) order; level >= 1 bucketed by field size (stable)
    std::vector<std::vector<int> > node_of(L + 1, std::vector<int>(totalV, -1));  // [level][global vertex] -> node
    parallel_tasks(L + 1, [&](int l) {  // the levels are independent here
        LevelLayout &lv = out->level[l];
        std::vector<std::pair<int, int> > order;  // (size, global vertex)
        order.reserve(totalV);
        for (int m = 0; m < nMol; ++m)
            for (int v = 0; v < nVertices[m]; ++v)
                order.push_back(std::make_pair((int)out->mols[m].phi[l][v].size(), out->mol_first_vertex[m] + v));
        if (l > 0) std::stable_sort(order.begin(), order.end(),
                                    [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
        lv.nNodes = totalV;
        lv.node_s.resize(totalV);
        lv.node_mol.resize(totalV);
        lv.node_vertex.resize(totalV);
        lv.node_row.resize(totalV);
        lv.node_p.resize(totalV);
        lv.node_pair.resize(totalV);
        int64_t row = 0, pp = 0, pair = 0;
        for (int n = 0; n < totalV; ++n) {
            const int s = order[n].first, gv = order[n].second;
            const int m = (int)(std::upper_bound(out->mol_first_vertex.begin(), out->mol_first_vertex.end(), gv) -
                                out->mol_first_vertex.begin()) - 1;
            node_of[l][gv] = n;
            lv.node_s[n] = s;
            lv.node_mol[n] = m;
            lv.node_vertex[n] = gv - out->mol_first_vertex[m];
            lv.node_row[n] = row;
            lv.node_p[n] = pp;
            lv.node_pair[n] = pair;
            if (lv.buckets.empty() || lv.buckets.back().s != s) {
                Bucket b = {s, 0, n, row, pp};
                lv.buckets.push_back(b);
            }
            lv.buckets.back().count += 1;
            row += (int64_t)s * s;
            pp += (int64_t)s * s * s;
            pair += s;
        }
        lv.rows = row;
        lv.ppos = pp;
        lv.pairs = pair;
        if (l == 0) {  // (a level-0 field is the vertex itself: the sources of level 1)
            lv.field.resize((size_t)totalV);
            for (int n = 0; n < totalV; ++n) lv.field[(size_t)n] = lv.node_vertex[n];
        }
        lv.node_center.assign(totalV, 0);
        {  // counting sort of the nodes by (register class of s, molecule); stable, so size-major order survives inside a key
            // (a class of their own for the nodes above 32 positions: tables-forward launches per register class over ranges of this
            //  order, and those nodes run another kernel)
            auto cls = [](int s) { return s <= 1 ? 0 : s <= 4 ? 1 : s <= 8 ? 2 : s <= 16 ? 3 : s <= 32 ? 4 : 5; };
            std::vector<int> start((size_t)6 * nMol + 1, 0);
            for (int n = 0; n < totalV; ++n) start[(size_t)cls(lv.node_s[n]) * nMol + lv.node_mol[n] + 1] += 1;
            for (size_t k = 0; k + 1 < start.size(); ++k) start[k + 1] += start[k];
            lv.mol_order.assign(totalV, 0);
            for (int n = 0; n < totalV; ++n) lv.mol_order[(size_t)start[(size_t)cls(lv.node_s[n]) * nMol + lv.node_mol[n]]++] = n;
        }
        {  // work items of the one-launch gather (smp_prep.h), molecule-major: (source, 64-lane chunk of its (p, channel quad) rows,
           // eight positions q) -- a source of more than eight positions is several items (round 4: eight accumulators per item)
            const int nl = std::max(1, cfg.nChanels / 4);
            std::vector<int> start((size_t)nMol + 1, 0);
            for (int n = 0; n < totalV; ++n) start[(size_t)lv.node_mol[n] + 1] += 1;
            for (size_t k = 0; k + 1 < start.size(); ++k) start[k + 1] += start[k];
            std::vector<int> order((size_t)totalV);
            for (int n = 0; n < totalV; ++n) order[(size_t)start[(size_t)lv.node_mol[n]]++] = n;
            lv.gather_items.clear();
            for (size_t i = 0; i < order.size(); ++i) {
                const int n = order[i], s = lv.node_s[n], chunks = (s * nl + 63) / 64;
                const int qchunks = s <= 8 ? 1 : (gather_pad(s) + 7) / 8;
                for (int h = 0; h < qchunks; ++h)
                    for (int c = 0; c < chunks; ++c) {
                        lv.gather_items.push_back(n);
                        lv.gather_items.push_back(c | (h << 16));
                    }
            }
        }
        for (int n = 0; n < totalV; ++n) {
            const std::vector<int> &fld = out->mols[lv.node_mol[n]].phi[l][lv.node_vertex[n]];
            const int v = lv.node_vertex[n];
            int c = -1;
            for (size_t i = 0; i < fld.size(); ++i)
                if (fld[i] == v) c = (int)i;
            lv.node_center[n] = c;  // always found: the centre survives the cap (SMP_omega.h:476-507)
        }
    });
    out->top_node_of_vertex = node_of[L];
    out->node_of_vertex = node_of;
    const std::chrono::steady_clock::time_point t_order = std::chrono::steady_clock::now();

    // The levels' tables are independent of each other (they only read node_of and the previous level's node bookkeeping), so
    // the serial parts of all levels run side by side (one task per level) and the node loops of all levels share one
    // parallel_for: four phases instead of four per level.
    std::vector<std::vector<int> > src_node_of_pair(L + 1);
    parallel_tasks(L, [&](int li) {   // phase A: sizes, quad tables and their launch order
        const int l = li + 1;
        LevelLayout &lv = out->level[l];
        const LevelLayout &prev = out->level[l - 1];
        std::vector<int> &pair_src_node = src_node_of_pair[l];
        (void)prev;
        (void)pair_src_node;
        // every element of these is written by the node loop below: resize only (no fill pass over ~50 MB per batch)
        const bool dev = out->device_tables;
        lv.adj.resize(dev ? 0 : (size_t)lv.rows);
        lv.rsum.resize(dev ? 0 : (size_t)lv.pairs);
        lv.rowscale.resize(dev ? 0 : (size_t)lv.nNodes * 2);
        lv.field.resize((size_t)lv.pairs);
        lv.quad_node.clear();
        lv.quad_b0.clear();
        lv.pair_node.resize((size_t)lv.pairs);
        lv.pair_src_row.resize((size_t)lv.pairs);
        lv.pair_src_pair.resize((size_t)lv.pairs);
        lv.pair_src_s.resize((size_t)lv.pairs);
        lv.pi.resize(dev ? 0 : (size_t)lv.rows);
        pair_src_node.assign((size_t)lv.pairs, 0);
        {
            lv.node_panel.resize((size_t)lv.nNodes);
            int np = 0;
            for (int n = 0; n < lv.nNodes; ++n) {
                lv.node_panel[(size_t)n] = np;
                const int sz = lv.node_s[n], gpp = sz >= 32 ? 1 : std::min(8, 32 / (sz < 1 ? 1 : sz));
                // (a node of more than 32 positions has no row panels: its row groups do not fit the 32-row tiles of the panel kernels --
                //  the workgroup kernels take it, smp_fused.hip: big_part; nodes are numbered by size, so those nodes come last)
                if (sz <= 32) np += (sz + gpp - 1) / gpp;
            }
            lv.npanels = np;
        }
        for (int n = 0; n < lv.nNodes; ++n)
            for (int b0 = 0; b0 < lv.node_s[n]; b0 += 4) {
                lv.quad_node.push_back(n);
                lv.quad_b0.push_back(b0);
            }
        {  // launch order of tables-forward: the consumers of one molecule's source tensors run together (stable counting sort)
            auto cls = [](int s) { return s <= 4 ? 0 : s <= 8 ? 1 : s <= 16 ? 2 : 3; };
            const size_t nq = lv.quad_node.size();
            std::vector<int> start((size_t)4 * nMol + 1, 0);
            for (size_t q = 0; q < nq; ++q) {
                const int n = lv.quad_node[q];
                start[(size_t)cls(lv.node_s[n]) * nMol + lv.node_mol[n] + 1] += 1;
            }
            for (size_t k = 0; k + 1 < start.size(); ++k) start[k + 1] += start[k];
            lv.quad_order.assign(nq, 0);
            for (size_t q = 0; q < nq; ++q) {
                const int n = lv.quad_node[q];
                lv.quad_order[(size_t)start[(size_t)cls(lv.node_s[n]) * nMol + lv.node_mol[n]]++] = (int)q;
            }
        }
    });
    const std::chrono::steady_clock::time_point t_A = std::chrono::steady_clock::now();
    parallel_for(L * totalV, [&](int k) {   // phase B: per node -- reduced adjacency, row sums, selection maps
        const int l = 1 + k / totalV, n = k % totalV;
        LevelLayout &lv = out->level[l];
        const LevelLayout &prev = out->level[l - 1];
        std::vector<int> &pair_src_node = src_node_of_pair[l];
        (void)prev;
        (void)pair_src_node;
            const int m = lv.node_mol[n], v = lv.node_vertex[n], s = lv.node_s[n];
            const int V = nVertices[m], v0 = out->mol_first_vertex[m];
            const int *madj = adj + adj_off[m];
            const std::vector<int> &field = out->mols[m].phi[l][v];
            for (int i = 0; i < s; ++i) lv.field[(size_t)lv.node_pair[n] + i] = field[i];
            const bool dev = out->device_tables;
            // reduced adjacency (:556-581): 1 on the diagonal and adj[v1][v2] elsewhere, or the Coulomb entries
            const double *mcoul = coulomb ? coulomb + adj_off[m] : nullptr;
            if (!dev) {
            for (int i = 0; i < s; ++i)
                for (int j = 0; j < s; ++j)
                    lv.adj[(size_t)lv.node_row[n] + (size_t)i * s + j] =
                        mcoul ? (float)mcoul[field[i] * V + field[j]]
                              : ((field[i] == field[j]) ? 1.f : (float)madj[field[i] * V + field[j]]);
            for (int i = 0; i < s; ++i) {  // gated row sums (RisiContraction_18.h:90: entries with A <= 0 are skipped)
                float rs = 0.f;
                for (int j = 0; j < s; ++j) {
                    const float av = lv.adj[(size_t)lv.node_row[n] + (size_t)i * s + j];
                    if (av > 0.f) rs += av;
                }
                lv.rsum[(size_t)lv.node_pair[n] + i] = rs;
            }
            {  // tot = sum of the gated adjacency, tr = its trace: the factors of contraction cases 1/3 and 7 (Appendix A.2)
                float tot = 0.f, tr = 0.f;
                for (int i = 0; i < s; ++i) {
                    tot += lv.rsum[(size_t)lv.node_pair[n] + i];
                    const float av = lv.adj[(size_t)lv.node_row[n] + (size_t)i * s + i];
                    if (av > 0.f) tr += av;
                }
                lv.rowscale[2 * (size_t)n] = tot;      // (expanded to one pair per row on the device: gf_smp_prepare)
                lv.rowscale[2 * (size_t)n + 1] = tr;
            }
            }
            int16_t pos[kMaxVertices];  // position of a vertex inside phi_{l-1}(w), -1 outside; reset after each neighbour
            if (V > kMaxVertices) std::abort();  // (gf_smp_prepare rejects such molecules before it gets here)
            for (int i = 0; i < V; ++i) pos[i] = -1;
            for (int a = 0; a < s; ++a) {
                const int w = field[a];
                const int64_t e = lv.node_pair[n] + a;
                const int src = node_of[l - 1][v0 + w];
                const std::vector<int> &wf = out->mols[m].phi[l - 1][w];
                lv.pair_node[(size_t)e] = n;
                pair_src_node[(size_t)e] = src;
                lv.pair_src_row[(size_t)e] = prev.node_row[src];
                lv.pair_src_pair[(size_t)e] = prev.node_pair[src];
                lv.pair_src_s[(size_t)e] = (int)wf.size();
                if (dev) continue;
                // selection map: X[i][k] = [phi_l(v)[i] == phi_{l-1}(w)[k]]   (:461-474)
                for (size_t k = 0; k < wf.size(); ++k) pos[wf[k]] = (int16_t)k;
                for (int p = 0; p < s; ++p) lv.pi[(size_t)lv.node_row[n] + (size_t)a * s + p] = pos[field[p]];
                for (size_t k = 0; k < wf.size(); ++k) pos[wf[k]] = -1;
            }
            });
    const std::chrono::steady_clock::time_point t_B = std::chrono::steady_clock::now();
    parallel_tasks(L, [&](int li) {   // phase C: consumer lists of the backward gather (prefix sums: serial per level)
        const int l = li + 1;
        LevelLayout &lv = out->level[l];
        const LevelLayout &prev = out->level[l - 1];
        std::vector<int> &pair_src_node = src_node_of_pair[l];
        (void)prev;
        (void)pair_src_node;
        // inverse index for the backward gather: consumers of a source node in increasing pair order (= fixed
        // summation order on the device), offsets by prefix sums, then a parallel fill
        lv.cons_ptr.assign((size_t)prev.nNodes + 1, 0);
        for (int64_t e = 0; e < lv.pairs; ++e) lv.cons_ptr[(size_t)pair_src_node[(size_t)e] + 1] += 1;
        for (int w = 0; w < prev.nNodes; ++w) lv.cons_ptr[(size_t)w + 1] += lv.cons_ptr[(size_t)w];
        std::vector<int64_t> cursor(lv.cons_ptr.begin(), lv.cons_ptr.end() - 1);
        tvec<int64_t> &cons_pair = lv.cons_pair;
        cons_pair.assign((size_t)lv.pairs, 0);
        for (int64_t e = 0; e < lv.pairs; ++e) cons_pair[(size_t)cursor[(size_t)pair_src_node[(size_t)e]]++] = e;
        if (out->device_tables) {   // (the device derives the per-consumer entries from cons_pair: smp.hip, build_consumer_entries)
            lv.cons_slab.clear();
            lv.cons_s.clear();
            lv.cons_row.clear();
            lv.cons_a.clear();
        } else {
            lv.cons_slab.assign((size_t)lv.pairs, 0);
            lv.cons_s.assign((size_t)lv.pairs, 0);
            lv.cons_row.assign((size_t)lv.pairs, 0);
            lv.cons_a.assign((size_t)lv.pairs, 0);
        }
        lv.cons_inv_off.assign((size_t)lv.pairs, 0);
        int64_t inv_total = 0;
        for (int w = 0; w < prev.nNodes; ++w)
            for (int64_t c = lv.cons_ptr[(size_t)w]; c < lv.cons_ptr[(size_t)w + 1]; ++c) {
                lv.cons_inv_off[(size_t)c] = inv_total;
                inv_total += prev.node_s[w];
            }
        lv.inv_count = inv_total;
        if (out->device_tables)
            lv.inv.clear();
        else
            lv.inv.assign((size_t)inv_total, (int16_t)-1);
        lv.cons_qbase.assign((size_t)prev.nNodes, 0);
        int64_t q = 0;
        for (int w = 0; w < prev.nNodes; ++w) {
            lv.cons_qbase[(size_t)w] = q;
            q += (lv.cons_ptr[(size_t)w + 1] - lv.cons_ptr[(size_t)w]) * gather_pad(prev.node_s[w]);
        }
        lv.qrec_total = q;
    });
    const std::chrono::steady_clock::time_point t_C = std::chrono::steady_clock::now();
    if (!out->device_tables)
    parallel_for(L * totalV, [&](int k) {   // phase D: per source node -- its consumers' entries and inverse maps (host-built tables only)
        const int l = 1 + k / totalV, w = k % totalV;
        LevelLayout &lv = out->level[l];
        const LevelLayout &prev = out->level[l - 1];
        std::vector<int> &pair_src_node = src_node_of_pair[l];
        (void)prev;
        (void)pair_src_node;
            for (int64_t c = lv.cons_ptr[(size_t)w]; c < lv.cons_ptr[(size_t)w + 1]; ++c) {
                const int64_t e = lv.cons_pair[(size_t)c];
                const int n = lv.pair_node[(size_t)e];
                const int s = lv.node_s[n], a = (int)(e - lv.node_pair[n]);
                lv.cons_slab[(size_t)c] = lv.node_p[n] + (int64_t)a * s * s;
                lv.cons_s[(size_t)c] = s;
                lv.cons_row[(size_t)c] = lv.node_row[n];
                lv.cons_a[(size_t)c] = a;
                if (out->device_tables) continue;
                int16_t *iv = &lv.inv[(size_t)lv.cons_inv_off[(size_t)c]];
                for (int p = 0; p < s; ++p) {
                    const int16_t k = lv.pi[(size_t)lv.node_row[n] + (size_t)a * s + p];
                    if (k >= 0) iv[k] = (int16_t)p;
                }
            }
            });
    if (timing) {
        const std::chrono::steady_clock::time_point t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        std::fprintf(stderr, "build_batch: molecules %.1f ms, node order %.1f ms, level tables %.1f ms (sizes / quads %.1f, per node %.1f, "
                             "consumer lists %.1f, inverse maps %.1f)\n",
                     ms(t_begin, t_mol), ms(t_mol, t_order), ms(t_order, t_end), ms(t_order, t_A), ms(t_A, t_B), ms(t_B, t_C), ms(t_C, t_end));
    }
}

}  // namespace gfsmp
