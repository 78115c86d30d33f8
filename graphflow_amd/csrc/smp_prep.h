// smp_prep.h -- host-side graph preparation of the batched SMP_omega driver.
//
// Restates, per molecule, what SMP_omega::complete_computation_graph does before it builds the op DAG
// (GraphFlow/SMP_omega.h:584-605): Floyd-Warshall hop distances (:358-380), Weisfeiler-Lehman histogram features
// (:382-404), lexicographic vertex ranking (:406-434), receptive fields phi_l(v) with the omega cap (:476-537),
// selection maps (the 0/1 matrices X of :461-474, :539-554, kept as index lists) and reduced adjacencies (:556-581).
// Then it lays a BATCH of molecules out for the device: per level, nodes (= one (molecule, vertex) pair) are sorted
// into buckets of equal receptive-field size so each bucket is one uniform-N contraction launch.
#ifndef GF_SMP_PREP_H_INCLUDED
#define GF_SMP_PREP_H_INCLUDED

#include <cstddef>
#include <cstdint>
#include <vector>

namespace gfsmp {

// Host memory of the tables that are uploaded every batch comes from these hooks: plain malloc / free by default (this file
// is pure host C++), page-locked memory once the HIP side installs its pair -- an upload from pageable memory is staged through
// blit KERNELS that queue behind the running step's compute kernels, from pinned memory it is a DMA-engine copy that overlaps.
extern void *(*table_alloc)(size_t bytes);
extern void (*table_free)(void *p);
template <class T>
struct TableAlloc {
    typedef T value_type;
    TableAlloc() {}
    template <class U>
    TableAlloc(const TableAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(table_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t) { table_free(p); }
    template <class U>
    bool operator==(const TableAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const TableAlloc<U> &) const { return false; }
};
template <class T>
using tvec = std::vector<T, TableAlloc<T> >;

struct Config {
    int nLevels, nChanels, nFeatures, nDepth, max_receptive_field, has_WL_ordering;
    int nContractions = 18;  // contraction family of the levels: 18 (SMP_omega/beta, SMP_2D_ver8), 10 (ver6), 50 (ver7)
    int custom_matmul = 0;   // 1: K_l is [C][nContractions C] and applied by CustomMatMulTensor (SMP_2D_ver6-8)
    // 1: the `_physics` / `_pairgraphs` family (GraphFlow/SMP_omega_physics.h): raw vertex features (no WL histogram, no
    // WL ordering, and the cap orders by hop distance only, :436-450), channels halve from level to level (:141-151), every
    // level is read out (:572-590).  One such model body is a "tower": its output is the concatenated level features.
    int physics = 0;
    // 1 (a physics tower's DEVICE configuration, gf_smp_create): every level is computed at nChanels -- the halving channel counts
    // zero-padded to one width -- so the tower's levels run the fused level kernels
    int uniform = 0;
    bool square() const { return !physics || uniform; }   // K_l is [nContractions C][C] at every level
    int level_channels(int l) const {
        if (square()) return nChanels;
        int c = nChanels >> l;
        return c < 1 ? 1 : c;
    }
    int fdim() const { return nFeatures * (nDepth + 1); }
};

// everything the reference derives from one molecule
struct Molecule {
    int V = 0;
    std::vector<int> hops;                            // [V][V] shortest paths
    std::vector<double> wl;                           // [V][F(D+1)] WL histogram features
    std::vector<int> rank;                            // [V]
    std::vector<std::vector<std::vector<int> > > phi;  // [L+1][V] receptive fields
};

void prepare_molecule(const Config &cfg, int V, const int *adj, const double *feature, Molecule *out);

struct Bucket {
    int s;             // receptive-field size of every node in the bucket
    int count;         // nodes
    int first_node;    // index of the first node (nodes of a level are numbered in bucket order)
    int64_t first_row; // offset in (i,j) rows (sum of s^2 before the bucket)
    int64_t first_p;   // offset in (a,b,c) positions (sum of s^3 before the bucket)
};

// one level of the batch, host copy of what gets uploaded
struct LevelLayout {
    int nNodes = 0;
    int64_t rows = 0;   // sum s^2
    int64_t ppos = 0;   // sum s^3
    int64_t pairs = 0;  // sum s   (one (node, neighbour) pair per promoted tensor)
    std::vector<Bucket> buckets;
    tvec<int> node_s, node_mol, node_vertex;
    tvec<int64_t> node_row, node_p, node_pair;
    // The rows-sized tables (adj, pi, inv) and what is summed from adj (rsum, rowscale) are built on the DEVICE when
    // BatchLayout::device_tables is set (smp.hip: build_level_rows / build_level_inv, from `field`, the pair tables and the
    // molecules' adjacency matrices); the host then leaves them empty.
    tvec<int> node_panel;  // [nNodes] first row panel of the node (combine-forward on row panels, smp_level_c64_fwd.hip)
    int npanels = 0;       //   a node of size s has ceil(s / gpp) panels of gpp = max(1, min(8, 32 / s)) row groups
    tvec<int> field;  // [pairs] the receptive fields back to back: field[node_pair[n] + i] = i-th vertex (index inside its molecule)
    int64_t inv_count = 0;  // elements of inv
    tvec<float> adj;  // [rows] reduced adjacency, node-major [s][s]
    tvec<float> rsum; // [pairs] r[d] = sum_e A+[d][e] (A+ = A where A > 0), pair = node_pair[n] + d
    tvec<float> rowscale;  // [nNodes][2] (tot, tr) of the node's gated adjacency: per-row factors of the level's block GEMMs
    // wave-per-pair kernels: one workgroup (4 waves) per group of 4 consecutive indices of one node
    tvec<int> quad_node, quad_b0;  // [quads]
    tvec<int> quad_order;          // [quads] quads by (size class 4/8/16/32, molecule): launch order of tables-forward
    // forward gather (levels >= 1): per pair e = node_pair[n] + a
    tvec<int> pair_node;        // [pairs]
    tvec<int64_t> pair_src_row;  // [pairs] first row of the source node's tensor in level l-1
    tvec<int> pair_src_s;       // [pairs]
    tvec<int16_t> pi;           // [rows]  pi[node_row[n] + a*s + p] = index of phi_l(v)[p] in phi_{l-1}(w_a), or -1
    // compact diagonal path (smp_fused.hip): the two tables D_bb[x,y] = P[x,y,y] and D_ac[x,y] = P[x,y,x] are plain gathers
    // of the diagonal / the centre column of f_{l-1}[w_x], so their block products run on the sum-s rows of the level below
    tvec<int64_t> pair_src_pair; // [pairs] node_pair (level l-1) of the source node of pair e
    tvec<int> node_center;       // [nNodes] position of the node's own vertex inside its receptive field (every level)
    tvec<int64_t> cons_row;      // [pairs] node_row (level l) of the consumer's node
    tvec<int> cons_a;            // [pairs] the consumer's neighbour index a
    tvec<int64_t> cons_pair;     // [pairs] the consumer's pair id e = node_pair[n] + a (level l)
    tvec<int> mol_order;         // [nNodes] the level's nodes by (size class 1/4/8/16/32, molecule): launch order of the
                                        // backward gather, so the sources that re-read one consumer's rows run together
    // ONE launch of smp_bwd_gather_all per level: wave-sized work items (source node, 64-lane chunk of its (row p, channel quad)
    // space, half of the positions q for sources above 16), the sources above 16 first, then every other source molecule by
    // molecule.  item = {node, chunk | qhalf << 16}
    tvec<int> gather_items;      // [2 * n_items]
    // backward gather, indexed by the SOURCE node (level l-1): consumers = pairs that read it
    tvec<int64_t> cons_ptr;      // [nNodes(l-1) + 1]
    tvec<int64_t> cons_slab;     // [pairs] position offset (units of C floats) of the consumer's [s][s] slab in P
    tvec<int> cons_s;           // [pairs] consumer's s
    tvec<int64_t> cons_inv_off;  // [pairs] offset into inv
    tvec<int16_t> inv;          // per consumer: [s_w] position in the consumer's field of source position p, or -1
    // records of the backward gather (smp_fused.hip: smp_bwd_gather_v2): gather_pad(s_w) four-dword records per consumer entry,
    // the entries of a source back to back from cons_qbase[w] on
    tvec<int64_t> cons_qbase;    // [nNodes(l-1)]
    int64_t qrec_total = 0;
};

// Register classes of the backward gather: a source of size s_w runs the code path with gather_pad(s_w) accumulators (and
// gather_pad(s_w) records per consumer: no size test inside its loops).
#ifdef __HIPCC__
#define GF_PREP_HD __host__ __device__
#else
#define GF_PREP_HD
#endif
GF_PREP_HD inline int gather_pad(int s) {
    return s <= 1 ? 1 : s <= 2 ? 2 : s <= 4 ? 4 : s <= 5 ? 5 : s <= 6 ? 6 : s <= 8 ? 8 : s <= 10 ? 10 : s <= 12 ? 12 : s <= 16 ? 16 : s <= 32 ? 32 : 64;   // (64: round 6, sources of 33 .. 64 positions)
}

struct BatchLayout {
    bool device_tables = false;  // in: see LevelLayout::field
    // (device_tables) the batch's inputs in page-locked tables, for the device-side builder: vertex counts, V x V adjacency
    // (and Coulomb) matrices back to back, their offsets
    tvec<int> mol_nv, mol_adj;
    tvec<int64_t> mol_adj_off;
    tvec<double> mol_coul;
    int max_vertices = 1;
    int nMol = 0;
    std::vector<int> mol_first_vertex;  // [nMol+1] prefix sum of vertex counts (level-0 node = global vertex id)
    tvec<float> x;                      // [nVertices][F(D+1)] WL features, level-0 input
    std::vector<LevelLayout> level;     // [L+1]; level[0] has only node bookkeeping
    std::vector<int> top_node_of_vertex;  // [nVertices] node index at level L of global vertex id
    std::vector<std::vector<int> > node_of_vertex;  // [L+1][nVertices] the same for every level (per-level readout)
    std::vector<Molecule> mols;         // kept for introspection (receptive fields)
};

// coulomb: NULL, or the molecules' V x V Coulomb matrices back to back (DenseGraph::coulomb) -- the reduced adjacency of
// the use_coulomb constructors (SMP_omega.h:568-579) is then coulomb[v1][v2], diagonal included.
void build_batch(const Config &cfg, int nMol, const int *nVertices, const int *adj, const double *feature,
                 const double *coulomb, BatchLayout *out);

}  // namespace gfsmp
#endif
