// smp_internal.h -- state of one gf_smp handle, shared by smp.hip (op-by-op level pipeline) and smp_fused.hip.
#ifndef GF_SMP_INTERNAL_H_INCLUDED
#define GF_SMP_INTERNAL_H_INCLUDED

#include <vector>

#include "gf_internal.h"
#include "smp_prep.h"

namespace gf {
gf_status gemm(gf_ctx *ctx, bool ta, bool tb, int M, int N, int K, const float *A, int lda, long long sA, const float *B,
               int ldb, long long sB, float *C, int ldc, long long sC, int batch, int accumulate);
gf_status gemm_rs(gf_ctx *ctx, bool ta, bool tb, int M, int N, int K, const float *A, int lda, long long sA, const float *B,
                  int ldb, long long sB, float *C, int ldc, long long sC, int batch, int accumulate, const float *rs,
                  int rs_ld, int scol);
// one GEMM of a grouped launch (mixers.hip): op(A)[M,K] op(B)[K,N] -> C[M,N]; nseg > 0 splits K into pieces
struct GemmSpec {
    const float *A, *B;
    float *C;
    int M, N, K, lda, ldb, ldc;
    int nseg;
    long long a_off[4], b_off[4];
    int klen[4];
    const float *rs;  // optional per-row scaling of op(A) (GemmArgs::rs in mixers.hip); nullptr = none
    int rs_ld;
    int scol[4];      // column of rs per K piece (piece 0 when nseg == 0), < 0 = unscaled piece
};
// partial images of one product of a split launch: `splits` images of n floats back to back from `part` on
struct FoldGroup {
    const float *part;
    int splits;
    size_t n;
};
gf_status gemm_grouped_free(gf_ctx *ctx, bool tb, const GemmSpec *specs, int n, const char *name);
gf_status gemm_grouped_free_tn(gf_ctx *ctx, const GemmSpec *specs, int n, float *part, size_t part_floats, FoldGroup *out,
                               const char *name);
bool gemm_grouped_supported(const GemmSpec *specs, int n, bool ta, bool tb);
gf_status gemm_grouped_rows(gf_ctx *ctx, bool ta, bool tb, const GemmSpec *specs, int n, int rows, int accumulate = 0);
gf_status gemm_grouped_splitk(gf_ctx *ctx, const GemmSpec *specs, int n, int rows, float *dest, int accumulate);
// trow != nullptr: compact O / dO layout ([O_loc | U], 2C wide) with the transposed-row gather inside the product (see kernel)
// trowf (optional): the level's packed table with the presence bits of the S_ab / T6 blocks (DevLevel::trowf)
gf_status smp_rowpanel_products_c64(gf_ctx *ctx, bool forward, const float *A, const float *rowscale, const float *Wst, float *Out,
                                    int rows, const int *trow, const int *trowf = nullptr, bool skip_zero_grads = false,
                                    const void *wimg = nullptr, int C = 64,   // C = 32 (round 4): split products with prebuilt images only
                                    int nf = 2, int nx = 0);   // row factors per row of `rowscale`: 2 = (tot, tr); 8 = one per product (slice dropout, C = 32)
// the compact-layout products on the f16 matrix pipe with two-half fp32 operands (smp_level_c64_split.hip; GF_SMP_SPLIT=0: fp32 MFMA)
bool smp_split_products(const gf_ctx *ctx);
gf_status smp_rowpanel_split_c64(gf_ctx *ctx, bool forward, const float *A, const float *rowscale, const float *Wst, float *Out,
                                 int rows, const int *trow, int cus, const int *trowf = nullptr, bool skip_zero_grads = false,
                                 const void *wimg = nullptr, int C = 64, int nf = 2, int nx = 0);
// the split kernels' weight images of a level (both directions), built once per forward pass (smp_level_c64_split.hip)
size_t smp_split_image_bytes();
gf_status smp_split_build_images(gf_ctx *ctx, const float *const *Wst, void *const *img, int n, int C = 64, const float *const *X = nullptr);
// weight gradients of a fused level at C = 32 (smp_wgrad_direct<32>): partial images of 8 x 32 x 32 floats per workgroup
gf_status smp_wgrad_partials_direct_c32(gf_ctx *ctx, const float *T, const float *dO, const float *rowscale, int rows, int splits, float *part,
                                        const int *trow, const int *trowf, unsigned *words, const unsigned *chan = nullptr, float smax = 0.f,
                                        const unsigned *row_max = nullptr, int nf = 2, int C = 32, float *xpart = nullptr);
// C = 32 or (round 5) 16
bool smp_wgrad_extra_supported(int nf);
gf_status smp_wgrad_channel_maxima_ld(gf_ctx *ctx, const float *fprev, long long prev_rows, int ld0, const float *dsrc, long long drows, int ld1, int C,
                                      unsigned *words);
size_t smp_wgrad_direct_words_c32();
int smp_wgrad_direct_splits(gf_ctx *ctx, long long rows);
gf_status smp_small_split_c64(gf_ctx *ctx, bool transposed, int n, const int *prog, const float *const *In, float *const *Out, const int *rows,
                              const int *pos0, const void *wimg, const char *name, int C = 64);
// where the split-operand weight gradients take their per-column exponents from (smp_level_c64_split.hip: smp_wgrad_split): either
// `cmax`, explicit per-column bounds of the nine operand blocks (576 float bits, device), or `chan`, the level's per-channel maxima
// (128 float bits: max |f_{l-1}| then max |dz_l|) with the largest receptive field and row factors of the level
struct WgradScales {
    const unsigned *cmax = nullptr, *chan = nullptr;
    float smax = 0.f, max_tot = 0.f, max_tr = 0.f;
    const unsigned *row_max = nullptr;   // or: {max |tot|, max |tr|} as float bits in device memory
    bool any() const { return cmax || chan; }
};
gf_status smp_wgrad_partials_split_c64(gf_ctx *ctx, const float *T, const float *dO, const float *rowscale, int rows, int kchunk,
                                       int splits, float *part, const int *trow, const WgradScales &ws, const int *trowf = nullptr);
size_t smp_wgrad_bound_words();
gf_status smp_wgrad_channel_maxima(gf_ctx *ctx, const float *fprev, long long prev_rows, const float *dsrc, long long drows, unsigned *words);
size_t smp_wgrad_bound_words_exact();
gf_status smp_wgrad_column_bounds_exact(gf_ctx *ctx, const float *T, const float *dO, const float *rowscale, int rows, unsigned *words);
gf_status splitk_fold(gf_ctx *ctx, const float *part, float *dest, size_t total, int splits, int accumulate);
// the same product, partial images only (fold == caller's): `part` receives out->splits images of 8 * 64 * 64 floats
gf_status smp_wgrad_partials_c64(gf_ctx *ctx, const float *T, const float *dO, const float *rowscale, int rows, float *part,
                                 size_t part_floats, FoldGroup *out, const int *trow, const WgradScales &ws = WgradScales(),
                                 const int *trowf = nullptr);
}


namespace gf {
// min_pad: the smallest padded width the handle may compute at (32 for the towers of a slice-dropout model)
gf_status smp_create(gf_ctx *ctx, const gf_smp_config *cfg, bool pad_channels, gf_smp **out, int min_pad = 0);   // gf_smp_create = (.., true, ..)
void smp_derive_plan(gf_smp *s, bool allow_embed);
gf_status smp_switch_plan(gf_smp *s, bool embed);
constexpr int kPadMaxLevels = 15;   // levels a padded model's layout map holds (gf_smp_create: deeper models compute at nChanels)
}
struct gf_smp {
    gf_ctx *ctx = nullptr;
    gfsmp::Config cfg;    // what the device computes with: nChanels padded to 32 / 64 / a multiple of 4 (gf_smp_create, round 4)
    bool bwd_consumed = false;   // an op-by-op level's reverse sweep has overwritten its Q since the last forward
    gfsmp::Config ucfg;   // the caller's configuration: the layout of parameters, gradients, features and activations at the C ABI
    float *pad_p = nullptr, *pad_g = nullptr, *pad_feat = nullptr;   // padded copies (cfg.nChanels != ucfg.nChanels)
    // Round 5, SMP_2D_ver6 (RisiContraction_10) on the fused RisiContraction_18 level: != 0 = the caller's channel count C.  With a symmetric
    // reduced adjacency every one of the ten "1+1+1" slices is a slice of RisiContraction_18 applied to f_{l-1} or to its per-node
    // TRANSPOSE (smp.hip: v6_slot), so the device computes an 18-slice model on [f | f^T | 0] channels: channels [C, 2C) of every level's
    // activations hold the transposed matrices (dup_transposed_channels after each level, fold_transposed_channels in the reverse sweep)
    int dup_channels = 0;
    // ... and SMP_2D_ver7 (RisiContraction_50) likewise: 46 of its 50 slices are slices of RisiContraction_18 on f or f^T; the other four
    // (cases 25, 41, 42, 45: S_bc tr and the three pair marginals weighted by the adjacency's diagonal, which is 1 in a reduced adjacency)
    // are n_extra = 3 more C x C products on the level's tables -- (S_ab, 1), (S_bc, 1), (S_bc, tr) -- whose weights X_l sit BEHIND the
    // padded parameter vector ([.. W | X_1 | .. | X_L], three [Cc][Cc] blocks per level) and run as plain fp32 GEMMs on T
    int n_extra = 0;
    // what gf_smp_create was asked for (smp_derive_plan re-derives cfg / dup_channels / n_extra from ucfg and these), and whether the handle
    // currently runs its `_10` / `_50` levels op by op because the prepared batch cannot be embedded (gf_smp_prepare switches per batch)
    bool req_pad_channels = true, embed_auto_off = false;
    int req_min_pad = 0;
    const float *extra_w = nullptr;   // X of the running pass (set by gf_smp_forward / gf_smp_backward)
    float *extra_g = nullptr;         // ... and its gradient
    float *rs_inv = nullptr;          // [rows of the largest level][2] (1 / tot, tr / tot): the extra products of an op-by-op level
    size_t rs_inv_rows = 0;
    size_t pad_feat_n = 0;
    gfsmp::BatchLayout lay;
    bool prepared = false, forwarded = false;
    // context workspace this batch needs.  gf_smp_prepare only RECORDS it (prepare may run on a loader thread while another
    // handle's step uses the context's workspace on the compute thread); gf_smp_forward / gf_smp_backward grow the workspace, on
    // the thread that owns the context's stream
    size_t ws_need = 0;
    bool has_targets = false;  // the last gf_smp_forward was given targets: only then does dy hold a loss gradient
    // data-parallel reverse sweep (the context has a communicator, gf_dist.hip): the gradient segment of a level is
    // all-reduced on the communicator's stream as soon as it is complete, beside the rest of the sweep
    bool drop_on = false;      // RisiContraction_18_dropout instead of RisiContraction_18 (SMP_sigma_pairgraphs)
    float drop_scale = 1.f;    // test mode: nKept / 18 on every slice
    unsigned *mask_stage = nullptr;   // page-locked staging of the slice masks, [levels][vertices] in node order (gf_smp_dropout_masks)
    size_t mask_stage_n = 0;
    hipEvent_t ev_mask = nullptr;     // the last upload out of mask_stage
    int grad_allreduce = 1;
    float *dp_grads = nullptr;           // gradient buffer of the running gf_smp_backward, null when not data-parallel
    hipEvent_t ev_grad = nullptr, ev_comm = nullptr;
    bool dp_join_pending = false;        // ev_comm was recorded behind the last sweep's all-reduces and nobody has waited for it on the host yet
    int fused = 1;  // use the fused level path where supported (gf_smp_set_fused)
    int bwd_gather = 0;  // fused levels: evaluate dP inside the consumer gather instead of materialising it (GF_SMP_BWD_GATHER)
    // device buffers (owned)
    struct DevLevel {
        int *node_s = nullptr;
        long long *node_row = nullptr, *node_p = nullptr, *node_pair = nullptr;
        float *adj = nullptr, *rsum = nullptr, *rowscale = nullptr, *node_scale = nullptr;
        int *quad_node = nullptr, *quad_b0 = nullptr, *quad_order = nullptr;
        int *pair_node = nullptr, *pair_src_s = nullptr, *cons_s = nullptr;
        long long *pair_src_row = nullptr, *cons_ptr = nullptr, *cons_slab = nullptr, *cons_inv_off = nullptr;
        short *pi = nullptr, *inv = nullptr;
        int4 *cons_hdr = nullptr, *cons_qrec = nullptr;  // tables of smp_bwd_gather_v2 (smp_fused.hip: build_gather_records)
        long long *cons_qbase = nullptr;                 // [nodes of level l-1] first record of the source's consumer entries
        // fused forward level at C = 64 (smp_level_c64_fwd.hip): row panels of whole (node, x) groups, per-row gather indices
        int4 *fwd_pan = nullptr;
        int *cons_of_pair = nullptr;       // [pairs] index of a pair in its source's consumer list (build_node_tables)
        bool node_tables_merged = false;   // this batch's rows-sized tables of the level came from build_node_tables (smp.hip)
        int *fwd_pan_node = nullptr, *node_panel = nullptr;
        int2 *fwd_goff = nullptr;
        int fwd_npanels = 0;
        // Rows (a, b) of T whose source a does not contain vertex b (pi < 0: half of them at QM9 sizes) are structurally zero in
        // the S_ab and T6 blocks.  Their zeros are written ONCE per prepared batch (tables_zero_fill, the first fused forward) and
        // tables-forward then skips them; rowflag[row] = 1 where the row is written every step.  t_zeros goes false whenever
        // something else may have overwritten the T region of Q (an op-by-op forward of the level).
        bool t_zeros = false;
        bool t_filled = false;   // ... and the zeros are physically there (tables_zero_fill ran: some reader of T does not mask them)
        unsigned char *rowflag = nullptr;
        int *trow = nullptr;  // [rows] row of (e, x) for row (x, e) of the same node (compact O layout of the fused C = 64 level)
        int4 *tf_recs = nullptr;  // [2 nNodes] records of tables-forward in launch order (build_tf_records)
        float *psum = nullptr;     // top level, C = 64: [fwd_npanels][64] column sums of the row panels of f_L (readout)
        float *dshl = nullptr;     // towers: [nodes][C] gradient of the level's read-out per node (the fused level's combine-backward adds it)
        bool psum_ready = false;   // ... written by this forward pass
        // per-channel maxima for the weight gradients' column exponents (smp_level_c64_split.hip: smp_wgrad_column_bounds), C = 64:
        float *pmax = nullptr;     // [fwd_npanels][64] largest |f_l| of every row panel, left by combine-forward (levels below the top)
        float *dzmax = nullptr;    // [max(quads, row panels)][64] largest |dz| of every workgroup / panel of combine-backward
        long long dz_rows = 0;     // ... rows of it the last combine-backward wrote
        int dz_ld = 64;            // ... and their width in floats
        long long dz_rows2 = 0, dz_off2 = 0;   // a second set behind them (floats from dzmax), of another width: the workgroup kernel's rows for the
        int dz_ld2 = 64;                       // nodes above 32 positions of a 32- / 16-channel level (panel rows are C floats wide, its rows 32 / 64)
        bool pmax_ready = false;
        void *wimg = nullptr;  // the split product kernels' weight images of this pass (smp_split_build_images), C = 64
        bool wimg_ready = false;
        bool fwd_c64 = false;  // the last forward ran this level's products on the dedicated row-panel kernels (compact O = [O_loc | U])
        int *trowf = nullptr;  // [rows] trow | bit 31: rowflag of the row | bit 30: rowflag of the transposed row (smp_rowpanel_split)
        float max_tot = 0.f, max_tr = 0.f;  // largest |tot|, |tr| of the level's row factors (split-operand weight gradients' column bounds)
        const unsigned *row_max = nullptr;  // the same two as float bits in device memory when the tables are built there
        long long *pair_src_pair = nullptr, *cons_row = nullptr, *cons_pair = nullptr;  // compact diagonal path (smp_prep.h)
        int *node_center = nullptr, *cons_a = nullptr, *mol_order = nullptr, *gather_items = nullptr;
        float *Fdc = nullptr, *Gc = nullptr, *dGc = nullptr, *dFdc = nullptr;  // [pairs of level l-1][2C] each
        float *f = nullptr, *df = nullptr, *Q = nullptr;  // activations [rows][C], their gradient, contraction out [rows][18C]
        // physics towers (every level is read out): per-node sums of f, their LeakyReLU, vertex -> node and node -> molecule maps
        float *sh = nullptr, *vf = nullptr;
        int *node_of_vertex = nullptr, *node_mol = nullptr;
        int *node_present = nullptr;  // [nodes] rows with data of the node (device-built tables)
        int *field = nullptr;  // [pairs] receptive fields back to back (device-built level tables: smp.hip build_level_rows)
        unsigned *keep_mask = nullptr;  // [nodes] slice masks of RisiContraction_18_dropout for this forward (gf_smp_dropout_masks)
        float *nodefac = nullptr, *rowfac8 = nullptr;  // fused levels under slice dropout: [nodes][18] slice factors, [rows][8] per-product row factors
        // fused level (smp_fused.hip): small per-(node,x) / per-node tables and stacked weights
        float *Vt = nullptr, *dVt = nullptr;        // [pairs][4C]  rowsum_a | colsum_b | D8 | D11
        float *St = nullptr, *dSt = nullptr;        // [nodes][4C]  total | s14 | s15 | s18
        float *scal = nullptr;                      // [pairs][4C]  partial scalars owned by (node, b)
        float *Vout = nullptr, *dVout = nullptr;    // [pairs][C]
        float *Sout = nullptr, *dSout = nullptr;    // [nodes][C]
        float *dSpart = nullptr, *dbpart = nullptr; // [pairs][C]
        float *Wst = nullptr, *dWst = nullptr;      // [18][C][C] block-permuted K_l and its gradient
    };
    std::vector<DevLevel> lv;
    // device-built level tables: the batch's adjacency matrices and the per-level statistics the kernels leave behind
    int *mol_nv = nullptr, *mol_adj = nullptr;
    long long *mol_adj_off = nullptr;
    double *mol_coul = nullptr;
    unsigned *tab_stats = nullptr;          // [levels + 1][4]: max |tot|, max |tr| (float bits), rows with data (two words)
    std::vector<unsigned> h_tab_stats;
    std::vector<long long> h_covered;       // [levels + 1] cache of gf_smp_level_covered_rows (-1 = not read yet), cleared by prepare
    // [levels + 1][smp_wgrad_bound_words()] per-channel maxima of f_{l-1} and of df_l and the column bounds the split-operand weight
    // gradients derive from them (smp_level_c64_split.hip: smp_wgrad_column_bounds); C = 64 only, zeroed at the start of every forward
    unsigned *wbound = nullptr;
    float *x = nullptr;      // [nVertices][FD]
    float *P = nullptr;      // shared promotion / dP buffer, max over levels of ppos*C; allocated on first use (ensure_P):
    size_t P_count = 0;      // the fused levels with the folded backward gather never materialise it
    float *sh = nullptr, *vf = nullptr;  // [nNodes][C] readout pre/post activation
    float *dsh = nullptr;                // [nNodes][C] gradient of sh (the fused top level reads it per node)
    float *g = nullptr;      // [nMol][C] graph features
    float *yhat = nullptr, *dy = nullptr;  // [nMol]
    float *colpart = nullptr;  // partial column sums for bias gradients, [colpart_rows][C]
    size_t colpart_rows = 0;
    int *top_node_mol = nullptr, *mol_ptr = nullptr, *mol_nodes = nullptr;
    // device buffers of the current batch come from a pool that survives gf_smp_prepare: a training loop prepares a new
    // batch every step, and hipMalloc of the level buffers (GBs) cost 4x the host graph preparation itself
    struct Block {
        void *p;
        size_t bytes;
        bool used;
        int idle;  // consecutive prepares that did not use the block
    };
    std::vector<Block> pool;
    // Two handles on one context can alternate (prepare of one while the device runs the step of the other): uploads go
    // through the handle's own stream, and recycling the handle's buffers waits for ITS last launch only, not for the stream.
    hipStream_t upload = nullptr;
    hipEvent_t ev_last = nullptr;
    bool used = false;
    // Adam state (gf_smp_adam_step); survives gf_smp_prepare, freed by gf_smp_destroy
    float *adam_m = nullptr, *adam_v = nullptr;
    // handle-owned model (host-pointer mode of the driver): parameters and their gradient, [param_count] each
    float *own_p = nullptr, *own_g = nullptr;
    float *own_t = nullptr, *own_y = nullptr, *own_loss = nullptr, *own_feat = nullptr;  // per-batch, freed by release()
    unsigned long long adam_n = 0;  // parameter elements processed so far (the reference's running beta powers)
};

namespace gf {
// channel counts the row-panel kernel family of the fused level is built for (models are padded to the next one: gf_smp_create)
inline bool smp_panel_channels(int C) { return C == 64 || C == 32 || C == 16; }
// Largest receptive field a fused level takes (round 6): up to 32 positions every kernel of the level; 33 .. 64 at C = 64 -- the few such
// nodes of a level (a 48-atom molecule's level-3 fields reach 35) run tables-forward and the two combine steps on workgroup kernels,
// everything else is row-based and does not care (smp_fused.hip: big_part)
constexpr int kFusedMaxField = 64;
bool smp_fused_supported(const gf_smp *s, int l);
gf_status smp_backward_admissible(const gf_smp *s);   // smp.hip: refusals of a reverse sweep that must come before any work is issued
gf_status smp_fused_forward_level(gf_smp *s, int l, const float *Kl, const float *bl);
gf_status smp_fused_ensure_zero_fill(gf_smp *s, int l);
// node_df != nullptr (top level): df_l is the same C-vector at every position of a node, given as [nodes][C]
gf_status smp_fused_backward_level(gf_smp *s, int l, const float *Kl, float *dKl, float *dbl, const float *node_df, bool rows_too = false);
gf_status smp_fused_gather_backward(gf_smp *s, int l);
bool smp_fused_gather_enabled(const gf_smp *s, int l);
gf_status smp_fused_stack_all(gf_smp *s, const std::vector<const float *> &K);
gf_status smp_build_gather_records(gf_smp *s, int l, hipStream_t stream);
gf_status smp_build_tf_records(gf_smp *s, int l, hipStream_t stream);
gf_status smp_fwd_fused_build_tables(gf_smp *s, int l, hipStream_t stream, bool gather_offsets = true);
gf_status smp_combine_fwd_panels_c64(gf_smp *s, int l, const float *O, const float *bias, float *psum = nullptr, float *pmax = nullptr,
                                     const float *nodefac = nullptr);
// combine-backward on the forward's row panels (C = 64 / 32, compact dO): dzmax = [fwd_npanels][C] per-panel column maxima of |dz| or null
gf_status smp_combine_bwd_panels_c64(gf_smp *s, int l, const float *dfrows, const float *node_df, float *dO, float *dzmax);
gf_status ensure_P(gf_smp *s);
size_t feature_width(const gfsmp::Config &c);  // physics tower: sum over the levels of their channel counts
// level l's K_l / b_l gradients are complete on the context's CURRENT stream (l == 0: H): start their all-reduce
gf_status smp_dp_level_done(gf_smp *s, int l);
}
#endif
