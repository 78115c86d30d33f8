// smp_model.hip -- the `_physics` and `_pairgraphs` SMP models of GraphFlow as one handle:
//   SMP_omega_physics / SMP_beta_physics            one tower + a one-hidden-layer head       (GraphFlow/SMP_omega_physics.h:29-606)
//   SMP_omega_pairgraphs / SMP_beta_pairgraphs      two towers (a graph and its partner, e.g. the line graph) + a two-hidden-layer
//                                                   head                                    (GraphFlow/SMP_omega_pairgraphs.h:81-730)
//   SMP_sigma_pairgraphs                            the same with RisiContraction_18_dropout   (GraphFlow/SMP_sigma_pairgraphs.h)
// A tower is a gf_smp handle with cfg.physics = 1 (smp.hip): raw features, channels halving per level, every level read out.
// The model's parameter / gradient vectors are flat device buffers in the REFERENCE's registration order:
//   physics     H, (K_l, b_l) l = 1..L, W1, W2                                                 (SMP_omega_physics.h:254-262)
//   pairgraphs  H_1, H_2, (K1_l, b1_l, K2_l, b2_l) l = 1..L, W1, W2, W3                         (SMP_omega_pairgraphs.h:361-375)
// and the feature row handed to the head concatenates the level features level by level, tower 1 before tower 2 inside a
// level (SMP_omega_pairgraphs.h:699-704).  The towers keep contiguous copies of their own parameters / gradients; segments
// are copied device to device around every pass.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "libc_random.h"
#include "smp_internal.h"

struct gf_smp_model {
    gf_ctx *ctx = nullptr;
    gf_smp_model_config cfg;
    int nTowers = 1, L = 0;
    gf_smp *tower[2] = {nullptr, nullptr};
    size_t tower_params[2] = {0, 0}, head_params = 0, n_params = 0;
    std::vector<int> widths;                 // head layer widths: [feature width, hidden..., ]
    int nLayers = 0;
    std::vector<int> lvlC;                   // channels per level
    int fwidth = 0;                          // feature columns of ONE tower
    struct Seg { size_t model_off, tower_off, n; };
    std::vector<Seg> segs[2];
    size_t head_off = 0;
    // device buffers (own)
    float *tp[2] = {nullptr, nullptr}, *tg[2] = {nullptr, nullptr};  // tower parameter / gradient copies
    float *feat[2] = {nullptr, nullptr}, *dfeat[2] = {nullptr, nullptr};
    float *x = nullptr, *dx = nullptr, *work = nullptr;
    // [L + 2] column offsets of the levels inside a tower's feature row (configuration only).  Allocated by create, UPLOADED by the
    // first prepare: create must not copy anything to the device -- a host-to-device copy makes the HIP runtime draw from rand()
    // (observed: a varying number of draws), and the reference's classes are constructed right after srand(seed) with the weights
    // drawn from rand() next (gf_smp_model_uniform_init_host): the same seed has to give the same model.
    int *lvl_off = nullptr;
    std::vector<int> lvl_off_host;
    bool lvl_off_uploaded = false;
    int nMol = 0, cap_mol = 0;
    bool train = true, forwarded = false;
    std::vector<int> nV[2];
    // handle-owned model for the host-pointer mode (C++ classes): parameters, gradients, Adam moments
    float *own_p = nullptr, *own_g = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    unsigned long long adam_n = 0;
    float *own_t = nullptr, *own_y = nullptr, *own_loss = nullptr;
};

namespace gf {
namespace {

// x[m][...] = level-major interleave of the towers' feature rows (reverse: scatter of dx)
__global__ void interleave_features(const float *__restrict__ f0, const float *__restrict__ f1, float *__restrict__ x, int nTowers, int fwidth,
                                    const int *__restrict__ lvl_off, int nLevels1, int reverse, float *__restrict__ d0, float *__restrict__ d1) {
    const int m = blockIdx.x;
    const int xw = nTowers * fwidth;
    for (int c = threadIdx.x; c < fwidth; c += blockDim.x) {
        int l = 0;
        while (l + 1 < nLevels1 && c >= lvl_off[l + 1]) ++l;
        const int wl = lvl_off[l + 1] - lvl_off[l];
        for (int t = 0; t < nTowers; ++t) {
            const int xc = nTowers * lvl_off[l] + t * wl + (c - lvl_off[l]);
            if (reverse)
                (t ? d1 : d0)[(size_t)m * fwidth + c] = x[(size_t)m * xw + xc];
            else
                x[(size_t)m * xw + xc] = (t ? f1 : f0)[(size_t)m * fwidth + c];
        }
    }
}

__global__ void add_into(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

gf_status dadd(gf_ctx *ctx, float *dst, const float *src, size_t n) {
    if (!n) return GF_OK;
    const size_t b = (n + 255) / 256;
    GF_LAUNCH(ctx, "model_add", add_into, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, dst, src, n);
    return GF_OK;
}

gf_status dcopy(gf_ctx *ctx, float *dst, const float *src, size_t n) {
    if (n) GF_HIP_TRY(ctx, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return GF_OK;
}

void free_batch(gf_smp_model *m) {
    float **bufs[] = {&m->feat[0], &m->feat[1], &m->dfeat[0], &m->dfeat[1], &m->x, &m->dx, &m->work, &m->own_t, &m->own_y, &m->own_loss};
    for (float **b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    m->cap_mol = 0;
}

}  // namespace
}  // namespace gf

using gf::fail;

extern "C" {

gf_status gf_smp_model_create(gf_ctx *ctx, const gf_smp_model_config *cfg, gf_smp_model **out) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!cfg || !out) return fail(ctx, GF_ERR_INVALID, "gf_smp_model_create: null argument");
    if (cfg->nTowers < 1 || cfg->nTowers > 2 || cfg->nLevels < 1 || cfg->nChanels < 1 || cfg->max_receptive_field < 1 || cfg->nKept < 0 ||
        cfg->nKept > 18)
        return fail(ctx, GF_ERR_INVALID, "gf_smp_model_create: bad configuration");
    gf_smp_model *m = new gf_smp_model();
    m->ctx = ctx;
    m->cfg = *cfg;
    m->nTowers = cfg->nTowers;
    m->L = cfg->nLevels;
    for (int l = 0; l <= m->L; ++l) {
        int c = cfg->nChanels >> l;
        m->lvlC.push_back(c < 1 ? 1 : c);
        m->fwidth += m->lvlC.back();
    }
    size_t off = 0;
    // the H matrices come first (one per tower), then the levels, towers interleaved inside a level
    size_t toff[2] = {0, 0};
    for (int t = 0; t < m->nTowers; ++t) {
        gf_smp_config tc = {cfg->nLevels, cfg->nChanels, cfg->nFeatures[t], 0, cfg->max_receptive_field, 0, 18, 0, 1};
        // (nKept > 0, RisiContraction_18_dropout: towers of up to 32 channels run the fused levels with per-product slice factors since
        //  round 5 -- padded like the others; wider ones keep their levels op by op, at their own halving widths)
        gf_status st = gf::smp_create(ctx, &tc, /*pad_channels=*/cfg->nKept <= 0 || cfg->nChanels <= 32, &m->tower[t]);
        if (st != GF_OK) {
            gf_smp_model_destroy(m);
            return st;
        }
        m->tower_params[t] = gf_smp_param_count(m->tower[t]);
        const size_t nH = (size_t)cfg->nChanels * cfg->nFeatures[t];
        m->segs[t].push_back({off, 0, nH});
        off += nH;
        toff[t] = nH;
    }
    for (int l = 1; l <= m->L; ++l)
        for (int t = 0; t < m->nTowers; ++t) {
            const size_t n = (size_t)18 * m->lvlC[l - 1] * m->lvlC[l] + m->lvlC[l];
            m->segs[t].push_back({off, toff[t], n});
            off += n;
            toff[t] += n;
        }
    m->head_off = off;
    // head widths: physics nTotal -> nTotal / 2 -> 1 (:229-238); pairgraphs nTotal -> max(nTotal / 2, 10) -> max(that / 2, 10) -> 1
    const int nTotal = m->nTowers * m->fwidth;
    m->widths.push_back(nTotal);
    if (m->nTowers == 1) {
        m->widths.push_back(nTotal / 2);
    } else {
        const int h1 = nTotal / 2 > 10 ? nTotal / 2 : 10, h2 = h1 / 2 > 10 ? h1 / 2 : 10;
        m->widths.push_back(h1);
        m->widths.push_back(h2);
    }
    m->nLayers = (int)m->widths.size() - 1;
    if (m->widths[1] < 1) {
        gf_smp_model_destroy(m);
        return fail(ctx, GF_ERR_INVALID, "gf_smp_model_create: %d feature columns leave no hidden units (the reference uses nTotal / 2)", nTotal);
    }
    m->head_params = gf_head_param_count(m->nLayers, m->widths.data());
    m->n_params = off + m->head_params;
    for (int t = 0; t < m->nTowers; ++t) {
        if (hipMalloc(reinterpret_cast<void **>(&m->tp[t]), m->tower_params[t] * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&m->tg[t]), m->tower_params[t] * sizeof(float)) != hipSuccess) {
            gf_smp_model_destroy(m);
            return fail(ctx, GF_ERR_NOMEM, "gf_smp_model_create: device allocation failed");
        }
    }
    m->lvl_off_host.assign(m->L + 2, 0);
    for (int l = 0; l <= m->L; ++l) m->lvl_off_host[l + 1] = m->lvl_off_host[l] + m->lvlC[l];
    if (hipMalloc(reinterpret_cast<void **>(&m->lvl_off), sizeof(int) * m->lvl_off_host.size()) != hipSuccess) {
        gf_smp_model_destroy(m);
        return fail(ctx, GF_ERR_NOMEM, "gf_smp_model_create: device allocation failed");
    }
    *out = m;
    return GF_OK;
}

gf_status gf_smp_model_destroy(gf_smp_model *m) {
    if (!m) return GF_OK;
    if (m->ctx) (void)hipStreamSynchronize(m->ctx->stream);
    for (int t = 0; t < 2; ++t) {
        if (m->tower[t]) gf_smp_destroy(m->tower[t]);
        if (m->tp[t]) (void)hipFree(m->tp[t]);
        if (m->tg[t]) (void)hipFree(m->tg[t]);
    }
    gf::free_batch(m);
    if (m->lvl_off) (void)hipFree(m->lvl_off);
    float *own[] = {m->own_p, m->own_g, m->adam_m, m->adam_v};
    for (float *p : own)
        if (p) (void)hipFree(p);
    delete m;
    return GF_OK;
}

size_t gf_smp_model_param_count(const gf_smp_model *m) { return m ? m->n_params : 0; }

gf_status gf_smp_model_set_mode(gf_smp_model *m, int train) {
    if (!m) return fail(nullptr, GF_ERR_INVALID, "null model");
    m->train = train != 0;
    return GF_OK;
}

gf_status gf_smp_model_prepare(gf_smp_model *m, int nMol, const int *nVertices1, const int *adj1, const double *feature1,
                               const int *nVertices2, const int *adj2, const double *feature2) {
    if (!m) return fail(nullptr, GF_ERR_INVALID, "null model");
    gf_ctx *ctx = m->ctx;
    if (nMol < 1 || !nVertices1 || !adj1 || !feature1 || (m->nTowers == 2 && (!nVertices2 || !adj2 || !feature2)))
        return fail(ctx, GF_ERR_INVALID, "gf_smp_model_prepare: bad argument");
    const int *nv[2] = {nVertices1, nVertices2}, *ad[2] = {adj1, adj2};
    const double *fe[2] = {feature1, feature2};
    for (int t = 0; t < m->nTowers; ++t) {
        gf_status st = gf_smp_prepare(m->tower[t], nMol, nv[t], ad[t], fe[t]);
        if (st != GF_OK) return st;
        m->nV[t].assign(nv[t], nv[t] + nMol);
    }
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!m->lvl_off_uploaded) {  // (not in create: see lvl_off)
        GF_HIP_TRY(ctx, hipMemcpyAsync(m->lvl_off, m->lvl_off_host.data(), sizeof(int) * m->lvl_off_host.size(), hipMemcpyHostToDevice, ctx->stream));
        m->lvl_off_uploaded = true;
    }
    if (nMol > m->cap_mol) {
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        gf::free_batch(m);
        const size_t fw = (size_t)nMol * m->fwidth, xw = (size_t)nMol * m->widths[0];
        for (int t = 0; t < m->nTowers; ++t) {
            GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->feat[t]), fw * sizeof(float)));
            GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->dfeat[t]), fw * sizeof(float)));
        }
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->x), xw * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->dx), xw * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->work), gf_head_work_floats(m->nLayers, m->widths.data(), nMol) * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->own_t), (size_t)nMol * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->own_y), (size_t)nMol * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->own_loss), (size_t)nMol * sizeof(float)));
        if (gf::poison_buffers()) {  // GF_POISON=1 (debug): NaN patterns in everything nobody has written yet
            float *bufs[] = {m->feat[0], m->feat[1], m->dfeat[0], m->dfeat[1], m->x, m->dx, m->work};
            const size_t n[] = {fw, fw, fw, fw, xw, xw, gf_head_work_floats(m->nLayers, m->widths.data(), nMol)};
            for (int i = 0; i < 7; ++i)
                if (bufs[i]) GF_HIP_TRY(ctx, hipMemsetAsync(bufs[i], 0xff, n[i] * sizeof(float), ctx->stream));  // (ordered with the kernels)
        }
        m->cap_mol = nMol;
    }
    m->nMol = nMol;
    m->forwarded = false;
    return GF_OK;
}

gf_status gf_smp_model_forward(gf_smp_model *m, const float *params, const float *targets, float *predict, float *loss) {
    if (!m) return fail(nullptr, GF_ERR_INVALID, "null model");
    gf_ctx *ctx = m->ctx;
    if (!params) params = m->own_p;
    if (!params || m->nMol < 1) return fail(ctx, GF_ERR_INVALID, "gf_smp_model_forward: no parameters / no prepared batch");
    gf_status st;
    // RisiContraction_18_dropout: the reference draws nKept slices with rand() in every contraction's forward(), sample by sample,
    // tower 1 (levels 1..L, vertices in order) before tower 2 (SMP_sigma_pairgraphs.h:596-660, RisiContraction_18_dropout.h:113-125)
    if (m->cfg.nKept > 0) {
        std::vector<unsigned> masks[2];
        std::vector<int> first[2];
        int totalV[2] = {0, 0};
        for (int t = 0; t < m->nTowers; ++t) {
            first[t].assign(m->nMol + 1, 0);
            for (int i = 0; i < m->nMol; ++i) first[t][i + 1] = first[t][i] + m->nV[t][i];
            totalV[t] = first[t][m->nMol];
            masks[t].assign((size_t)m->L * totalV[t], 0x3ffffu);
        }
        if (m->train) {
            gf::LibcRandom rng(true);   // (libc's own stream, stepped inline; handed back when this scope ends)
            for (int i = 0; i < m->nMol; ++i)
                for (int t = 0; t < m->nTowers; ++t)
                    for (int l = 1; l <= m->L; ++l)
                        for (int v = 0; v < m->nV[t][i]; ++v) {
                            unsigned use = 0;
                            for (int k = 0; k < m->cfg.nKept; ++k)
                                for (;;) {
                                    const int j = rng.next() % 18;
                                    if (!((use >> j) & 1u)) {
                                        use |= 1u << j;
                                        break;
                                    }
                                }
                            masks[t][(size_t)(l - 1) * totalV[t] + first[t][i] + v] = use;
                        }
        }
        for (int t = 0; t < m->nTowers; ++t) {
            st = gf_smp_dropout_masks(m->tower[t], masks[t].data(), m->train ? 1.f : (float)m->cfg.nKept / 18.f);
            if (st != GF_OK) return st;
        }
    }
    for (int t = 0; t < m->nTowers; ++t) {
        for (const gf_smp_model::Seg &sg : m->segs[t]) {
            st = gf::dcopy(ctx, m->tp[t] + sg.tower_off, params + sg.model_off, sg.n);
            if (st != GF_OK) return st;
        }
        st = gf_smp_forward(m->tower[t], m->tp[t], nullptr, nullptr, nullptr, m->feat[t]);
        if (st != GF_OK) return st;
    }
    const int *d_off = m->lvl_off;
    GF_LAUNCH(ctx, "model_interleave", gf::interleave_features, dim3(m->nMol), dim3(64), 0, m->feat[0], m->feat[1], m->x, m->nTowers, m->fwidth,
              d_off, m->L + 1, 0, (float *)nullptr, (float *)nullptr);
    st = gf_head_forward_f32(ctx, m->nLayers, m->widths.data(), m->x, m->nMol, params + m->head_off, targets, predict, loss, m->work);
    if (st != GF_OK) return st;
    m->forwarded = targets != nullptr;
    return GF_OK;
}

gf_status gf_smp_model_backward(gf_smp_model *m, const float *params, float *grads, int accumulate) {
    if (!m) return fail(nullptr, GF_ERR_INVALID, "null model");
    gf_ctx *ctx = m->ctx;
    if (!params && !grads) {
        params = m->own_p;
        grads = m->own_g;
    }
    if (!params || !grads) return fail(ctx, GF_ERR_INVALID, "gf_smp_model_backward: null argument");
    if (!m->forwarded) return fail(ctx, GF_ERR_INVALID, "gf_smp_model_backward: needs a forward with targets first");
    gf_status st;
    for (int t = 0; t < m->nTowers; ++t) {   // (a tower's refusal comes before the head has written anything)
        st = gf::smp_backward_admissible(m->tower[t]);
        if (st != GF_OK) return st;
    }
    if (!accumulate) GF_HIP_TRY(ctx, hipMemsetAsync(grads, 0, m->n_params * sizeof(float), ctx->stream));
    st = gf_head_backward_f32(ctx, m->nLayers, m->widths.data(), m->x, m->nMol, params + m->head_off, m->work, m->dx, grads + m->head_off);
    if (st != GF_OK) return st;
    const int *d_off = m->lvl_off;  // (its own buffer: the head's activations in `work` stay valid for a repeated backward)
    GF_LAUNCH(ctx, "model_interleave", gf::interleave_features, dim3(m->nMol), dim3(64), 0, (const float *)nullptr, (const float *)nullptr, m->dx,
              m->nTowers, m->fwidth, d_off, m->L + 1, 1, m->dfeat[0], m->dfeat[1]);
    for (int t = 0; t < m->nTowers; ++t) {
        st = gf_smp_backward_features(m->tower[t], m->tp[t], m->tg[t], m->dfeat[t], 0);
        if (st != GF_OK) return st;
        // model gradient (+)= tower gradient, segment by segment
        for (const gf_smp_model::Seg &sg : m->segs[t]) {
            if (accumulate) {
                st = gf::dadd(ctx, grads + sg.model_off, m->tg[t] + sg.tower_off, sg.n);
            } else {
                st = gf::dcopy(ctx, grads + sg.model_off, m->tg[t] + sg.tower_off, sg.n);
            }
            if (st != GF_OK) return st;
        }
    }
    return GF_OK;
}

// weights_initialization of the model classes (SMP_omega_physics.h:291-295 -> GraphFlow::uniform_init, GraphFlow.h:1297-1306) over
// the parameters in registration order, drawn from rand(): the same srand() gives the reference's initial weights.  Host buffer.
gf_status gf_smp_model_uniform_init_host(const gf_smp_model *m, float *params) {
    if (!m || !params) return GF_ERR_INVALID;
    std::vector<size_t> sizes;
    const int C = m->cfg.nChanels;
    for (int t = 0; t < m->nTowers; ++t) sizes.push_back((size_t)C * m->cfg.nFeatures[t]);
    for (int l = 1; l <= m->L; ++l)
        for (int t = 0; t < m->nTowers; ++t) {
            sizes.push_back((size_t)18 * m->lvlC[l - 1] * m->lvlC[l]);
            sizes.push_back((size_t)m->lvlC[l]);
        }
    for (int i = 1; i <= m->nLayers; ++i) sizes.push_back((size_t)m->widths[i] * m->widths[i - 1]);
    sizes.push_back((size_t)m->widths[m->nLayers]);
    size_t off = 0;
    for (size_t v = 0; v < sizes.size(); ++v)
        for (size_t i = 0; i < sizes[v]; ++i) {
            double x = (double)(rand() % 10) / (10.0 * (double)sizes[v]);
            if (rand() % 2 == 1) x = -x;
            params[off++] = (float)x;
        }
    return off == m->n_params ? GF_OK : GF_ERR_INVALID;
}

// ---- host-pointer mode (the C++ model classes): the handle owns parameters, gradients and the Adam moments ---------------
gf_status gf_smp_model_parameters_upload(gf_smp_model *m, const float *host_params) {
    if (!m || !host_params) return fail(m ? m->ctx : nullptr, GF_ERR_INVALID, "gf_smp_model_parameters_upload: null argument");
    gf_ctx *ctx = m->ctx;
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!m->own_p) {
        float **bufs[] = {&m->own_p, &m->own_g, &m->adam_m, &m->adam_v};
        for (float **b : bufs) {
            GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(b), m->n_params * sizeof(float)));
            GF_HIP_TRY(ctx, hipMemsetAsync(*b, 0, m->n_params * sizeof(float), ctx->stream));
        }
        m->adam_n = 0;
    }
    GF_HIP_TRY(ctx, hipMemcpyAsync(m->own_p, host_params, m->n_params * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}

gf_status gf_smp_model_parameters_download(gf_smp_model *m, float *host_params, float *host_grads) {
    if (!m || !m->own_p) return fail(m ? m->ctx : nullptr, GF_ERR_INVALID, "gf_smp_model_parameters_download: no handle-owned model");
    gf_ctx *ctx = m->ctx;
    if (host_params) GF_HIP_TRY(ctx, hipMemcpyAsync(host_params, m->own_p, m->n_params * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (host_grads) GF_HIP_TRY(ctx, hipMemcpyAsync(host_grads, m->own_g, m->n_params * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}

// forward on the handle-owned parameters with host targets / results (targets NULL: predict only).  Blocking.
gf_status gf_smp_model_forward_host(gf_smp_model *m, const double *targets, double *predict, double *loss) {
    if (!m || !m->own_p) return fail(m ? m->ctx : nullptr, GF_ERR_INVALID, "gf_smp_model_forward_host: no handle-owned model");
    gf_ctx *ctx = m->ctx;
    std::vector<float> tmp((size_t)m->nMol);
    if (targets) {
        for (int i = 0; i < m->nMol; ++i) tmp[i] = (float)targets[i];
        GF_HIP_TRY(ctx, hipMemcpyAsync(m->own_t, tmp.data(), sizeof(float) * m->nMol, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    gf_status st = gf_smp_model_forward(m, m->own_p, targets ? m->own_t : nullptr, m->own_y, m->own_loss);
    if (st != GF_OK) return st;
    struct Out { double *dst; const float *src; } outs[2] = {{predict, m->own_y}, {targets ? loss : nullptr, m->own_loss}};
    for (const Out &o : outs) {
        if (!o.dst) continue;
        GF_HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), o.src, sizeof(float) * m->nMol, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < m->nMol; ++i) o.dst[i] = (double)tmp[i];
    }
    return GF_OK;
}

// sgd->Learn(learning_rate, nBatch) on the handle-owned model (Adam over the whole registration-order vector)
gf_status gf_smp_model_adam_step(gf_smp_model *m, double learning_rate, int nBatch) {
    if (!m || !m->own_p) return fail(m ? m->ctx : nullptr, GF_ERR_INVALID, "gf_smp_model_adam_step: no handle-owned model");
    gf_status st = gf_adam_step_f32(m->ctx, m->own_p, m->own_g, m->adam_m, m->adam_v, m->n_params, learning_rate, nBatch, m->adam_n);
    if (st == GF_OK) m->adam_n += m->n_params;
    return st;
}

}  // extern "C"
