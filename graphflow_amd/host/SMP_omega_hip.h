// SMP_omega_hip.h -- drop-in for the public training / inference API of the reference's SMP_omega / SMP_beta model classes
// (GraphFlow/SMP_omega.h:29-1246) on top of the batched device driver of the C ABI (gf_smp_*, include/gf_hip.h).
//
//   reference                                              here
//   SMP_omega([use_coulomb,] max_nVertices, max_rf,        same arguments; the constructor draws the initial weights from
//             nLevels, nChanels, nFeatures, nDepth[, wl])  rand() exactly as weights_initialization does (:334-338)
//   BatchLearn(nBatch, DenseGraph**, target, lr)  :798     one device pass over the whole batch (loss before, summed
//                                                          gradients, Adam::Learn(lr, nBatch), loss after)
//   Threaded_BatchLearn(...)                      :750     the same update without the two loss evaluations
//   getLoss / Predict / Threaded_Predict / Feature         forward only            (:695, :924, :944, :984)
//   init_multi_threads(nThreads)                  :115     accepted, nothing to do: a batch already runs as one launch
//   save_model / load_model                       :1033    byte-compatible text checkpoints
//
// The molecule type is a template parameter of the methods: anything with the public fields of the reference's
// DenseGraph (nVertices, nFeatures, int **adj, double **feature; GraphFlow/DenseGraph.h:24-119) works, including
// DenseGraph itself when this header is compiled inside the reference tree.  No CPU fallback: every call ends in
// libgf_hip.so and aborts with the library's message if the device path fails (the reference aborts on assert).
#ifndef GF_SMP_OMEGA_HIP_H_INCLUDED
#define GF_SMP_OMEGA_HIP_H_INCLUDED

#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "gf_runtime.h"

class SMP_omega_hip {
public:
    SMP_omega_hip(int max_nVertices, int max_receptive_field, int nLevels, int nChanels, int nFeatures, int nDepth,
                  bool has_WL_ordering = true)
        : max_nVertices(max_nVertices), max_receptive_field(max_receptive_field), nLevels(nLevels), nChanels(nChanels),
          nFeatures(nFeatures), nDepth(nDepth), use_coulomb(false), net(NULL) {
        init(has_WL_ordering);
    }
    // the use_coulomb constructors (SMP_omega.h:71-113): reduced adjacencies from molecule->coulomb
    SMP_omega_hip(bool use_coulomb, int max_nVertices, int max_receptive_field, int nLevels, int nChanels, int nFeatures,
                  int nDepth, bool has_WL_ordering = true)
        : max_nVertices(max_nVertices), max_receptive_field(max_receptive_field), nLevels(nLevels), nChanels(nChanels),
          nFeatures(nFeatures), nDepth(nDepth), use_coulomb(use_coulomb), net(NULL) {
        init(has_WL_ordering);
    }
    ~SMP_omega_hip() { gf_smp_destroy(net); }

protected:
    // the SMP_2D_ver6 / ver7 / ver8 wirings (below): contraction family, CustomMatMulTensor weights, Momentum optimiser
    SMP_omega_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, bool has_WL_ordering,
                  int nContractions, double momentum_param)
        : max_nVertices(max_nVertices), max_receptive_field(max_nVertices), nLevels(nLevels), nChanels(nChanels),
          nFeatures(nFeatures), nDepth(nDepth), use_coulomb(false), net(NULL) {
        wiring_contractions = nContractions;
        wiring_custom = 1;
        momentum = momentum_param;
        use_momentum = true;
        init(has_WL_ordering);
    }

private:
    int wiring_contractions = 0, wiring_custom = 0;
    double momentum = 0.0;
    bool use_momentum = false;
    void init(bool has_WL_ordering) {
        gf_smp_config cfg = {nLevels, nChanels, nFeatures, nDepth, max_receptive_field, has_WL_ordering ? 1 : 0,
                             wiring_contractions, wiring_custom};
        must(gf_smp_create(gfhost::default_context(), &cfg, &net), "gf_smp_create");
        std::vector<float> w(gf_smp_param_count(net));
        must(gf_smp_uniform_init_host(&cfg, &w[0]), "gf_smp_uniform_init_host");  // weights_initialization()
        must(gf_smp_parameters_upload(net, &w[0]), "gf_smp_parameters_upload");
    }

public:
    void init_multi_threads(int) {}

    template <class Graph>
    double getLoss(int nBatch, Graph **molecule, double *target) {
        bind(nBatch, molecule);
        std::vector<double> loss(nBatch);
        must(gf_smp_forward_host(net, target, NULL, &loss[0], NULL), "gf_smp_forward_host");
        double total = 0.0;
        for (int i = 0; i < nBatch; ++i) total += loss[i];
        return total;
    }

    template <class Graph>
    std::pair<double, double> BatchLearn(int nBatch, Graph **molecule, double *target, double learning_rate) {
        std::pair<double, double> ret;
        ret.first = getLoss(nBatch, molecule, target);  // leaves the batch bound and forwarded
        step(learning_rate, nBatch);
        std::vector<double> loss(nBatch);
        must(gf_smp_forward_host(net, target, NULL, &loss[0], NULL), "gf_smp_forward_host");
        ret.second = 0.0;
        for (int i = 0; i < nBatch; ++i) ret.second += loss[i];
        return ret;
    }

    template <class Graph>
    void Threaded_BatchLearn(int nBatch, Graph **molecule, double *target, double learning_rate) {
        bind(nBatch, molecule);
        must(gf_smp_forward_host(net, target, NULL, NULL, NULL), "gf_smp_forward_host");
        step(learning_rate, nBatch);
    }

    template <class Graph>
    double Predict(Graph *molecule) {
        double y = 0.0;
        Threaded_Predict(1, &molecule, &y);
        return y;
    }

    template <class Graph>
    void Threaded_Predict(int nBatch, Graph **molecule, double *predict) {
        bind(nBatch, molecule);
        must(gf_smp_forward_host(net, NULL, predict, NULL, NULL), "gf_smp_forward_host");
    }

    template <class Graph>
    std::vector<double> Feature(Graph *molecule) {
        bind(1, &molecule);
        std::vector<double> f(nChanels);
        must(gf_smp_forward_host(net, NULL, NULL, NULL, &f[0]), "gf_smp_forward_host");
        return f;
    }

    void save_model(std::string filename) { must(gf_smp_save_model(net, NULL, filename.c_str()), "gf_smp_save_model"); }
    void load_model(std::string filename) { must(gf_smp_load_model(net, NULL, filename.c_str()), "gf_smp_load_model"); }

    // flat views of sgd->params[i]->value / ->gradient in registration order (H, K_1, b_1, ..., W)
    std::vector<float> parameters() {
        std::vector<float> p(gf_smp_param_count(net));
        must(gf_smp_parameters_download(net, &p[0], NULL), "gf_smp_parameters_download");
        return p;
    }
    std::vector<float> gradients() {
        std::vector<float> g(gf_smp_param_count(net));
        must(gf_smp_parameters_download(net, NULL, &g[0]), "gf_smp_parameters_download");
        return g;
    }
    void set_parameters(const std::vector<float> &p) { must(gf_smp_parameters_upload(net, &p[0]), "gf_smp_parameters_upload"); }

    int max_nVertices, max_receptive_field, nLevels, nChanels, nFeatures, nDepth;
    bool use_coulomb;

private:
    // molecule->coulomb when the molecule type has one (DenseGraph does), NULL otherwise
    template <class Graph>
    static auto coulomb_of(const Graph *g, int) -> decltype(&g->coulomb[0][0], (double **)0) { return g->coulomb; }
    template <class Graph>
    static double **coulomb_of(const Graph *, long) { return NULL; }

    // DenseGraph** -> the flat batch of gf_smp_prepare (host graph preparation + index upload)
    template <class Graph>
    void bind(int nBatch, Graph **molecule) {
        nV.resize(nBatch);
        adj.clear();
        feature.clear();
        coulomb.clear();
        for (int m = 0; m < nBatch; ++m) {
            const Graph *g = molecule[m];
            if (g->nVertices > max_nVertices || g->nFeatures != nFeatures) {
                std::fprintf(stderr, "SMP_omega_hip: molecule %d has %d vertices / %d features (model: <= %d / %d)\n", m,
                             g->nVertices, g->nFeatures, max_nVertices, nFeatures);
                std::abort();
            }
            nV[m] = g->nVertices;
            for (int i = 0; i < g->nVertices; ++i) {
                adj.insert(adj.end(), g->adj[i], g->adj[i] + g->nVertices);
                feature.insert(feature.end(), g->feature[i], g->feature[i] + nFeatures);
            }
            if (use_coulomb) {
                double **cm = coulomb_of(g, 0);
                if (!cm) {
                    std::fprintf(stderr, "SMP_omega_hip: use_coulomb needs a molecule type with a `coulomb` matrix\n");
                    std::abort();
                }
                for (int i = 0; i < g->nVertices; ++i) coulomb.insert(coulomb.end(), cm[i], cm[i] + g->nVertices);
            }
        }
        must(gf_smp_prepare_coulomb(net, nBatch, &nV[0], &adj[0], &feature[0], use_coulomb ? &coulomb[0] : NULL),
             "gf_smp_prepare");
    }
    void step(double learning_rate, int nBatch) {
        must(gf_smp_backward(net, NULL, NULL, 0), "gf_smp_backward");
        if (use_momentum)
            must(gf_smp_momentum_step(net, NULL, NULL, learning_rate, nBatch, momentum), "gf_smp_momentum_step");
        else
            must(gf_smp_adam_step(net, NULL, NULL, learning_rate, nBatch), "gf_smp_adam_step");
    }
    void must(gf_status st, const char *what) {
        if (st != GF_OK) gfhost::die(gfhost::default_context(), what, st);
    }
    gf_smp *net;
    std::vector<int> nV, adj;
    std::vector<double> feature, coulomb;
};

// SMP_beta (GraphFlow/SMP_beta.h:29-1190): the same model without the receptive-field cap -- the constructors drop the
// max_receptive_field argument (:31, :48, :65, :82); everything else is inherited.
class SMP_beta_hip : public SMP_omega_hip {
public:
    SMP_beta_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, bool has_WL_ordering = true)
        : SMP_omega_hip(max_nVertices, max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL_ordering) {}
    SMP_beta_hip(bool use_coulomb, int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth,
                 bool has_WL_ordering = true)
        : SMP_omega_hip(use_coulomb, max_nVertices, max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL_ordering) {}
};

// SMP_2D_ver6 / ver7 / ver8 (GraphFlow/SMP_2D_ver6.h:30-930 and its two siblings): the same DAG with RisiContraction_10 /
// _50 / _18, the level weight [C][nContractions C] applied by CustomMatMulTensor, no receptive-field cap, and the
// Momentum optimiser (sgd = new Momentum(momentum_param), :204).  Same public methods as above.
class SMP_2D_ver6_hip : public SMP_omega_hip {
public:
    SMP_2D_ver6_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, double momentum_param,
                    bool has_WL_ordering = true)
        : SMP_omega_hip(max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL_ordering, 10, momentum_param) {}
};
class SMP_2D_ver7_hip : public SMP_omega_hip {
public:
    SMP_2D_ver7_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, double momentum_param,
                    bool has_WL_ordering = true)
        : SMP_omega_hip(max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL_ordering, 50, momentum_param) {}
};
class SMP_2D_ver8_hip : public SMP_omega_hip {
public:
    SMP_2D_ver8_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures, int nDepth, double momentum_param,
                    bool has_WL_ordering = true)
        : SMP_omega_hip(max_nVertices, nLevels, nChanels, nFeatures, nDepth, has_WL_ordering, 18, momentum_param) {}
};

#endif
