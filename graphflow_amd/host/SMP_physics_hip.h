// SMP_physics_hip.h -- drop-ins for the `_physics` and `_pairgraphs` model classes of the reference on top of gf_smp_model_*
// (include/gf_hip.h):
//   SMP_omega_physics_hip(max_nVertices, max_receptive_field, nLevels, nChanels, nFeatures)      GraphFlow/SMP_omega_physics.h:31
//   SMP_beta_physics_hip(max_nVertices, nLevels, nChanels, nFeatures)                              GraphFlow/SMP_beta_physics.h:31
//   SMP_omega_pairgraphs_hip(max_nV_1, max_nV_2, max_rf, nLevels, nChanels, nFeatures_1, _2)       GraphFlow/SMP_omega_pairgraphs.h:81
//   SMP_beta_pairgraphs_hip(max_nV_1, max_nV_2, nLevels, nChanels, nFeatures_1, _2)                GraphFlow/SMP_beta_pairgraphs.h:81
//   SMP_sigma_pairgraphs_hip(..., nKept) + setMode / setTrainMode / setTestMode                    GraphFlow/SMP_sigma_pairgraphs.h:81,139
// with the reference's public training / inference methods: BatchLearn, Threaded_BatchLearn, getLoss, Predict,
// Threaded_Predict, init_multi_threads (accepted, nothing to do: a batch is one device pass), save_model / load_model (the
// text format of SMP_omega_physics.h:927-949: every parameter value in registration order).  The constructors draw the initial
// weights from rand() exactly as weights_initialization does: the same srand() starts from the reference's model.
// The molecule type is a template parameter (any type with DenseGraph's public nVertices / nFeatures / adj / feature).
// No CPU fallback: every call ends in libgf_hip.so and aborts with the library's message if the device path fails.
#ifndef GF_SMP_PHYSICS_HIP_H_INCLUDED
#define GF_SMP_PHYSICS_HIP_H_INCLUDED

#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "gf_runtime.h"

class SMP_model_hip {
protected:
    SMP_model_hip(int nTowers, int maxV1, int maxV2, int max_rf, int nLevels, int nChanels, int F1, int F2, int nKept)
        : net(NULL), towers(nTowers) {
        maxV[0] = maxV1;
        maxV[1] = maxV2;
        nF[0] = F1;
        nF[1] = F2;
        gf_smp_model_config cfg = {nTowers, nLevels, nChanels, max_rf, {F1, F2}, nKept};
        must(gf_smp_model_create(gfhost::default_context(), &cfg, &net), "gf_smp_model_create");
        std::vector<float> w(gf_smp_model_param_count(net));
        must(gf_smp_model_uniform_init_host(net, &w[0]), "gf_smp_model_uniform_init_host");  // weights_initialization()
        must(gf_smp_model_parameters_upload(net, &w[0]), "gf_smp_model_parameters_upload");
    }

public:
    ~SMP_model_hip() { gf_smp_model_destroy(net); }
    void init_multi_threads(int) {}
    void setMode(bool train) { must(gf_smp_model_set_mode(net, train ? 1 : 0), "gf_smp_model_set_mode"); }
    void setTrainMode() { setMode(true); }
    void setTestMode() { setMode(false); }

    void save_model(std::string filename) {
        std::vector<float> p = parameters();
        FILE *f = std::fopen(filename.c_str(), "w");
        if (!f) die("save_model: cannot open file");
        for (size_t i = 0; i < p.size(); ++i) std::fprintf(f, "%g ", (double)p[i]);
        std::fclose(f);
    }
    void load_model(std::string filename) {
        std::vector<float> p(gf_smp_model_param_count(net));
        FILE *f = std::fopen(filename.c_str(), "r");
        if (!f) die("load_model: cannot open file");
        double v;
        size_t got = 0;
        while (got < p.size() && std::fscanf(f, "%lf", &v) == 1) p[got++] = (float)v;
        std::fclose(f);
        if (got != p.size()) die("load_model: the file holds fewer values than the model has parameters");
        must(gf_smp_model_parameters_upload(net, &p[0]), "gf_smp_model_parameters_upload");
    }
    std::vector<float> parameters() {
        std::vector<float> p(gf_smp_model_param_count(net));
        must(gf_smp_model_parameters_download(net, &p[0], NULL), "gf_smp_model_parameters_download");
        return p;
    }
    std::vector<float> gradients() {
        std::vector<float> g(gf_smp_model_param_count(net));
        must(gf_smp_model_parameters_download(net, NULL, &g[0]), "gf_smp_model_parameters_download");
        return g;
    }

protected:
    template <class Graph>
    void pack(int t, int nBatch, Graph **molecule) {
        nV[t].resize(nBatch);
        adj[t].clear();
        feature[t].clear();
        for (int m = 0; m < nBatch; ++m) {
            const Graph *g = molecule[m];
            if (g->nVertices > maxV[t] || g->nFeatures != nF[t]) die("a molecule exceeds max_nVertices or has the wrong feature count");
            nV[t][m] = g->nVertices;
            for (int i = 0; i < g->nVertices; ++i) {
                adj[t].insert(adj[t].end(), g->adj[i], g->adj[i] + g->nVertices);
                feature[t].insert(feature[t].end(), g->feature[i], g->feature[i] + nF[t]);
            }
        }
    }
    void prepare(int nBatch) {
        must(gf_smp_model_prepare(net, nBatch, &nV[0][0], &adj[0][0], &feature[0][0], towers == 2 ? &nV[1][0] : NULL,
                                  towers == 2 ? &adj[1][0] : NULL, towers == 2 ? &feature[1][0] : NULL),
             "gf_smp_model_prepare");
    }
    double loss_of_batch(int nBatch, double *target) {
        std::vector<double> loss(nBatch);
        must(gf_smp_model_forward_host(net, target, NULL, &loss[0]), "gf_smp_model_forward_host");
        double total = 0.0;
        for (int i = 0; i < nBatch; ++i) total += loss[i];
        return total;
    }
    std::pair<double, double> learn(int nBatch, double *target, double learning_rate, bool with_losses) {
        std::pair<double, double> ret(0.0, 0.0);
        if (with_losses)
            ret.first = loss_of_batch(nBatch, target);  // getLoss (its forward draws dropout masks too, as the reference's does)
        must(gf_smp_model_forward_host(net, target, NULL, NULL), "gf_smp_model_forward_host");
        must(gf_smp_model_backward(net, NULL, NULL, 0), "gf_smp_model_backward");
        must(gf_smp_model_adam_step(net, learning_rate, nBatch), "gf_smp_model_adam_step");
        if (with_losses) ret.second = loss_of_batch(nBatch, target);
        return ret;
    }
    void predict_batch(int nBatch, double *predict) { must(gf_smp_model_forward_host(net, NULL, predict, NULL), "gf_smp_model_forward_host"); }
    void must(gf_status st, const char *what) {
        if (st != GF_OK) gfhost::die(gfhost::default_context(), what, st);
    }
    void die(const char *msg) {
        std::fprintf(stderr, "SMP_model_hip: %s\n", msg);
        std::abort();
    }
    gf_smp_model *net;
    int towers, maxV[2], nF[2];
    std::vector<int> nV[2], adj[2];
    std::vector<double> feature[2];
};

// one tower: SMP_omega_physics (SMP_omega_physics.h:607-925)
class SMP_omega_physics_hip : public SMP_model_hip {
public:
    SMP_omega_physics_hip(int max_nVertices, int max_receptive_field, int nLevels, int nChanels, int nFeatures)
        : SMP_model_hip(1, max_nVertices, 0, max_receptive_field, nLevels, nChanels, nFeatures, 0, 0) {}
    template <class Graph>
    double getLoss(int nBatch, Graph **molecule, double *target) {
        pack(0, nBatch, molecule);
        prepare(nBatch);
        return loss_of_batch(nBatch, target);
    }
    template <class Graph>
    std::pair<double, double> BatchLearn(int nBatch, Graph **molecule, double *target, double learning_rate) {
        pack(0, nBatch, molecule);
        prepare(nBatch);
        return learn(nBatch, target, learning_rate, true);
    }
    template <class Graph>
    void Threaded_BatchLearn(int nBatch, Graph **molecule, double *target, double learning_rate) {
        pack(0, nBatch, molecule);
        prepare(nBatch);
        learn(nBatch, target, learning_rate, false);
    }
    template <class Graph>
    void Threaded_Predict(int nBatch, Graph **molecule, double *predict) {
        pack(0, nBatch, molecule);
        prepare(nBatch);
        predict_batch(nBatch, predict);
    }
    template <class Graph>
    double Predict(Graph *molecule) {
        double y = 0.0;
        Threaded_Predict(1, &molecule, &y);
        return y;
    }
};

class SMP_beta_physics_hip : public SMP_omega_physics_hip {  // no receptive-field cap (SMP_beta_physics.h:31)
public:
    SMP_beta_physics_hip(int max_nVertices, int nLevels, int nChanels, int nFeatures)
        : SMP_omega_physics_hip(max_nVertices, max_nVertices, nLevels, nChanels, nFeatures) {}
};

// two towers: SMP_omega_pairgraphs (SMP_omega_pairgraphs.h:732-1080)
class SMP_omega_pairgraphs_hip : public SMP_model_hip {
public:
    SMP_omega_pairgraphs_hip(int max_nVertices_1, int max_nVertices_2, int max_receptive_field, int nLevels, int nChanels,
                             int nFeatures_1, int nFeatures_2, int nKept = 0)
        : SMP_model_hip(2, max_nVertices_1, max_nVertices_2, max_receptive_field, nLevels, nChanels, nFeatures_1, nFeatures_2, nKept) {}
    template <class Graph>
    double getLoss(int nBatch, Graph **molecule_1, Graph **molecule_2, double *target) {
        bind(nBatch, molecule_1, molecule_2);
        return loss_of_batch(nBatch, target);
    }
    template <class Graph>
    std::pair<double, double> BatchLearn(int nBatch, Graph **molecule_1, Graph **molecule_2, double *target, double learning_rate) {
        bind(nBatch, molecule_1, molecule_2);
        return learn(nBatch, target, learning_rate, true);
    }
    template <class Graph>
    void Threaded_BatchLearn(int nBatch, Graph **molecule_1, Graph **molecule_2, double *target, double learning_rate) {
        bind(nBatch, molecule_1, molecule_2);
        learn(nBatch, target, learning_rate, false);
    }
    template <class Graph>
    void Threaded_Predict(int nBatch, Graph **molecule_1, Graph **molecule_2, double *predict) {
        bind(nBatch, molecule_1, molecule_2);
        predict_batch(nBatch, predict);
    }
    template <class Graph>
    double Predict(Graph *molecule_1, Graph *molecule_2) {
        double y = 0.0;
        Threaded_Predict(1, &molecule_1, &molecule_2, &y);
        return y;
    }

private:
    template <class Graph>
    void bind(int nBatch, Graph **m1, Graph **m2) {
        pack(0, nBatch, m1);
        pack(1, nBatch, m2);
        prepare(nBatch);
    }
};

class SMP_beta_pairgraphs_hip : public SMP_omega_pairgraphs_hip {
public:
    SMP_beta_pairgraphs_hip(int max_nVertices_1, int max_nVertices_2, int nLevels, int nChanels, int nFeatures_1, int nFeatures_2)
        : SMP_omega_pairgraphs_hip(max_nVertices_1, max_nVertices_2, max_nVertices_1 > max_nVertices_2 ? max_nVertices_1 : max_nVertices_2,
                                   nLevels, nChanels, nFeatures_1, nFeatures_2) {}
};

// RisiContraction_18_dropout inside the towers (SMP_sigma_pairgraphs.h:81): nKept of the 18 slices in train mode
class SMP_sigma_pairgraphs_hip : public SMP_omega_pairgraphs_hip {
public:
    SMP_sigma_pairgraphs_hip(int max_nVertices_1, int max_nVertices_2, int max_receptive_field, int nLevels, int nChanels,
                             int nFeatures_1, int nFeatures_2, int nKept)
        : SMP_omega_pairgraphs_hip(max_nVertices_1, max_nVertices_2, max_receptive_field, nLevels, nChanels, nFeatures_1, nFeatures_2, nKept) {}
};

#endif
