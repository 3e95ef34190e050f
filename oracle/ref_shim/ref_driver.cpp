// oracle/ref_shim/ref_driver.cpp -- C entry points that run the REFERENCE's own code (compiled for
// the CPU through ref_shim/cuda_runtime.h) and expose its intermediate and final results.
// TEST INFRASTRUCTURE ONLY: this is what oracle/sift_oracle.c is pinned against.
//
//   ref_run      drives popsift::Pyramid exactly like PopSift::extractDownloadLoop does
//                (popsift.cpp:306-344: applyConfiguration, private_init, step1, step2,
//                get_descriptors) in the calling thread, so planes and initial extrema can be read;
//   ref_run_api  goes through the public API (PopSift::enqueue / SiftJob::get), worker threads and all.
#include "popsift.h"
#include "features.h"
#include "gauss_filter.h"
#include "s_image.h"
#include "sift_constants.h"
#include "sift_pyramid.h"

#include "../sift_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstring>
#include <vector>

using namespace popsift;

namespace {

struct RefResult {
    int num_octaves = 0, L = 0;
    std::vector<int> W, H;
    std::vector<std::vector<float>> gauss, dog;      // [o*L + l]
    std::vector<std::vector<osift_iext>> iext;
    std::vector<osift_feature> feat;
    std::vector<float> desc;
    int ext_total = 0, ori_total = 0;
};

Config make_config(const osift_config* c)
{
    Config conf;
    conf.setOctaves(c->octaves);
    conf.setLevels(c->levels);
    conf.setSigma(c->sigma);
    conf.setEdgeLimit(c->edge_limit);
    conf.setThreshold(c->threshold);
    conf.setDownsampling(-c->upscale_factor);
    conf.setGaussMode((Config::GaussMode)c->gauss_mode);
    conf.setMode((Config::SiftMode)c->sift_mode);
    conf.setNormMode((Config::NormMode)c->norm_mode);
    conf.setNormalizationMultiplier(c->norm_multi);
    conf.setInitialBlur(c->assume_initial_blur ? c->initial_blur : 0.0f);
    conf.setFilterMaxExtrema(c->filter_max_extrema);
    conf.setFilterGridSize(c->filter_grid_size);
    conf.setFilterSorting((Config::GridFilterMode)c->grid_filter_mode);
    conf.setScalingMode((Config::ScalingMode)c->scaling_mode);
    conf.setDescMode((Config::DescMode)c->desc_mode);
    return conf;
}

void collect_features(FeaturesHost* fh, RefResult* r)
{
    r->ext_total = fh->getFeatureCount();
    r->ori_total = fh->getDescriptorCount();
    r->feat.resize(r->ext_total);
    r->desc.resize((size_t)r->ori_total * 128);
    if (r->ori_total) memcpy(r->desc.data(), fh->getDescriptors(), (size_t)r->ori_total * 128 * sizeof(float));
    const Feature* f = fh->getFeatures();
    for (int i = 0; i < r->ext_total; i++) {
        osift_feature& o = r->feat[i];
        o.debug_octave = f[i].debug_octave;
        o.xpos = f[i].xpos; o.ypos = f[i].ypos; o.sigma = f[i].sigma; o.num_ori = f[i].num_ori;
        for (int k = 0; k < OSIFT_ORI_MAX; k++) {
            o.orientation[k] = f[i].orientation[k];
            o.desc_idx[k] = f[i].desc[k] ? (int)(f[i].desc[k] - fh->getDescriptors()) : -1;
        }
    }
}

const float* array_of(cudaTextureObject_t t) { return reinterpret_cast<const shim::TexObj*>(t)->arr->data; }

} // namespace

extern "C" {

void* ref_run(const osift_config* c, const void* img, int w, int h, int is_float)
{
    RefResult* r = new RefResult;
    Config conf = make_config(c);
    conf.levels = std::max(2, conf.levels);                          // PopSift::configure, popsift.cpp:86

    // PopSift::applyConfiguration, popsift.cpp:91-107
    init_filter(conf, conf.sigma, conf.levels);
    init_constants(conf.sigma, conf.levels, conf.getPeakThreshold(), conf._edge_limit,
                   conf.getMaxExtrema(), conf.getNormalizationMultiplier());

    // PopSift::private_apply_scale_factor, popsift.cpp:109-126
    int pw = w, ph = h;
    {
        float upscaleFactor = conf.getUpscaleFactor();
        float scaleFactor = 1.0f / powf(2.0f, -upscaleFactor);
        if (conf.octaves < 0) {
            int oct = std::max(int(floor(logf((float)std::min(pw, ph)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
            conf.octaves = oct;
        }
        pw = ceilf(pw * scaleFactor);
        ph = ceilf(ph * scaleFactor);
    }

    ImageBase* image = is_float ? (ImageBase*)new ImageFloat : (ImageBase*)new Image;
    image->resetDimensions(w, h);                                    // SiftJob::setImg, popsift.cpp:432-437
    image->load(const_cast<void*>(img));

    Pyramid* pyr = new Pyramid(conf, pw, ph);                        // private_init, popsift.cpp:139
    pyr->step1(conf, image);
    pyr->step2(conf);

    r->num_octaves = pyr->getNumOctaves();
    r->L = pyr->getNumLevels();
    for (int o = 0; o < r->num_octaves; o++) {
        Octave& oc = pyr->getOctave(o);
        const int W = oc.getWidth(), H = oc.getHeight();
        r->W.push_back(W); r->H.push_back(H);
        const float* gd = array_of(oc.getDataTexPoint());
        const float* dd = array_of(oc.getDogTexturePoint());
        for (int l = 0; l < r->L; l++)
            r->gauss.emplace_back(gd + (size_t)l * W * H, gd + (size_t)(l + 1) * W * H);
        for (int l = 0; l < r->L - 1; l++)
            r->dog.emplace_back(dd + (size_t)l * W * H, dd + (size_t)(l + 1) * W * H);
        // initial extrema of this octave that reach the orientation stage (hct / dobuf_shadow:
        // sift_pyramid.cu:41-49), read through i_ext_off as ori_par does (s_orientation.cu:83): after
        // extrema_filter_grid hct.ext_ct[o] counts the survivors and i_ext_off lists them
        std::vector<osift_iext> v;
        const int n = std::min(hct.ext_ct[o], h_consts.max_extrema);
        for (int i = 0; i < n; i++) {
            const InitialExtremum& e = dobuf_shadow.i_ext_dat[o][dobuf_shadow.i_ext_off[o][i]];
            osift_iext x;
            x.xpos = e.xpos; x.ypos = e.ypos; x.lpos = e.lpos; x.sigma = e.sigma; x.cell = e.cell; x.ignore = e.ignore;
            v.push_back(x);
        }
        r->iext.push_back(v);
    }

    FeaturesHost* fh = pyr->get_descriptors(conf);
    collect_features(fh, r);
    delete fh;
    delete pyr;
    delete image;
    return r;
}

void* ref_run_api(const osift_config* c, const void* img, int w, int h, int is_float)
{
    RefResult* r = new RefResult;
    Config conf = make_config(c);
    {
        PopSift ps(conf, Config::ExtractingMode, is_float ? PopSift::FloatImages : PopSift::ByteImages);
        SiftJob* job = is_float ? ps.enqueue(w, h, (const float*)img) : ps.enqueue(w, h, (const unsigned char*)img);
        FeaturesHost* fh = job->get();
        collect_features(fh, r);
        delete fh;
        delete job;
        ps.uninit();
    }
    return r;
}

int ref_gauss_tables(const osift_config* c, osift_tables* t)
{
    Config conf = make_config(c);
    conf.levels = std::max(2, conf.levels);
    memset((void*)&h_gauss, 0, sizeof(h_gauss));      // the tables are process-global: rows of an earlier, deeper configuration would linger
    try { init_filter(conf, conf.sigma, conf.levels); } catch (const std::exception&) { return -1; }
    memset(t, 0, sizeof(*t));
    memcpy(t->inc_filter, h_gauss.inc.filter, sizeof(t->inc_filter));
    memcpy(t->inc_sigma, h_gauss.inc.sigma, sizeof(t->inc_sigma));
    memcpy(t->inc_span, h_gauss.inc.span, sizeof(t->inc_span));
    memcpy(t->dd_filter, h_gauss.dd.filter, sizeof(t->dd_filter));
    memcpy(t->dd_sigma, h_gauss.dd.sigma, sizeof(t->dd_sigma));
    memcpy(t->dd_span, h_gauss.dd.span, sizeof(t->dd_span));
    memcpy(t->abs0_filter, h_gauss.abs_o0.filter, sizeof(t->abs0_filter));
    memcpy(t->abs0_sigma, h_gauss.abs_o0.sigma, sizeof(t->abs0_sigma));
    memcpy(t->abs0_span, h_gauss.abs_o0.span, sizeof(t->abs0_span));
    memcpy(t->absN_filter, h_gauss.abs_oN.filter, sizeof(t->absN_filter));
    memcpy(t->absN_sigma, h_gauss.abs_oN.sigma, sizeof(t->absN_sigma));
    memcpy(t->absN_span, h_gauss.abs_oN.span, sizeof(t->absN_span));
    memcpy(t->inc_ifilter, h_gauss.inc.i_filter, sizeof(t->inc_ifilter));
    memcpy(t->inc_ispan, h_gauss.inc.i_span, sizeof(t->inc_ispan));
    return 0;
}

float ref_peak_threshold(const osift_config* c) { return make_config(c).getPeakThreshold(); }

void ref_free(void* h) { delete static_cast<RefResult*>(h); }
int ref_num_octaves(void* h) { return static_cast<RefResult*>(h)->num_octaves; }
int ref_num_levels(void* h) { return static_cast<RefResult*>(h)->L; }
int ref_octave_width(void* h, int o) { return static_cast<RefResult*>(h)->W[o]; }
int ref_octave_height(void* h, int o) { return static_cast<RefResult*>(h)->H[o]; }
const float* ref_gauss_plane(void* h, int o, int l) { RefResult* r = static_cast<RefResult*>(h); return r->gauss[(size_t)o * r->L + l].data(); }
const float* ref_dog_plane(void* h, int o, int l) { RefResult* r = static_cast<RefResult*>(h); return r->dog[(size_t)o * (r->L - 1) + l].data(); }
int ref_iext_count(void* h, int o) { return (int)static_cast<RefResult*>(h)->iext[o].size(); }
const osift_iext* ref_get_iext(void* h, int o) { return static_cast<RefResult*>(h)->iext[o].data(); }
int ref_ext_total(void* h) { return static_cast<RefResult*>(h)->ext_total; }
int ref_ori_total(void* h) { return static_cast<RefResult*>(h)->ori_total; }
const osift_feature* ref_features(void* h) { return static_cast<RefResult*>(h)->feat.data(); }
const float* ref_descriptors(void* h) { return static_cast<RefResult*>(h)->desc.data(); }


// FeaturesDev::match of the reference (features.cu:270-304).  It only PRINTS its result (device printf in
// show_distance), so stdout is redirected into a temporary file around the call and the lines are parsed:
//   "<accept|reject> feat %4d [%4d] matches feat %4d [%4d] ( 2nd feat %4d [%4d] ) dist %.3f vs %.3f"
// out3[3i..] = {best, second, accept}; dist2[2i..] = the two printed distances (3 decimals).
int ref_match(const float* l, int nl, const float* r, int nr, int* out3, float* dist2)
{
    FeaturesDev lf(nl, nl), rf(nr, nr);
    cudaMemcpy(lf.getDescriptors(), l, (size_t)nl * 128 * sizeof(float), cudaMemcpyHostToDevice);
    cudaMemcpy(rf.getDescriptors(), r, (size_t)nr * 128 * sizeof(float), cudaMemcpyHostToDevice);
    std::vector<int> idl(nl), idr(nr);
    for (int i = 0; i < nl; i++) idl[i] = i;
    for (int i = 0; i < nr; i++) idr[i] = i;
    cudaMemcpy(lf.getReverseMap(), idl.data(), (size_t)nl * sizeof(int), cudaMemcpyHostToDevice);
    cudaMemcpy(rf.getReverseMap(), idr.data(), (size_t)nr * sizeof(int), cudaMemcpyHostToDevice);

    char path[] = "/tmp/ref_match_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0) return -1;
    fflush(stdout);
    const int saved = dup(1);
    dup2(fd, 1);
    lf.match(&rf);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    close(fd);

    FILE* f = fopen(path, "r");
    if (!f) return -2;
    char line[512];
    int n = 0;
    while (fgets(line, sizeof(line), f)) {
        char verdict[16]; int lfeat, li, rfeat1, m1, rfeat2, m2; float d1, d2;
        if (sscanf(line, "%15s feat %d [%d] matches feat %d [%d] ( 2nd feat %d [%d] ) dist %f vs %f",
                   verdict, &lfeat, &li, &rfeat1, &m1, &rfeat2, &m2, &d1, &d2) == 9 && li >= 0 && li < nl) {
            out3[3 * li + 0] = m1; out3[3 * li + 1] = m2; out3[3 * li + 2] = (strcmp(verdict, "accept") == 0);
            if (dist2) { dist2[2 * li + 0] = d1; dist2[2 * li + 1] = d2; }
            n++;
        }
    }
    fclose(f);
    remove(path);
    return n;
}

} // extern "C"
