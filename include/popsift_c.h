/*
 * include/popsift_c.h -- flat C binding of the C++ extraction API (PopSift / SiftJob / FeaturesHost).
 *
 * The reference exposes only C++ classes (popsift.h:44-317).  Hosts that cannot include C++ headers
 * (ctypes, cgo, JNI ...) bind these functions instead; bench.py drives the product's real end-to-end
 * path (host image in -> PopSift::enqueue -> SiftJob::get -> FeaturesHost out) through them.  Each
 * function is a one-line forward to the class method it names; exceptions become a NULL / negative
 * return plus popsift_c_last_error().
 */
#ifndef POPSIFT_C_H
#define POPSIFT_C_H

#include "popsift_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct popsift_c_handle   popsift_c_handle;     /* PopSift            */
typedef struct popsift_c_job      popsift_c_job;        /* SiftJob            */
typedef struct popsift_c_features popsift_c_features;   /* popsift::FeaturesHost */

/* PopSift::PopSift(config, mode, imode, device), popsift.h:166-168.  cfg is translated into a
 * popsift::Config through its public setters.  image_mode: 0 ByteImages, 1 FloatImages;
 * processing_mode: 0 ExtractingMode (MatchingMode results are device resident and have no C view). */
popsift_c_handle* popsift_c_create(const psx_config* cfg, int image_mode, int device);
/* PopSift::uninit + destructor */
void popsift_c_destroy(popsift_c_handle* h);
/* PopSift::enqueue, popsift.h:219-232: deep-copies the image; NULL when the image is refused */
popsift_c_job* popsift_c_enqueue_u8(popsift_c_handle* h, int w, int hgt, const unsigned char* img);
popsift_c_job* popsift_c_enqueue_f32(popsift_c_handle* h, int w, int hgt, const float* img);
/* SiftJob::get + delete job, popsift.h:75-81: blocks until the frame is done.  NULL on error. */
popsift_c_features* popsift_c_get(popsift_c_job* job);
/* FeaturesHost::getFeatureCount / getDescriptorCount */
int popsift_c_feature_count(const popsift_c_features* f);
int popsift_c_descriptor_count(const popsift_c_features* f);
/* Copies the keypoints as psx_feature records (Descriptor* turned into indices into the descriptor
 * array) and the descriptors (128 floats each); either pointer may be NULL. */
int popsift_c_copy(const popsift_c_features* f, psx_feature* features, float* descriptors);
/* direct view of the FeaturesHost descriptor array (getDescriptors()), valid until popsift_c_free */
const float* popsift_c_descriptors(const popsift_c_features* f);
/* delete the FeaturesHost */
void popsift_c_free(popsift_c_features* f);
/* message of the last failure on the calling thread */
const char* popsift_c_last_error(void);
/* Diagnostics (no reference counterpart): counters of the library's pinned result / job-image pool of one device
 * (device < 0: all pools summed): out[0] = buffers ever allocated (hipHostMalloc), out[1] = buffers really freed
 * (hipHostFree), out[2] = requests served from the free list, out[3] = buffers on the free list, out[4] = bytes on the
 * free list, out[5] = bytes handed out.  A steady stream of frames must not move out[0] / out[1] after warm-up. */
void popsift_c_pool_stats(int device, long long out[6]);

#ifdef __cplusplus
}
#endif
#endif
