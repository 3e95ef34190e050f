// popsift/popsift.h -- PopSift / SiftJob, the public extraction API.
//
// Same public surface as the reference (popsift.h:44-317): SiftJob and PopSift live in the
// global namespace, enqueue() deep-copies the image and returns a heap SiftJob*, get() blocks on
// a single-shot future and returns a heap FeaturesHost* the caller deletes.
//
// Implementation differs (MI355X-first): no HIP/CUDA type appears in this header; a PopSift
// owns POPSIFT_PIPE_DEPTH (default 8) worker threads, each with its own extraction context of the
// C-ABI (include/popsift_hip.h: pyramid + HIP stream).  A worker takes the next job from the queue,
// uploads, queues the whole kernel chain, sleeps until its frame is done and fulfils the job, so
// several frames overlap on the GPU.  The reference runs 2 threads over 3 queues with one pyramid
// and synchronises the whole device four times per image (popsift.cpp:293-344).
#pragma once

#include "common/sync_queue.h"
#include "common/device_prop.h"
#include "sift_conf.h"
#include "sift_config.h"
#include "sift_extremum.h"

#include <cstddef>
#include <exception>
#include <future>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace popsift
{
    class FeaturesBase;
    class FeaturesHost;
    class FeaturesDev;
} // namespace popsift

class SiftJob
{
    std::promise<popsift::FeaturesBase*> _p;
    std::future <popsift::FeaturesBase*> _f;
    int                 _w;
    int                 _h;
    unsigned char*      _imageData;
    size_t              _pinned_cap;   ///< > 0: _imageData is pinned, GPU-mapped pool memory (direct DMA source)
    bool                _is_float;
    std::exception_ptr  _err;

public:
    /// byte image, value range 0..255
    SiftJob( int w, int h, const unsigned char* imageData );
    /// float image, value range [0..1[
    SiftJob( int w, int h, const float* imageData );
    ~SiftJob( );

    /// deprecated alias of getHost()
    popsift::FeaturesHost* get();
    popsift::FeaturesBase* getBase();
    popsift::FeaturesHost* getHost();
    popsift::FeaturesDev*  getDev();

    /// fulfil the promise (internal)
    void setFeatures( popsift::FeaturesBase* f );
    void setError( std::exception_ptr ptr );

    // internal accessors used by the dispatcher
    int  getWidth() const  { return _w; }
    int  getHeight() const { return _h; }
    bool isFloat() const   { return _is_float; }
    const unsigned char* getData() const { return _imageData; }
    bool isPinned() const  { return _pinned_cap != 0; }
};

class PopSift
{
public:
    enum ImageMode
    {
        ByteImages,   ///< byte image, value range 0..255
        FloatImages   ///< float image, value range [0..1[
    };

    enum AllocTest
    {
        Ok,
        ImageExceedsLinearTextureLimit,
        ImageExceedsLayeredSurfaceLimit
    };

public:
    PopSift() = delete;
    PopSift(const PopSift&) = delete;

    explicit PopSift( ImageMode imode = ByteImages, int device = 0 );
    explicit PopSift( const popsift::Config&          config,
                      popsift::Config::ProcessingMode mode = popsift::Config::ExtractingMode,
                      ImageMode imode = ByteImages, int device = 0 );
    ~PopSift();

public:
    /// provide the configuration; refused (returns false) once extraction has started
    bool configure( const popsift::Config& config, bool force = false );

    void uninit( );

    /// always Ok on this backend unless the pyramid cannot be addressed (no texture limits on gfx950)
    AllocTest testTextureFit( int width, int height );
    std::string testTextureFitErrorString( AllocTest err, int w, int h );

    SiftJob* enqueue( int w, int h, const unsigned char* imageData );
    SiftJob* enqueue( int w, int h, const float* imageData );

    /// deprecated
    inline void uninit( int /*pipe*/ ) { uninit(); }
    /// deprecated
    inline bool init( int /*pipe*/, int w, int h ) { _last_init_w = w; _last_init_h = h; return true; }
    /// deprecated
    inline popsift::FeaturesBase* execute( int /*pipe*/, const unsigned char* imageData )
    {
        SiftJob* j = enqueue( _last_init_w, _last_init_h, imageData );
        if( !j ) return nullptr;
        popsift::FeaturesBase* f = j->getBase();
        delete j;
        return f;
    }

private:
    struct Impl;
    void start();
    void dispatchLoop();
    void resolveOctaves( int w, int h );

    std::unique_ptr<Impl> _impl;
    popsift::Config _config;
    popsift::Config::ProcessingMode _proc_mode;
    int             _last_init_w{};
    int             _last_init_h{};
    ImageMode       _image_mode;
    int             _device;
    bool            _isInit{true};
    popsift::cuda::device_prop_t _device_properties;
};
