// popsift.cpp -- PopSift / SiftJob on top of the C-ABI (include/popsift_hip.h).
//
// Reference behaviour restated: popsift.cpp:25-503.  Differences in mechanism (not in contract):
//   * one dispatcher thread drives PIPE_DEPTH extraction contexts (each = pyramid + HIP stream);
//     a job is uploaded and its whole kernel chain queued without any host synchronisation, the
//     dispatcher only blocks on the OLDEST frame in flight, so several frames overlap on the GPU;
//   * results arrive by zero-copy export (psx_attach_export): no per-image pin/unpin
//     (features.cu:86-111), no D2H copy commands;
//   * every failure is caught, stored in the job and re-thrown from get(); a job is always
//     fulfilled (the reference's extract loop has no try/catch and would std::terminate).
#include "popsift/popsift.h"
#include "popsift/features.h"
#include "host_pool.h"

#include "popsift_hip.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <sstream>
#include <unistd.h>

using namespace std;

namespace {

[[noreturn]] void pop_fatal( const char* file, int line, const std::string& msg )
{
    // POP_FATAL (common/debug_macros.h:122-127)
    std::ostringstream o;
    o << file << ":" << line << std::endl << "    " << msg;
    throw std::runtime_error( o.str() );
}
#define POP_FATAL(s) pop_fatal( __FILE__, __LINE__, (s) )

void to_psx( const popsift::Config& c, psx_config& p )
{
    psx_config_default( &p );
    p.octaves             = c.octaves;
    p.levels              = c.levels;
    p.sigma               = c.sigma;
    p.edge_limit          = c._edge_limit;
    p.threshold           = c.getThreshold();
    p.upscale_factor      = c.getUpscaleFactor();
    p.gauss_mode          = (int)c.getGaussMode();
    p.sift_mode           = (int)c.getSiftMode();
    p.scaling_mode        = (int)c.getScalingMode();
    p.desc_mode           = (int)c.getDescMode();
    p.norm_mode           = (int)c.getNormMode();
    p.norm_multi          = c.getNormalizationMultiplier();
    p.max_extrema         = c.getMaxExtrema();
    p.assume_initial_blur = c.hasInitialBlur() ? 1 : 0;
    p.initial_blur        = c.getInitialBlur();
    p.filter_max_extrema  = c.getFilterMaxExtrema();
    p.filter_grid_size    = c.getFilterGridSize();
    p.grid_filter_mode    = (int)c.getFilterSorting();
}

int pipe_depth()
{
    int d = 8;
    if( const char* e = getenv( "POPSIFT_PIPE_DEPTH" ) ) d = atoi( e );
    return d < 1 ? 1 : ( d > 32 ? 32 : d );
}

// export capacities per context: larger results fall back to psx_download
const int EXPORT_FEATURES    = 1 << 16;
const int EXPORT_DESCRIPTORS = 1 << 17;

} // namespace

/*********************************************************************************
 * SiftJob
 *********************************************************************************/

SiftJob::SiftJob( int w, int h, const unsigned char* imageData )
    : _w(w), _h(h), _imageData(nullptr), _is_float(false)
{
    _f = _p.get_future();
    const size_t bytes = (size_t)w * h;
    _imageData = (unsigned char*)malloc( bytes ? bytes : 1 );
    if( _imageData == nullptr )
        POP_FATAL( "Memory limitation\nE    Failed to allocate memory for SiftJob" );
    memcpy( _imageData, imageData, bytes );
}

SiftJob::SiftJob( int w, int h, const float* imageData )
    : _w(w), _h(h), _imageData(nullptr), _is_float(true)
{
    _f = _p.get_future();
    const size_t bytes = (size_t)w * h * sizeof(float);
    _imageData = (unsigned char*)malloc( bytes ? bytes : 1 );
    if( _imageData == nullptr )
        POP_FATAL( "Memory limitation\nE    Failed to allocate memory for SiftJob" );
    memcpy( _imageData, imageData, bytes );
}

SiftJob::~SiftJob( ) { free( _imageData ); }

void SiftJob::setFeatures( popsift::FeaturesBase* f ) { _p.set_value( f ); }
void SiftJob::setError( std::exception_ptr ptr )      { _err = ptr; }

popsift::FeaturesHost* SiftJob::get() { return getHost(); }

popsift::FeaturesBase* SiftJob::getBase()
{
    popsift::FeaturesBase* f = _f.get();
    if( _err != nullptr ) std::rethrow_exception( _err );
    return f;
}

popsift::FeaturesHost* SiftJob::getHost() { return dynamic_cast<popsift::FeaturesHost*>( getBase() ); }
popsift::FeaturesDev*  SiftJob::getDev()  { return dynamic_cast<popsift::FeaturesDev*>( getBase() ); }

/*********************************************************************************
 * PopSift
 *********************************************************************************/

namespace {
// one extraction context of the pipe: pyramid + stream + export targets + the job it works on
struct Slot
{
    psx_ctx*     ctx = nullptr;
    SiftJob*     job = nullptr;        // non-null: reserved by the submit thread / frame in flight
    psx_feature* xfeat = nullptr;      // pinned export window for the 52-byte feature records (per slot)
    size_t       xfeat_cap = 0;        // bytes
    float*       xdesc = nullptr;      // pinned descriptor buffer of the frame in flight (pool; handed to the result)
    size_t       xdesc_cap = 0;        // bytes
    int          desc_cap = 0;         // descriptors
};
} // namespace

struct PopSift::Impl
{
    popsift::SyncQueue<SiftJob*> queue;
    std::unique_ptr<std::thread> worker;      // submit thread (upload + launch chain)
    std::unique_ptr<std::thread> collector;   // result thread (wait, hand over buffers, fulfil jobs in FIFO order)
    std::vector<Slot>            slots;
    std::deque<int>              inflight;    // slot indices, oldest first
    std::mutex                   m;           // guards inflight, Slot::job, submit_done
    std::condition_variable      cv_inflight, cv_free;
    bool                         submit_done = false;
    std::atomic<int>             want_desc{ 32768 };   // descriptor capacity of the next export buffer
    // POPSIFT_PROFILE=1: seconds spent per phase, printed by uninit()
    bool   prof = false;
    double t_slot = 0, t_attach = 0, t_upload = 0, t_submit = 0, t_frame = 0, t_wrap = 0;
    int    n_done = 0;
    std::mutex                   cfg_mutex;
    bool                         contexts_exist = false;
    bool                         stopped = false;
};

PopSift::PopSift( const popsift::Config& config, popsift::Config::ProcessingMode mode, ImageMode imode, int device )
    : _impl( new Impl ), _proc_mode( mode ), _image_mode( imode ), _device( device )
{
    configure( config );
    start();
}

PopSift::PopSift( ImageMode imode, int device )
    : _impl( new Impl ), _proc_mode( popsift::Config::ExtractingMode ), _image_mode( imode ), _device( device )
{
    start();
}

PopSift::~PopSift()
{
    if( _isInit ) uninit();
}

void PopSift::start()
{
    _impl->worker.reset( new std::thread( &PopSift::dispatchLoop, this ) );
}

bool PopSift::configure( const popsift::Config& config, bool /*force*/ )
{
    std::lock_guard<std::mutex> g( _impl->cfg_mutex );
    if( _impl->contexts_exist ) return false;          // popsift.cpp:81-83
    _config = config;
    _config.levels = std::max( 2, config.levels );     // popsift.cpp:86
    return true;
}

void PopSift::uninit( )
{
    if( !_isInit ) {
        std::cerr << "[warning] Attempt to release resources from an uninitialized instance" << std::endl;
        return;
    }
    _impl->queue.push( nullptr );                       // shutdown sentinel (popsift.cpp:486)
    if( _impl->worker ) { _impl->worker->join(); _impl->worker.reset(); }
    _isInit = false;
}

PopSift::AllocTest PopSift::testTextureFit( int width, int height )
{
    // The CUDA texture / layered-surface limits tested by the reference (popsift.cpp:168-196) do not
    // exist here: the pyramid is plain HBM.  Reject only what cannot be addressed.
    if( width <= 0 || height <= 0 ) return AllocTest::ImageExceedsLinearTextureLimit;
    const float scale = 1.0f / powf( 2.0f, -_config.getUpscaleFactor() );
    const double px = ceil( (double)width * scale ) * ceil( (double)height * scale );
    if( px * ( _config.levels + 3 ) * 1.34 > 1.0e11 ) return AllocTest::ImageExceedsLayeredSurfaceLimit;
    return AllocTest::Ok;
}

std::string PopSift::testTextureFitErrorString( AllocTest err, int width, int height )
{
    ostringstream ostr;
    switch( err )
    {
        case AllocTest::Ok :
            ostr << "?    No error." << endl;
            break;
        case AllocTest::ImageExceedsLinearTextureLimit :
            ostr << "E    Cannot load unscaled image. " << endl
                 << "E    Invalid image size (" << width << "," << height << ")" << endl;
            break;
        case AllocTest::ImageExceedsLayeredSurfaceLimit :
            ostr << "E    Cannot use downscaling factor " << -_config.getUpscaleFactor()
                 << " (i.e. upscaling by " << pow( 2, _config.getUpscaleFactor() ) << "). " << endl
                 << "E    The scaled pyramid of a (" << width << "," << height << ") image with "
                 << _config.levels << " levels per octave does not fit into device memory." << endl;
            break;
        default:
            ostr << "E    Programming error, please report." << endl;
            break;
    }
    return ostr.str();
}

SiftJob* PopSift::enqueue( int w, int h, const unsigned char* imageData )
{
    if( _image_mode != ByteImages )
        POP_FATAL( "Image mode error\nE    Cannot load byte images into a PopSift pipeline configured for float images" );

    AllocTest a = testTextureFit( w, h );
    if( a != AllocTest::Ok ) {
        cerr << __FILE__ << ":" << __LINE__ << " Image too large" << endl << testTextureFitErrorString( a, w, h );
        return nullptr;
    }
    SiftJob* job = new SiftJob( w, h, imageData );
    _impl->queue.push( job );
    return job;
}

SiftJob* PopSift::enqueue( int w, int h, const float* imageData )
{
    if( _image_mode != FloatImages )
        POP_FATAL( "Image mode error\nE    Cannot load float images into a PopSift pipeline configured for byte images" );

    AllocTest a = testTextureFit( w, h );
    if( a != AllocTest::Ok ) {
        cerr << __FILE__ << ":" << __LINE__ << " Image too large" << endl << testTextureFitErrorString( a, w, h );
        return nullptr;
    }
    SiftJob* job = new SiftJob( w, h, imageData );
    _impl->queue.push( job );
    return job;
}

namespace {

void check( psx_ctx* ctx, int rc, const char* what )
{
    if( rc == PSX_OK ) return;
    const char* m = psx_last_error( ctx );
    std::string msg = std::string( what ) + " failed";
    if( m && *m ) msg += std::string( ":\n    " ) + m;
    throw std::runtime_error( msg );
}

inline double pnow() { return std::chrono::duration<double>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }

popsift::FeaturesHost* collect_host( Slot& s, std::atomic<int>& want_desc, double* t_frame )
{
    int ne = 0, no = 0;
    const double t0 = pnow();
    check( s.ctx, psx_counts( s.ctx, &ne, &no ), "psx_counts" );       // waits for the frame
    *t_frame += pnow() - t0;
    if( ne == 0 || no == 0 ) {
        if( no == 0 ) cerr << "Warning: no descriptors extracted" << endl;   // sift_desc.cu:88-92
    }
    want_desc = std::max( 32768, no + no / 2 );
    popsift::FeaturesHost* f = new popsift::FeaturesHost();
    size_t ext_cap = 0;
    popsift::Feature* dst = (popsift::Feature*)popsift::pool::get_plain( (size_t)std::max( ne, 1 ) * sizeof(popsift::Feature), &ext_cap );
    if( dst == nullptr ) { delete f; throw std::runtime_error( "out of host memory for features" ); }
    const psx_feature* src = s.xfeat;
    std::vector<psx_feature> tmp;
    popsift::Descriptor* base = nullptr;
    if( ne > EXPORT_FEATURES || no > s.desc_cap ) {
        // result larger than the export targets: ordinary download of the complete device copy
        size_t cap = 0;
        base = (popsift::Descriptor*)popsift::pool::get_pinned( (size_t)std::max( no, 1 ) * sizeof(popsift::Descriptor), &cap );
        if( base == nullptr ) { popsift::pool::put_plain( dst, ext_cap ); delete f; throw std::runtime_error( "out of host memory for descriptors" ); }
        f->adopt( ne, no, dst, ext_cap, base, cap );
        tmp.resize( ne );
        check( s.ctx, psx_download( s.ctx, tmp.data(), ne, (float*)base, no ), "psx_download" );
        src = tmp.data();
    } else {
        // the GPU wrote the descriptors straight into s.xdesc: the result object takes the buffer over
        base = (popsift::Descriptor*)s.xdesc;
        f->adopt( ne, no, dst, ext_cap, base, s.xdesc_cap );
        s.xdesc = nullptr; s.xdesc_cap = 0; s.desc_cap = 0;
    }
    for( int i = 0; i < ne; i++ ) {
        const psx_feature& a = src[i];
        popsift::Feature&  b = dst[i];
        b.debug_octave = a.debug_octave;
        b.xpos = a.xpos; b.ypos = a.ypos; b.sigma = a.sigma;
        b.num_ori = a.num_ori;
        for( int k = 0; k < ORIENTATION_MAX_COUNT; k++ ) {
            b.orientation[k] = a.orientation[k];
            b.desc[k] = ( a.desc_idx[k] >= 0 && a.desc_idx[k] < no ) ? base + a.desc_idx[k] : nullptr;
        }
    }
    return f;
}

popsift::FeaturesDev* collect_dev( Slot& s, int device )
{
    int ne = 0, no = 0;
    check( s.ctx, psx_counts( s.ctx, &ne, &no ), "psx_counts" );
    popsift::FeaturesDev* f = new popsift::FeaturesDev();
    f->setDevice( device );
    f->reset( ne, no );
    check( s.ctx, psx_clone_results( s.ctx, f->getFeatures(), f->getDescriptors(), f->getReverseMap() ),
           "psx_clone_results" );
    return f;
}

} // namespace

// Two host threads per PopSift, like the reference's upload / extract-download pair (popsift.cpp:276-344):
//   dispatchLoop (submit)  pulls jobs, reserves a free context, uploads the image (pinned staging inside
//                          the C-ABI) and queues the whole kernel chain; never waits for a frame
//   collectLoop            waits for the OLDEST frame in flight, wraps the buffers the GPU has written into
//                          a FeaturesHost (no copy of the descriptors) and fulfils the job
void PopSift::collectLoop( )
{
    Impl& p = *_impl;
    for( ;; ) {
        int si;
        {
            std::unique_lock<std::mutex> lk( p.m );
            p.cv_inflight.wait( lk, [&]{ return !p.inflight.empty() || p.submit_done; } );
            if( p.inflight.empty() ) break;
            si = p.inflight.front();
            p.inflight.pop_front();
        }
        Slot& s = p.slots[si];
        SiftJob* job = s.job;
        popsift::FeaturesBase* f = nullptr;
        const double tc0 = pnow();
        double tf = 0;
        try {
            if( _proc_mode == popsift::Config::ExtractingMode ) f = collect_host( s, p.want_desc, &tf );
            else                                                f = collect_dev( s, _device );
        } catch( ... ) {
            job->setError( std::current_exception() );
            f = nullptr;
        }
        p.t_frame += tf; p.t_wrap += pnow() - tc0 - tf; p.n_done++;
        {
            std::lock_guard<std::mutex> g( p.m );
            s.job = nullptr;
        }
        p.cv_free.notify_one();
        job->setFeatures( f );
    }
}

void PopSift::dispatchLoop( )
{
    Impl& p = *_impl;
    const int depth = pipe_depth();
    {
        std::lock_guard<std::mutex> g( p.m );
        p.submit_done = false;
    }
    p.collector.reset( new std::thread( &PopSift::collectLoop, this ) );

    for( ;; ) {
        SiftJob* job = p.queue.pull();
        if( job == nullptr ) break;     // shutdown sentinel
        int si = -1;
        try {
            if( !p.contexts_exist ) {
                std::lock_guard<std::mutex> g( p.cfg_mutex );
                psx_config pc;
                to_psx( _config, pc );
                p.slots.resize( depth );
                for( int i = 0; i < depth; i++ ) {
                    Slot& s = p.slots[i];
                    if( psx_create( _device, &pc, &s.ctx ) != PSX_OK ) {
                        const char* m = psx_last_error( nullptr );
                        throw std::runtime_error( std::string( "psx_create failed:\n    " ) + ( m ? m : "" ) );
                    }
                    if( _proc_mode == popsift::Config::ExtractingMode ) {
                        s.xfeat = (psx_feature*)popsift::pool::get_pinned( (size_t)EXPORT_FEATURES * sizeof(psx_feature), &s.xfeat_cap );
                        if( s.xfeat == nullptr ) throw std::runtime_error( "out of host memory for export buffers" );
                    }
                }
                p.contexts_exist = true;
            }
            const double ts0 = pnow();
            {
                // a free context: one without a job
                std::unique_lock<std::mutex> lk( p.m );
                p.cv_free.wait( lk, [&]{ for( int i = 0; i < depth; i++ ) if( p.slots[i].job == nullptr ) return true; return false; } );
                for( int i = 0; i < depth; i++ ) if( p.slots[i].job == nullptr ) { si = i; break; }
                p.slots[si].job = job;
            }
            Slot& s = p.slots[si];
            const double ts1 = pnow();
            if( _proc_mode == popsift::Config::ExtractingMode && s.xdesc == nullptr ) {
                // the previous result took this context's descriptor buffer with it: attach a fresh one
                const int want = p.want_desc;
                s.xdesc = (float*)popsift::pool::get_pinned( (size_t)want * sizeof(popsift::Descriptor), &s.xdesc_cap );
                if( s.xdesc == nullptr ) throw std::runtime_error( "out of host memory for export buffers" );
                s.desc_cap = (int)std::min<size_t>( s.xdesc_cap / sizeof(popsift::Descriptor), (size_t)1 << 30 );
                check( s.ctx, psx_attach_export( s.ctx, s.xfeat, EXPORT_FEATURES, s.xdesc, s.desc_cap ), "psx_attach_export" );
            }
            const double ts2 = pnow();
            if( job->isFloat() )
                check( s.ctx, psx_upload_f32( s.ctx, (const float*)job->getData(), job->getWidth(), job->getHeight() ), "psx_upload_f32" );
            else
                check( s.ctx, psx_upload_u8( s.ctx, job->getData(), job->getWidth(), job->getHeight() ), "psx_upload_u8" );
            const double ts3 = pnow();
            check( s.ctx, psx_extract( s.ctx ), "psx_extract" );
            const double ts4 = pnow();
            p.t_slot += ts1 - ts0; p.t_attach += ts2 - ts1; p.t_upload += ts3 - ts2; p.t_submit += ts4 - ts3;
            {
                std::lock_guard<std::mutex> g( p.m );
                p.inflight.push_back( si );
            }
            p.cv_inflight.notify_one();
        } catch( ... ) {
            if( si >= 0 ) {
                { std::lock_guard<std::mutex> g( p.m ); p.slots[si].job = nullptr; }
                p.cv_free.notify_one();
            }
            job->setError( std::current_exception() );
            job->setFeatures( nullptr );
        }
    }

    {
        std::lock_guard<std::mutex> g( p.m );
        p.submit_done = true;
    }
    p.cv_inflight.notify_all();
    if( p.collector ) { p.collector->join(); p.collector.reset(); }
    if( getenv( "POPSIFT_PROFILE" ) != nullptr && p.n_done > 0 ) {
        const double k = 1e3 / p.n_done;
        fprintf( stderr, "[popsift profile] %d frames, ms per frame: submit thread: wait-for-context %.3f attach %.3f upload %.3f launch %.3f | "
                         "collect thread: wait-for-frame %.3f wrap %.3f\n",
                 p.n_done, p.t_slot * k, p.t_attach * k, p.t_upload * k, p.t_submit * k, p.t_frame * k, p.t_wrap * k );
    }

    // jobs enqueued after the sentinel are never processed; release the contexts
    for( auto& s : p.slots ) {
        if( s.ctx ) { psx_attach_export( s.ctx, nullptr, 0, nullptr, 0 ); psx_destroy( s.ctx ); }
        popsift::pool::put_pinned( s.xfeat, s.xfeat_cap );
        popsift::pool::put_pinned( s.xdesc, s.xdesc_cap );
        s = Slot();
    }
    p.contexts_exist = false;
}
