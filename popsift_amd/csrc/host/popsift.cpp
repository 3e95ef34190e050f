// popsift.cpp -- PopSift / SiftJob on top of the C-ABI (include/popsift_hip.h).
//
// Reference behaviour restated: popsift.cpp:25-503.  Differences in mechanism (not in contract):
//   * POPSIFT_PIPE_DEPTH (default 8) worker threads, one extraction context (pyramid + HIP stream)
//     each: a worker uploads its job, queues the whole kernel chain without any host synchronisation
//     and sleeps on a blocking event until that frame is done, so several frames overlap on the GPU;
//   * the automatic octave count (Config::octaves < 0) is resolved ONCE per PopSift from the first image
//     enqueued, as the reference does (popsift.cpp:118-122), and handed to every context explicitly;
//   * the image is deep-copied into pinned pool memory at enqueue (one host copy, DMA straight from it);
//     results are DMA-ed into pinned pool buffers that become the FeaturesHost's arrays: no per-image
//     pin/unpin (features.cu:86-111) and no host copy of the descriptors;
//   * every failure is caught, stored in the job and re-thrown from get(); a job is always
//     fulfilled (the reference's extract loop has no try/catch and would std::terminate).
#include "popsift/popsift.h"
#include "popsift/features.h"
#include "host_pool.h"
#include "log_dump.h"
#include "trace.h"

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
#include <sched.h>
#include <cctype>
#include <cstdio>
#include <ctime>

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

// Worker threads (= extraction contexts) per PopSift.  Default: 8, but not more than this process' share of the
// host cores -- with one replica per GPU (torchrun sets LOCAL_WORLD_SIZE; POPSIFT_LOCAL_REPLICAS for other launchers)
// 8 x 8 workers plus the callers would oversubscribe a small host and the result hand-off, not the GPU, would bend the
// 1 -> 8 curve.  POPSIFT_PIPE_DEPTH overrides.
int pipe_depth()
{
    if( const char* e = getenv( "POPSIFT_PIPE_DEPTH" ) ) { const int d = atoi( e ); return d < 1 ? 1 : ( d > 32 ? 32 : d ); }
    int cores = 0;
    cpu_set_t set;
    if( sched_getaffinity( 0, sizeof(set), &set ) == 0 ) cores = CPU_COUNT( &set );
    if( cores <= 0 ) cores = (int)std::thread::hardware_concurrency();
    int replicas = 1;
    if( const char* e = getenv( "POPSIFT_LOCAL_REPLICAS" ) ) replicas = atoi( e );
    else if( const char* e2 = getenv( "LOCAL_WORLD_SIZE" ) ) replicas = atoi( e2 );
    if( replicas < 1 ) replicas = 1;
    const int share = cores > 0 ? cores / replicas : 8;
    // measured on MI355X (1080p bench frames, end to end): 2 workers 4356, 3: 5565, 4: 5887, 6: 5907, 8: 6105, 12: 6048,
    // 16: 6035 Mpix/s -- four contexts in flight already fill the GPU, eight is the plateau.  Since round 4 a worker
    // SLEEPS while its frame runs (wait_stream: ~0.03 ms of CPU per frame and worker), so eight workers fit a share of two
    // cores; only a smaller share falls back to four.
    return share >= 2 ? 8 : 4;
}

// pinned bytes that jobs and result objects may hold before they fall back to pageable memory (POPSIFT_PINNED_LIMIT_MB)
size_t pinned_limit()
{
    static const size_t lim = []{ const char* e = getenv( "POPSIFT_PINNED_LIMIT_MB" ); return (size_t)( e ? atol( e ) : 2048 ) << 20; }();
    return lim;
}

// export capacities per context: larger results fall back to psx_download
const int EXPORT_FEATURES    = 1 << 16;
const int EXPORT_DESCRIPTORS = 1 << 17;

} // namespace

/*********************************************************************************
 * SiftJob
 *********************************************************************************/

namespace {
// The deep copy of the caller's image (popsift.cpp:392-395 in the reference: malloc + memcpy) goes straight into
// pinned, GPU-mapped pool memory: the worker then DMAs from it without a second host copy through a staging buffer.
unsigned char* job_image( const void* src, size_t bytes, size_t* pinned_cap )
{
    *pinned_cap = 0;
    unsigned char* p = nullptr;
    if( popsift::pool::pinned_in_use() <= pinned_limit() ) p = (unsigned char*)popsift::pool::get_pinned( bytes ? bytes : 1, pinned_cap );
    if( p == nullptr ) { *pinned_cap = 0; p = (unsigned char*)malloc( bytes ? bytes : 1 ); }
    if( p == nullptr ) POP_FATAL( "Memory limitation\nE    Failed to allocate memory for SiftJob" );
    memcpy( p, src, bytes );
    return p;
}
} // namespace

SiftJob::SiftJob( int w, int h, const unsigned char* imageData )
    : _w(w), _h(h), _imageData(nullptr), _pinned_cap(0), _is_float(false)
{
    _f = _p.get_future();
    _imageData = job_image( imageData, (size_t)w * h, &_pinned_cap );
}

SiftJob::SiftJob( int w, int h, const float* imageData )
    : _w(w), _h(h), _imageData(nullptr), _pinned_cap(0), _is_float(true)
{
    _f = _p.get_future();
    _imageData = job_image( imageData, (size_t)w * h * sizeof(float), &_pinned_cap );
}

SiftJob::~SiftJob( )
{
    if( _pinned_cap ) popsift::pool::put_pinned( _imageData, _pinned_cap ); else free( _imageData );
}

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
// one extraction context of the pipe: pyramid + stream + export targets, owned by one worker thread
struct Slot
{
    psx_ctx*     ctx = nullptr;
    psx_feature* xfeat = nullptr;      // pinned export window for the 52-byte feature records
    size_t       xfeat_cap = 0;        // bytes
    float*       xdesc = nullptr;      // pinned descriptor buffer of the frame in flight (pool; handed to the result)
    size_t       xdesc_cap = 0;        // bytes
    int          desc_cap = 0;         // descriptors
};
} // namespace

struct PopSift::Impl
{
    popsift::SyncQueue<SiftJob*>             queue;
    std::vector<std::unique_ptr<std::thread>> workers;   // one per context
    std::atomic<int>             want_desc{ 32768 };     // descriptor capacity of the next export buffer
    std::mutex                   cfg_mutex;
    std::atomic<bool>            contexts_exist{ false };
    int                          octaves_resolved = -1;  // sticky automatic octave count (popsift.cpp:118-122); cfg_mutex
    // POPSIFT_PROFILE=1: seconds spent per phase (all workers), printed by uninit()
    std::mutex                   prof_mutex;
    std::mutex                   log_mutex;              // LogMode::All dumps share fixed file names: one frame at a time
    double t_attach = 0, t_upload = 0, t_submit = 0, t_frame = 0, t_wrap = 0, t_pool = 0;
    double t_cpu = 0;                                    // CPU seconds the worker threads burnt (CLOCK_THREAD_CPUTIME_ID)
    int    n_done = 0;
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
    // One worker thread per context (POPSIFT_PIPE_DEPTH, default 8).  A worker takes the next job, uploads,
    // queues the whole kernel chain, sleeps until ITS frame is done and fulfils the job: a host-side wait
    // inside the chain (the grid filter reads the extrema counts, as in the reference) stalls one worker,
    // not the pipe -- the other contexts keep the GPU busy.
    const int depth = pipe_depth();
    for( int i = 0; i < depth; i++ )
        _impl->workers.emplace_back( new std::thread( &PopSift::dispatchLoop, this ) );
}

bool PopSift::configure( const popsift::Config& config, bool /*force*/ )
{
    std::lock_guard<std::mutex> g( _impl->cfg_mutex );
    if( _impl->contexts_exist ) return false;          // popsift.cpp:81-83
    _config = config;
    _config.levels = std::max( 2, config.levels );     // popsift.cpp:86
    _impl->octaves_resolved = -1;
    if( _config.ifPrintGaussTables() ) {                // init_filter's debug output (gauss_filter.cu:146-161, 247-256)
        psx_config pc;
        to_psx( _config, pc );
        psx_print_gauss_tables( &pc, 10 );
    }
    return true;
}

// PopSift::private_apply_scale_factor (popsift.cpp:109-126): with Config::octaves < 0 the octave count is
// derived from the FIRST image this PopSift sees and then stays (the reference writes it into _config).
void PopSift::resolveOctaves( int w, int h )
{
    std::lock_guard<std::mutex> g( _impl->cfg_mutex );
    if( _config.octaves >= 0 || _impl->octaves_resolved >= 0 ) return;
    const float scaleFactor = 1.0f / powf( 2.0f, -_config.getUpscaleFactor() );
    _impl->octaves_resolved = std::max( int( floor( logf( (float)std::min( w, h ) ) / logf( 2.0f ) ) - 3.0f + scaleFactor ), 1 );
}

void PopSift::uninit( )
{
    if( !_isInit ) {
        std::cerr << "[warning] Attempt to release resources from an uninitialized instance" << std::endl;
        return;
    }
    _impl->queue.push( nullptr );                       // shutdown sentinel (popsift.cpp:486), passed on by each worker
    for( auto& w : _impl->workers ) if( w ) w->join();
    _impl->workers.clear();
    {
        SiftJob* j = nullptr;                           // swallow the sentinel the last worker passed on
        while( _impl->queue.try_pull( j ) ) { }
    }
    if( getenv( "POPSIFT_PROFILE" ) != nullptr && _impl->n_done > 0 ) {
        const double k = 1e3 / _impl->n_done;
        const popsift::pool::Stats ps = popsift::pool::pinned_stats( _device );
        fprintf( stderr, "[popsift profile] device %d: %d frames, host ms per frame (per worker thread): attach %.3f (pool %.3f) upload %.3f launch %.3f "
                         "wait-for-frame %.3f wrap %.3f; host CPU ms per frame (all workers) %.3f; pinned pool: %ld allocations, %ld frees, %ld hits, "
                         "%ld buffers / %.0f MB free\n", _device, _impl->n_done, _impl->t_attach * k, _impl->t_pool * k, _impl->t_upload * k,
                 _impl->t_submit * k, _impl->t_frame * k, _impl->t_wrap * k, _impl->t_cpu * k,
                 ps.allocs, ps.frees, ps.hits, ps.free_buffers, ps.free_bytes / 1048576.0 );
    }
    _impl->contexts_exist = false;
    _isInit = false;
    // the calling thread's matcher scratch (psx_match) is NOT released here: it belongs to the thread, another PopSift
    // on the same thread may be using it, and releasing it would switch the caller's current device;
    // it goes at thread exit or on an explicit psx_match_release()
}

PopSift::AllocTest PopSift::testTextureFit( int width, int height )
{
    // The CUDA texture / layered-surface limits tested by the reference (popsift.cpp:168-196) do not
    // exist here: the pyramid is plain HBM.  Reject only what cannot be addressed.
    if( width <= 0 || height <= 0 ) return AllocTest::ImageExceedsLinearTextureLimit;
    const float scale = 1.0f / powf( 2.0f, -_config.getUpscaleFactor() );
    const double px = ceil( (double)width * scale ) * ceil( (double)height * scale );
    if( px * ( _config.levels + 3 ) * 1.34 > 1.0e11 ) return AllocTest::ImageExceedsLayeredSurfaceLimit;
    if( px * 4.0 >= 4.0e9 ) return AllocTest::ImageExceedsLayeredSurfaceLimit;    // 32-bit offsets inside a plane
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
    resolveOctaves( w, h );
    popsift::pool::DeviceScope pool_of( _device );      // the job image comes from THIS device's pinned pool
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
    resolveOctaves( w, h );
    popsift::pool::DeviceScope pool_of( _device );
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
    // the frame was launched with export targets attached (POPSIFT_EXPORT=1) and fits them: the kernels already
    // stored records and descriptors into s.xfeat / s.xdesc.  Everything else is downloaded from the device copy --
    // also a frame with keypoints and no descriptors: s.xfeat holds nothing the GPU wrote unless an export was attached.
    const bool exported = s.xdesc != nullptr && s.desc_cap > 0 && ne <= EXPORT_FEATURES && no <= s.desc_cap;
    const bool hoarding = popsift::pool::pinned_in_use() > pinned_limit();
    if( !exported && s.xdesc != nullptr && no > s.desc_cap ) {
        // export was attached but this frame did not fit: without this the too-small buffer would stay attached and
        // every later large frame would pay the PCIe stores AND a download.  Hand it back; the next
        // frame attaches a want_desc-sized one.
        popsift::pool::put_pinned( s.xdesc, s.xdesc_cap );
        s.xdesc = nullptr; s.xdesc_cap = 0; s.desc_cap = 0;
    }
    if( !exported ) {
        psx_feature* ftarget = s.xfeat;                       // pinned: the D2H copy stays asynchronous
        if( ftarget == nullptr || ne > EXPORT_FEATURES ) { tmp.resize( std::max( ne, 1 ) ); ftarget = tmp.data(); }
        if( hoarding ) {
            // a caller that hoards results must not exhaust pinned memory: beyond the limit the result gets ordinary
            // page-aligned arrays (what the reference hands out, features.cu:58-84) and the download goes there
            popsift::pool::put_plain( dst, ext_cap );
            try { f->reset( ne, no ); } catch( ... ) { delete f; throw; }
            dst = f->getFeatures();
            base = f->getDescriptors();
        } else {
            size_t cap = 0;
            base = (popsift::Descriptor*)popsift::pool::get_pinned( (size_t)std::max( no, 1 ) * sizeof(popsift::Descriptor), &cap );
            if( base == nullptr ) { popsift::pool::put_plain( dst, ext_cap ); delete f; throw std::runtime_error( "out of host memory for descriptors" ); }
            f->adopt( ne, no, dst, ext_cap, base, cap );
        }
        try {
            check( s.ctx, psx_download( s.ctx, ftarget, ne, (float*)base, no ), "psx_download" );
        } catch( ... ) { delete f; throw; }
        src = ftarget;
    } else if( hoarding ) {
        popsift::pool::put_plain( dst, ext_cap );
        try { f->reset( ne, no ); } catch( ... ) { delete f; throw; }
        memcpy( f->getDescriptors(), s.xdesc, (size_t)no * sizeof(popsift::Descriptor) );
        dst = f->getFeatures();
        base = f->getDescriptors();
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

namespace {
// Best effort: run this worker on the CPUs local to the GPU's PCIe root (the upload staging copy and the
// result hand-off then stay on the GPU's NUMA node; it matters with 8 GPUs x 8 workers on a two-socket host).
// POPSIFT_NUMA_PIN=0 disables it.  Linux sysfs only; any failure leaves the affinity untouched.
void pin_to_device_numa_node( int device )
{
    (void)popsift::pool::pin_thread_to_device_cpus( device );      // one reentrant parser, one sysfs read per device (host_pool.h)
}
} // namespace

void PopSift::dispatchLoop( )
{
    Impl& p = *_impl;
    Slot s;
    pin_to_device_numa_node( _device );
    popsift::pool::set_thread_device( _device );        // result and export buffers: this device's pinned pool
    double t_attach = 0, t_upload = 0, t_submit = 0, t_frame = 0, t_wrap = 0, t_pool = 0; int n_done = 0;

    for( ;; ) {
        SiftJob* job = p.queue.pull();
        if( job == nullptr ) { p.queue.push( nullptr ); break; }      // pass the shutdown sentinel on

        popsift::FeaturesBase* f = nullptr;
        try {
            if( s.ctx == nullptr ) {
                psx_config pc;
                {
                    std::lock_guard<std::mutex> g( p.cfg_mutex );   // configure() is refused from now on (popsift.cpp:81-83)
                    p.contexts_exist = true;
                    to_psx( _config, pc );
                    if( pc.octaves < 0 && p.octaves_resolved >= 0 ) pc.octaves = p.octaves_resolved;
                }
                if( psx_create( _device, &pc, &s.ctx ) != PSX_OK ) {
                    const char* m = psx_last_error( nullptr );
                    s.ctx = nullptr;
                    throw std::runtime_error( std::string( "psx_create failed:\n    " ) + ( m ? m : "" ) );
                }
                psx_set_wait_mode( s.ctx, 1 );                       // sleep, do not spin: there are PIPE_DEPTH of us
                if( _proc_mode == popsift::Config::ExtractingMode ) {
                    s.xfeat = (psx_feature*)popsift::pool::get_pinned( (size_t)EXPORT_FEATURES * sizeof(psx_feature), &s.xfeat_cap );
                    if( s.xfeat == nullptr ) throw std::runtime_error( "out of host memory for export buffers" );
                }
            }
            const double t0 = pnow();
            // Results reach the host by DMA after the frame (psx_download into pinned pool buffers that the
            // FeaturesHost then owns).  The alternative, POPSIFT_EXPORT=1, lets the descriptor kernel store them
            // straight into mapped host memory (psx_attach_export_mapped): no second wait, but the PCIe stores slow
            // the kernel down -- measured 8 % slower end to end at 18 k descriptors per frame on MI355X (round 5: bench.py's
            // export leg; round 2 had 4750 against 5340 Mpix/s), so it is opt-in.
            static const bool use_export = []{ const char* e = getenv( "POPSIFT_EXPORT" ); return e != nullptr && e[0] == '1'; }();
            if( _proc_mode == popsift::Config::ExtractingMode && s.xdesc == nullptr && use_export ) {
                // the previous result took this context's descriptor buffer with it: attach a fresh one
                const int want = p.want_desc;
                const double tp0 = pnow();
                s.xdesc = (float*)popsift::pool::get_pinned( (size_t)want * sizeof(popsift::Descriptor), &s.xdesc_cap );
                t_pool += pnow() - tp0;
                if( s.xdesc == nullptr ) throw std::runtime_error( "out of host memory for export buffers" );
                s.desc_cap = (int)std::min<size_t>( s.xdesc_cap / sizeof(popsift::Descriptor), (size_t)1 << 30 );
                check( s.ctx, psx_attach_export_mapped( s.ctx, s.xfeat, EXPORT_FEATURES, s.xdesc, s.desc_cap ), "psx_attach_export_mapped" );
            }
            const double t1 = pnow();
            popsift::trace::Range r_frame( "popsift frame" );
            { popsift::trace::Range r_up( "inserting image" );                   // popsift.cpp:441-446
            if( job->isPinned() )
                check( s.ctx, psx_upload_pinned( s.ctx, job->getData(), job->getWidth(), job->getHeight(), job->isFloat() ? 1 : 0 ), "psx_upload_pinned" );
            else if( job->isFloat() )
                check( s.ctx, psx_upload_f32( s.ctx, (const float*)job->getData(), job->getWidth(), job->getHeight() ), "psx_upload_f32" );
            else
                check( s.ctx, psx_upload_u8( s.ctx, job->getData(), job->getWidth(), job->getHeight() ), "psx_upload_u8" );
            }
            const double t2 = pnow();
            { popsift::trace::Range r_ex( "extract (launch chain)" );
              check( s.ctx, psx_extract( s.ctx ), "psx_extract" ); }
            const double t3 = pnow();
            double tf = 0;
            if( _proc_mode == popsift::Config::ExtractingMode ) {
                popsift::trace::Range r_dl( "download descriptors" );              // sift_pyramid.cu:288-319
                f = collect_host( s, p.want_desc, &tf );
                if( _config.getLogMode() == popsift::Config::All ) {      // popsift.cpp:330-338
                    // the reference writes these dumps from its single worker (popsift.cpp:330-338); here PIPE_DEPTH
                    // workers would interleave writes to the same dir-octave / dir-dog / dir-desc files
                    std::lock_guard<std::mutex> lg( p.log_mutex );
                    std::string why;
                    if( !popsift::log_dump( s.ctx, static_cast<popsift::FeaturesHost*>( f ), _config.getUpscaleFactor(), "pyramid", &why ) )
                        throw std::runtime_error( "log output failed:\n    " + why );
                }
            }
            else                                                f = collect_dev( s, _device );
            const double t4 = pnow();
            t_attach += t1 - t0; t_upload += t2 - t1; t_submit += t3 - t2; t_frame += tf; t_wrap += t4 - t3 - tf; n_done++;
        } catch( ... ) {
            job->setError( std::current_exception() );
            f = nullptr;
        }
        job->setFeatures( f );
    }

    if( s.ctx ) { psx_attach_export( s.ctx, nullptr, 0, nullptr, 0 ); psx_destroy( s.ctx ); }
    popsift::pool::put_pinned( s.xfeat, s.xfeat_cap );
    popsift::pool::put_pinned( s.xdesc, s.xdesc_cap );
    {
        std::lock_guard<std::mutex> g( p.prof_mutex );
        p.t_pool += t_pool; p.t_attach += t_attach; p.t_upload += t_upload; p.t_submit += t_submit; p.t_frame += t_frame; p.t_wrap += t_wrap;
        p.n_done += n_done;
        struct timespec ts;
        if( clock_gettime( CLOCK_THREAD_CPUTIME_ID, &ts ) == 0 ) p.t_cpu += (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
}
