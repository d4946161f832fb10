// vstab_pipeline.cpp -- the FrameSourceWarp replacement behind the C ABI: tracking workspace,
// look-ahead ring in HBM, consume_frame / pull_frame state machine (FrameSourceWarp.cpp:397-476),
// plus the stateless tracking / motion entry points.  Host C++; every pixel touches a HIP kernel.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "vstab_geometry.hpp"
#include "vstab_hostlogic.hpp"
#include "vstab_internal.hpp"
#include "vstab_motion.hpp"
#include "vstab_track.hpp"

namespace vstab {

#define VSTAB_TRY(expr)                   \
    do {                                  \
        vstab_status st_ = (expr);        \
        if (st_ != VSTAB_OK) return st_;  \
    } while (0)

static bool debug_spec() {  // VSTAB_DEBUG_SPEC=1: the key-frame speculation narrated on stderr (development aid)
    static const bool on = getenv("VSTAB_DEBUG_SPEC") != nullptr;
    return on;
}

struct DevBuf {
    void *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr, n = 0;
    }
    vstab_status ensure(size_t bytes) {
        if (bytes <= n) return VSTAB_OK;
        release();
        if (hipMalloc(&p, bytes) != hipSuccess) return fail(VSTAB_ERR_NOMEM, "hipMalloc failed");
        n = bytes;
        return VSTAB_OK;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

struct PinnedBuf {  // host memory the device can read and write directly (mapped, coherent)
    void *p = nullptr;
    size_t n = 0;
    void *dev() const {
        void *d = nullptr;
        return hipHostGetDevicePointer(&d, p, 0) == hipSuccess ? d : nullptr;
    }
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    vstab_status ensure(size_t bytes) {
        if (bytes <= n) return VSTAB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr, n = 0;
        if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) return fail(VSTAB_ERR_NOMEM, "hipHostMalloc failed");
        n = bytes;
        return VSTAB_OK;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

// ---------------------------------------------------------------------------------------------
// Tracker: device workspace for goodFeaturesToTrack + calcOpticalFlowPyrLK
// ---------------------------------------------------------------------------------------------
constexpr int PREFETCH_MAX = 16;             // upper bound of the read-ahead (the ring and the pyramid sets are sized for it)
constexpr int PYR_SETS = PREFETCH_MAX + 2;  // previous + current (in flight) + the prefetched frames
#ifdef VSTAB_DEV
constexpr int PYR_DEV_EXTRA = 1;  // a set nobody reads: VSTAB_DEV_PYR_TWICE=1 builds every pyramid a second time into it (sensitivity of the frame rate to the pyramid kernels)
#else
constexpr int PYR_DEV_EXTRA = 0;
#endif
// frames pulled from upstream ahead of the one being tracked: deep enough that the speculative corner detection of a
// key frame (137 us of kernels beside everything else + the host selection) is finished before its turn comes
// default read-ahead; VSTAB_PREFETCH=n (1 .. PREFETCH_MAX) for experiments.  Twelve since the end of round 4 (eight before): the rates are the
// same, but the speculative corner detection launched when the frame before a planned key frame is read ahead then has ~430 us at 4K for
// its ~150 + 45 us beside the saturating warp instead of ~290 -- the margin that keeps a slow box from waiting for corners at key frames
constexpr int PREFETCH_DEPTH = 12;

class Tracker {
  public:
    // One record buffer per tracked frame pair, in rotation: a launch covers up to LK_SEG_MAX pairs and launches run up to
    // PREFETCH_DEPTH frames ahead of the frame the host reads, so a buffer comes round again long after its reader is done
    // (and after any launch whose results were dropped has finished: it precedes its replacement on the tracker stream).
    static constexpr int REC_BUFS = 32, PTS_BUFS = 8;
    vstab_status init(int w, int h) {
        w_ = w, h_ = h, levels_ = lk_levels(w, h);
        int lw = w, lh = h;
        for (int l = 1; l < levels_; l++) {
            lw = (lw + 1) / 2, lh = (lh + 1) / 2;
            lvl_w_[l] = lw, lvl_h_[l] = lh;
            for (int s = 0; s < PYR_SETS + PYR_DEV_EXTRA; s++) VSTAB_TRY(pyr_[s][l].ensure((size_t)lw * lh));
        }
        lvl_w_[0] = w, lvl_h_[0] = h;
        VSTAB_TRY(small_.ensure(256));
        VSTAB_TRY(hsmall_.ensure(256));
        // record and point buffers for the pipeline's 200 features (FrameSourceWarp.cpp:230), so that nothing is
        // reallocated while a launch that uses them is queued
        for (int b = 0; b < REC_BUFS; b++) {
            VSTAB_TRY(hrec_[b].ensure(256 * 16));
            VSTAB_TRY(drec_[b].ensure(256 * 16));
            VSTAB_HIP_TRY(hipMemset(drec_[b].p, 0, 256 * 16));  // tag 0 is never a sequence number: a chained slot never mistakes stale bytes for its predecessor
        }
        for (int b = 0; b < PTS_BUFS; b++) VSTAB_TRY(hpts_[b].ensure(256 * sizeof(float2)));
        return VSTAB_OK;
    }

    // levels 1.. of the pyramid of `gray` into slot s (level 0 is the frame itself)
    // `done` (optional) completes with the LAST kernel of the pyramid, bound to that launch (launch_pyr_down); *done_bound says whether a kernel
    // took it (an image too small for a second level has no pyramid kernel: the caller records the event itself)
    // level 1 of set s, for a caller that fills it itself (launch_pack_pyr: the copy into the ring and the first level in one launch)
    uint8_t *level1(int s) { return levels_ >= 2 ? pyr_[s][1].as<uint8_t>() : nullptr; }
    size_t level1_pitch() const { return (size_t)lvl_w_[1]; }
    vstab_status build_pyramid(int s, const uint8_t *gray, size_t pitch, hipStream_t st, hipEvent_t done = nullptr, bool *done_bound = nullptr, bool have_level1 = false) {
#ifdef VSTAB_DEV
        static const bool twice = getenv("VSTAB_DEV_PYR_TWICE") != nullptr;
        if (twice && s != PYR_SETS) VSTAB_TRY(build_pyramid(PYR_SETS, gray, pitch, st));
#endif
        const uint8_t *src = gray;
        size_t sp = pitch;
        if (done_bound) *done_bound = false;
        for (int l = 1; l < levels_; l++) {
            // levels 2 and 3 in ONE launch (k_pyr_down_x2): as kernels of their own the small levels are launch- and latency-bound
            if (l == 2 && levels_ == 4 && pyr_down_x2_ok(lvl_w_[1], lvl_h_[1]) && !single_level_pyramid_) {
                VSTAB_TRY(launch_pyr_down_x2(src, sp, lvl_w_[1], lvl_h_[1], pyr_[s][2].as<uint8_t>(), (size_t)lvl_w_[2], pyr_[s][3].as<uint8_t>(), (size_t)lvl_w_[3], st, done));
                if (done_bound) *done_bound = done != nullptr;
                break;
            }
            const bool last = l == levels_ - 1;
            if (l == 1 && have_level1) {  // (written by k_pack_pyr together with the copy; if it is the only level the caller records the event)
                src = pyr_[s][l].as<uint8_t>(), sp = (size_t)lvl_w_[l];
                continue;
            }
            VSTAB_TRY(launch_pyr_down(src, sp, lvl_w_[l - 1], lvl_h_[l - 1], pyr_[s][l].as<uint8_t>(), (size_t)lvl_w_[l], st, last ? done : nullptr));
            if (last && done_bound) *done_bound = done != nullptr;
            src = pyr_[s][l].as<uint8_t>(), sp = (size_t)lvl_w_[l];
        }
        return VSTAB_OK;
    }

    LkPyramid pyramid(int s, const uint8_t *gray, size_t pitch) const {
        LkPyramid p;
        p.levels = levels_;
        for (int l = 0; l < LK_MAX_LEVELS; l++) p.img[l] = nullptr, p.pitch[l] = 0, p.w[l] = p.h[l] = 0;
        p.img[0] = gray, p.pitch[0] = pitch, p.w[0] = w_, p.h[0] = h_;
        for (int l = 1; l < levels_; l++) p.img[l] = pyr_[s][l].as<uint8_t>(), p.pitch[l] = (size_t)lvl_w_[l], p.w[l] = lvl_w_[l], p.h[l] = lvl_h_[l];
        return p;
    }

    // host half of goodFeaturesToTrack: sort the candidate keys (value descending, ties -> later raster
    // position first: greaterThanPtr in OpenCV) and run the minimum-distance grid
    void select_corners(unsigned long long *k, unsigned int n, int max_corners, double min_distance, std::vector<float> &xy) {
        xy.clear();
        const int cell = (int)std::nearbyint(min_distance);
        const int gw = cell >= 1 ? (w_ + cell - 1) / cell : 0, gh = cell >= 1 ? (h_ + cell - 1) / cell : 0;
        const double md2 = min_distance * min_distance;
        static thread_local std::vector<int> grid_head_, grid_next_;  // min-distance grid: per-cell lists of accepted corners
        if (cell >= 1) grid_head_.assign((size_t)gw * gh, -1), grid_next_.clear();
        // The greedy pass consumes candidates in sorted order and usually stops after a few hundred, so the
        // keys are sorted lazily in chunks: nth_element splits off the next `chunk` largest keys (O(n)),
        // only that chunk is sorted.  The visiting order is exactly the fully sorted order.
        unsigned int done = 0;
        const auto greater = [](unsigned long long a, unsigned long long b) { return a > b; };
        while (done < n) {
            const unsigned int chunk = std::min(n - done, 1024u);
            if (done + chunk < n) std::nth_element(k + done, k + done + chunk, k + n, greater);
            std::sort(k + done, k + done + chunk, greater);
            for (unsigned int i = done; i < done + chunk; i++) {
                const unsigned int idx = (unsigned int)(k[i] & 0xffffffffu);
                const int x = (int)(idx % w_), y = (int)(idx / w_);
                if (cell < 1) {
                    xy.push_back((float)x), xy.push_back((float)y);
                } else {
                    const int xc = x / cell, yc = y / cell;
                    const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
                    bool good = true;
                    for (int yy = y1; yy <= y2 && good; yy++)
                        for (int xx = x1; xx <= x2 && good; xx++)
                            for (int j = grid_head_[(size_t)yy * gw + xx]; j >= 0; j = grid_next_[j]) {
                                const float dx = (float)x - xy[2 * j], dy = (float)y - xy[2 * j + 1];
                                if ((double)(dx * dx + dy * dy) < md2) {
                                    good = false;
                                    break;
                                }
                            }
                    if (!good) continue;
                    grid_next_.push_back(grid_head_[(size_t)yc * gw + xc]);
                    grid_head_[(size_t)yc * gw + xc] = (int)(xy.size() / 2);
                    xy.push_back((float)x), xy.push_back((float)y);
                }
                if (max_corners > 0 && (int)(xy.size() / 2) == max_corners) return;
            }
            done += chunk;
        }
    }

    // goodFeaturesToTrack(gray, max_corners, quality, min_distance); synchronises the stream
    vstab_status good_features(const uint8_t *gray, size_t pitch, int max_corners, double quality, double min_distance,
                               std::vector<float> &xy, hipStream_t st, float *eig_out = nullptr) {
        xy.clear();
        float *eig = eig_out;
        int *max_bits = small_.as<int>();
        unsigned int *count = small_.as<unsigned int>() + 4;
        if (cap_ == 0) {
            cap_ = 1u << 18;
            VSTAB_TRY(keys_.ensure(sizeof(unsigned long long) * cap_));
        }
        if (!eig_out && !two_pass_detector_) {
            // one pass: eigenvalue, threshold and 3x3 maximum test fused, the eigenvalue map never stored
            VSTAB_TRY(raw_keys_.ensure(corners_fused_scratch_bytes(w_, h_)));
#ifdef VSTAB_DEV
            if (getenv("VSTAB_DEV_DET_TWICE")) VSTAB_TRY(launch_corners_fused(gray, pitch, w_, h_, quality, raw_keys_.p, keys_.as<unsigned long long>(), cap_, small_.as<unsigned int>(), st));
#endif
            VSTAB_TRY(launch_corners_fused(gray, pitch, w_, h_, quality, raw_keys_.p, keys_.as<unsigned long long>(), cap_, small_.as<unsigned int>(), st));
            VSTAB_HIP_TRY(hipMemcpyAsync(hsmall_.p, count, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
            VSTAB_HIP_TRY(hipStreamSynchronize(st));
            const unsigned int kept = hsmall_.as<unsigned int>()[0];
            if (kept <= cap_) {
                if (kept == 0) return VSTAB_OK;
                VSTAB_TRY(hkeys_.ensure(sizeof(unsigned long long) * kept));
                VSTAB_HIP_TRY(hipMemcpyAsync(hkeys_.p, keys_.p, sizeof(unsigned long long) * kept, hipMemcpyDeviceToHost, st));
                VSTAB_HIP_TRY(hipStreamSynchronize(st));
                select_corners(hkeys_.as<unsigned long long>(), kept, max_corners, min_distance, xy);
                return VSTAB_OK;
            }
            fused_overflows_++;  // more corners above the threshold than the key buffer holds: the two-pass detector below grows it
        }
        VSTAB_TRY(eig_.ensure(sizeof(float) * (size_t)w_ * h_));
        if (!eig) eig = eig_.as<float>();
        VSTAB_TRY(launch_min_eig(gray, pitch, w_, h_, eig, max_bits, st));
        unsigned int n = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            VSTAB_TRY(launch_corner_candidates(eig, w_, h_, max_bits, quality, keys_.as<unsigned long long>(), count, cap_, st));
            VSTAB_HIP_TRY(hipMemcpyAsync(hsmall_.p, count, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
            VSTAB_HIP_TRY(hipStreamSynchronize(st));
            n = *hsmall_.as<unsigned int>();
            if (n <= cap_) break;
            cap_ = n;  // more local maxima than the buffer holds: grow and re-run the compaction
            VSTAB_TRY(keys_.ensure(sizeof(unsigned long long) * cap_));
        }
        if (n == 0) return VSTAB_OK;
        VSTAB_TRY(hkeys_.ensure(sizeof(unsigned long long) * n));
        VSTAB_HIP_TRY(hipMemcpyAsync(hkeys_.p, keys_.p, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, st));
        VSTAB_HIP_TRY(hipStreamSynchronize(st));
        select_corners(hkeys_.as<unsigned long long>(), n, max_corners, min_distance, xy);
        return VSTAB_OK;
    }

    // Speculative detection: the same two kernels enqueued on another stream ahead of time (the counter
    // half of the key-frame rule is predictable), with the count and the first SPEC_CAP keys copied to
    // pinned memory behind them.  spec_finish() only has to wait for the event and run the host half.
    static constexpr unsigned int SPEC_CAP = 1u << 15;
    vstab_status spec_launch(const uint8_t *gray, size_t pitch, double quality, hipStream_t st, long tag) {
        spec_join();  // (a previous asynchronous selection still reading the pinned buffer: never in practice)
        VSTAB_TRY(spec_raw_.ensure(corners_fused_scratch_bytes(w_, h_)));
        VSTAB_TRY(spec_keys_.ensure(sizeof(unsigned long long) * SPEC_CAP));
        VSTAB_TRY(spec_small_.ensure(256));
        VSTAB_TRY(spec_host_.ensure(64 + sizeof(unsigned long long) * SPEC_CAP));
        if (!spec_ev_) VSTAB_HIP_TRY(hipEventCreateWithFlags(&spec_ev_, hipEventDisableTiming));
        unsigned int *count = spec_small_.as<unsigned int>() + 4;
#ifdef VSTAB_DEV
        static const bool det_twice = getenv("VSTAB_DEV_DET_TWICE") != nullptr;  // sensitivity of the frame rate to the detector: everything twice, same result
        if (det_twice) VSTAB_TRY(launch_corners_fused(gray, pitch, w_, h_, quality, spec_raw_.p, spec_keys_.as<unsigned long long>(), SPEC_CAP, spec_small_.as<unsigned int>(), st));
#endif
        VSTAB_TRY(launch_corners_fused(gray, pitch, w_, h_, quality, spec_raw_.p, spec_keys_.as<unsigned long long>(), SPEC_CAP, spec_small_.as<unsigned int>(), st));
        VSTAB_HIP_TRY(hipMemcpyAsync(spec_host_.p, count, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));  // {keys kept, tiles that spilled}
        VSTAB_HIP_TRY(hipMemcpyAsync(spec_host_.as<uint8_t>() + 64, spec_keys_.p, sizeof(unsigned long long) * SPEC_CAP, hipMemcpyDeviceToHost, st));
        VSTAB_HIP_TRY(hipEventRecord(spec_ev_, st));
        spec_tag_ = tag;
        return VSTAB_OK;
    }
    long spec_tag() const { return spec_tag_; }
    long selections_by_caller() const { return selections_by_caller_; }
    long selections_by_helper() const { return selections_by_helper_.load(std::memory_order_relaxed); }
    void set_two_pass_detector(bool on) { two_pass_detector_ = on; }
    long fused_overflows() const { return fused_overflows_; }
    // Host half of the speculative detection on a helper thread: waits for the kernels' results and runs the
    // sort + minimum-distance pass, so that by the time the key frame comes its corners are simply there.
    void spec_select_async(int max_corners, double min_distance) {
        spec_join();
        spec_owner_.store(0, std::memory_order_release);  // nobody has taken this selection yet (the helper thread, or the caller: spec_poll_inline)
        spec_state_.store(1, std::memory_order_release);
        if (!spec_thread_started_) {
            spec_thread_started_ = true;
            int dev = 0;
            (void)hipGetDevice(&dev);
            spec_thread_ = std::thread([this, dev] {
                (void)hipSetDevice(dev);  // the handle's device, not the new thread's default
                std::unique_lock<std::mutex> lk(spec_m_);
                for (;;) {
                    spec_cv_.wait(lk, [this] { return spec_job_ || spec_quit_; });
                    if (spec_quit_) return;
                    spec_job_ = false;
                    {
                        // (development: VSTAB_SPEC_HELPER_DELAY_US=n makes this thread wake up late, so that a test reaches spec_poll_inline;
                        //  read ONCE, when the Tracker is constructed -- getenv beside a setenv of the host process is undefined behaviour)
                        const long late_us = spec_late_us_;
                        if (late_us > 0) {
                            lk.unlock();
                            std::this_thread::sleep_for(std::chrono::microseconds(late_us));
                            lk.lock();
                            if (spec_quit_) return;
                        }
                        int unclaimed = 0;  // the caller may have done this selection itself while this thread was waking up
                        if (!spec_owner_.compare_exchange_strong(unclaimed, 1, std::memory_order_acq_rel)) continue;
                        selections_by_helper_.fetch_add(1, std::memory_order_relaxed);
                    }
                    lk.unlock();
                    int result = 3;
                    const auto t0 = std::chrono::steady_clock::now();
                    // poll instead of a blocking wait: the wake-up latency of hipEventSynchronize (hundreds of microseconds
                    // on this runtime) would eat the lead the detection was given
                    hipError_t q = hipErrorNotReady;
                    if (spec_ev_) {
                        for (long spins = 0; (q = hipEventQuery(spec_ev_)) == hipErrorNotReady && spins < 4000000; spins++) __builtin_ia32_pause();
                        if (q == hipErrorNotReady) q = hipEventSynchronize(spec_ev_);
                    }
                    if (q == hipSuccess) {
                        const auto t1 = std::chrono::steady_clock::now();
                        const unsigned int n = spec_host_.as<unsigned int>()[0];
                        if (n <= SPEC_CAP) {
                            select_corners(reinterpret_cast<unsigned long long *>(spec_host_.as<uint8_t>() + 64), n, spec_max_, spec_dist_, spec_xy_);
                            result = 2;
                        }
                        if (debug_spec())
                            std::fprintf(stderr, "async selection: waited %.0f us for the detection, selected %zu of %u candidates in %.0f us\n",
                                         std::chrono::duration<double, std::micro>(t1 - t0).count(), spec_xy_.size() / 2, n,
                                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count());
                    }
                    spec_state_.store(result, std::memory_order_release);
                    lk.lock();
                }
            });
        }
        {
            std::lock_guard<std::mutex> lk(spec_m_);
            spec_max_ = max_corners, spec_dist_ = min_distance, spec_job_ = true;
        }
        spec_cv_.notify_one();
    }
    // The helper thread sleeps on a condition variable between two detections (720 us apart at 4K); on a busy host its wake-up can take
    // longer than the detection itself.  Whoever looks for the corners first while the job is still unclaimed and the detection's event
    // has completed does the selection on the spot (45 us of host time instead of a wait for another thread's wake-up).
    void spec_poll_inline() {
        if (spec_state_.load(std::memory_order_acquire) != 1 || !spec_ev_ || hipEventQuery(spec_ev_) != hipSuccess) {
            (void)hipGetLastError();  // "not ready" is an answer, not an error the next launch check should find
            return;
        }
        int unclaimed = 0;
        if (!spec_owner_.compare_exchange_strong(unclaimed, 2, std::memory_order_acq_rel)) return;  // the helper thread has it
        selections_by_caller_++;
        int result = 3;
        const unsigned int n = spec_host_.as<unsigned int>()[0];
        if (n <= SPEC_CAP) {
            select_corners(reinterpret_cast<unsigned long long *>(spec_host_.as<uint8_t>() + 64), n, spec_max_, spec_dist_, spec_xy_);
            result = 2;
        }
        if (debug_spec()) std::fprintf(stderr, "selection done by the caller (the helper thread had not woken up): %zu of %u candidates\n", spec_xy_.size() / 2, n);
        spec_state_.store(result, std::memory_order_release);
    }
    // 0 = no asynchronous selection, 1 = running, 2 = corners ready, 3 = failed (candidate overflow / device error)
    int spec_state() const { return spec_state_.load(std::memory_order_acquire); }
    void spec_join() {
        while (spec_state_.load(std::memory_order_acquire) == 1) std::this_thread::yield();
    }
    // take the asynchronously selected corners (state must be 2)
    void spec_take(std::vector<float> &xy) {
        xy = spec_xy_;
        spec_state_.store(0, std::memory_order_release), spec_tag_ = -1;
    }
    ~Tracker() {
        // the helper thread polls spec_ev_ while a selection is running: stop and join it BEFORE the events go
        if (spec_thread_started_) {
            {
                std::lock_guard<std::mutex> lk(spec_m_);
                spec_quit_ = true;
            }
            spec_cv_.notify_one();
            spec_thread_.join();
        }
        for (hipEvent_t e : {spec_ev_, ev_a_, ev_b_})
            if (e) (void)hipEventDestroy(e);
    }
    // returns true and fills xy if the speculative result is usable (candidate count within SPEC_CAP)
    bool spec_finish(int max_corners, double min_distance, std::vector<float> &xy) {
        if (spec_state() != 0) {  // the helper thread has (or is about to have) the answer -- or nobody yet: then this thread, if the kernels are through
            spec_poll_inline();
            spec_join();
            const bool ok = spec_state() == 2;
            if (ok) xy = spec_xy_;
            spec_state_.store(0, std::memory_order_release), spec_tag_ = -1;
            return ok;
        }
        spec_tag_ = -1;
        if (!spec_ev_ || hipEventSynchronize(spec_ev_) != hipSuccess) return false;
        const unsigned int n = spec_host_.as<unsigned int>()[0];
        if (n > SPEC_CAP) return false;
        select_corners(reinterpret_cast<unsigned long long *>(spec_host_.as<uint8_t>() + 64), n, max_corners, min_distance, xy);
        return true;
    }

    // calcOpticalFlowPyrLK(prev, next, pts), split in two so the caller can enqueue more work behind the
    // kernel before blocking.  Points travel through mapped host memory: the kernel reads prev_pts and
    // writes one self-validating record per feature over the link directly (a few KB), and the host polls
    // the records' sequence tags instead of paying a copy launch + stream-sync round trip.
    //
    // A launch is a SEGMENT of consecutive frame pairs (k_lk_track): pair i tracks from pyramid i into pyramid i + 1, every
    // slot starts pair i + 1 from the point it reached in pair i (FrameSourceWarp.cpp:427) and lost slots stay lost.
    // Chained launches: a segment that continues where another ended reads its start points from the device copy of the
    // parent's last records, so it can be enqueued without waiting for the host.
    struct Launch {
        int n_slots = 0, n_frames = 0;
        int buf[LK_SEG_MAX] = {0};        // record buffer of every frame pair
        uint32_t seq[LK_SEG_MAX] = {0};   // and its sequence tag
        bool chained = false, timed = false;
    };

    // pyr: n_frames + 1 pyramids (the frame before the segment's first, then the segment's frames)
    vstab_status track_launch(const LkPyramid *pyr, int n_frames, const std::vector<float> &prev_xy, hipStream_t st, bool timed, Launch &L) {
        L = Launch();
        L.n_slots = (int)(prev_xy.size() / 2), L.timed = timed;
        if (L.n_slots == 0) return VSTAB_OK;
        PinnedBuf &pts = hpts_[pts_launches_++ % PTS_BUFS];  // one per launch: a launch still queued keeps its points
        VSTAB_TRY(pts.ensure((size_t)L.n_slots * sizeof(float2)));
        if (!pts.dev()) return fail(VSTAB_ERR_DEVICE, "hipHostGetDevicePointer failed");
        std::memcpy(pts.p, prev_xy.data(), sizeof(float) * prev_xy.size());
        return launch_segment(pyr, n_frames, static_cast<const float2 *>(pts.dev()), nullptr, 0, st, L);
    }

    // the launch for the frames FOLLOWING `parent`'s last one, chained behind it on the same stream (same slots; see above)
    vstab_status track_launch_chained(const LkPyramid *pyr, int n_frames, const Launch &parent, hipStream_t st, Launch &L) {
        L = Launch();
        if (parent.n_slots == 0 || parent.n_frames == 0) return VSTAB_OK;
        L.n_slots = parent.n_slots, L.chained = true;
        const int last = parent.n_frames - 1;
        return launch_segment(pyr, n_frames, nullptr, drec_[parent.buf[last]].p, parent.seq[last], st, L);
    }

    // results of frame pair `idx` of launch L in the order of the (compacted) point list it tracked: expect_n entries
    vstab_status track_wait(const Launch &L, int idx, size_t expect_n, std::vector<float> &next_xy, std::vector<uint8_t> &status, hipStream_t st,
                            double *gpu_ms) {
        next_xy.clear(), status.clear();
        next_xy.reserve(2 * expect_n), status.reserve(expect_n);
        const int n = L.n_slots;
        if (n == 0) return expect_n == 0 ? VSTAB_OK : fail(VSTAB_ERR_DEVICE, "tracker bookkeeping mismatch");
        if (idx < 0 || idx >= L.n_frames) return fail(VSTAB_ERR_DEVICE, "tracker bookkeeping mismatch (frame pair outside its launch)");
        const uint32_t seq = L.seq[idx];
        const volatile uint32_t *rec = hrec_[L.buf[idx]].as<uint32_t>();
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long spins = 0;
        int next = 0;
        for (;;) {  // (vstab_hostlogic.hpp: records are decoded as they arrive; a record is valid once both of its tags are)
            const LkParse r = lk_parse_records(rec, next, n, seq, expect_n, next_xy, status, &next);
            if (r == LK_PARSE_OK) break;
            if (r == LK_PARSE_BAD_CHAIN) return fail(VSTAB_ERR_DEVICE, "LK chain: a slot's predecessor record does not carry its parent's tag");
            if (r == LK_PARSE_COUNT_MISMATCH) return fail(VSTAB_ERR_DEVICE, "tracker bookkeeping mismatch");
            while (!lk_record_ready(rec, next, seq)) {
                __builtin_ia32_pause();
                if ((++spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    VSTAB_HIP_TRY(hipStreamSynchronize(st));  // surfaces a launch / execution error if there is one
                    if (!lk_record_ready(rec, next, seq)) return fail(VSTAB_ERR_DEVICE, "LK kernel did not complete");
                }
            }
        }
        if (L.timed && gpu_ms) {
            float ms = 0;
            if (hipEventSynchronize(ev_b_) == hipSuccess && hipEventElapsedTime(&ms, ev_a_, ev_b_) == hipSuccess) *gpu_ms += ms;
        }
        return VSTAB_OK;
    }

    vstab_status track(const LkPyramid &I, const LkPyramid &J, const std::vector<float> &prev_xy, std::vector<float> &next_xy,
                       std::vector<uint8_t> &status, hipStream_t st, double *gpu_ms = nullptr) {
        Launch L;
        const LkPyramid pyr[2] = {I, J};
        VSTAB_TRY(track_launch(pyr, 1, prev_xy, st, gpu_ms != nullptr, L));
        return track_wait(L, 0, prev_xy.size() / 2, next_xy, status, st, gpu_ms);
    }

    int levels() const { return levels_; }

    // VSTAB_LK_CLOCK=1 (development aid): every LK launch stamps its first-workgroup start and last-workgroup end
    // (100 MHz wall clock) into a slot of a mapped ring; report_clock() prints durations and start-to-start gaps
    void *clock_slot() {
        static const bool on = getenv("VSTAB_LK_CLOCK") != nullptr;
        if (!on) return nullptr;
        if (!clk_.p) {
            if (clk_.ensure(sizeof(unsigned long long) * 2 * CLK_N) != VSTAB_OK) return nullptr;
            std::vector<unsigned long long> init(2 * CLK_N, 0);
            for (int i = 0; i < CLK_N; i++) init[2 * i] = ~0ull;
            if (hipMemcpy(clk_.p, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        }
        if (clk_used_ >= CLK_N) return nullptr;
        return clk_.as<unsigned long long>() + 2 * (clk_used_++);
    }
    void report_clock() {
        if (!clk_.p || clk_used_ < 200) return;
        std::vector<unsigned long long> host(2 * CLK_N);
        if (hipMemcpy(host.data(), clk_.p, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
        const unsigned long long *c = host.data();
        double dur = 0, gap = 0, idle = 0;
        int n = 0;
        for (int i = clk_used_ - 400 > 0 ? clk_used_ - 400 : 1; i < clk_used_; i++) {
            if (c[2 * i + 1] == 0 || c[2 * i - 1] == 0) continue;
            dur += (c[2 * i + 1] - c[2 * i]) * 0.01, gap += ((double)c[2 * i] - (double)c[2 * i - 2]) * 0.01, idle += ((double)c[2 * i] - (double)c[2 * i - 1]) * 0.01;
            n++;
        }
        if (n) std::fprintf(stderr, "LK launches (last %d): duration %.1f us, start-to-start %.1f us, idle before start %.1f us\n", n, dur / n, gap / n, idle / n);
        std::vector<double> idles, durs;
        for (int i = clk_used_ - 400 > 0 ? clk_used_ - 400 : 1; i < clk_used_; i++)
            if (c[2 * i + 1] && c[2 * i - 1]) idles.push_back(((double)c[2 * i] - (double)c[2 * i - 1]) * 0.01), durs.push_back((c[2 * i + 1] - c[2 * i]) * 0.01);
        std::sort(idles.begin(), idles.end()), std::sort(durs.begin(), durs.end());
        if (idles.size() > 10) {
            const size_t m = idles.size();
            std::fprintf(stderr, "  idle percentiles 10/50/90/99: %.1f %.1f %.1f %.1f   duration 10/50/90/99: %.1f %.1f %.1f %.1f\n", idles[m / 10], idles[m / 2],
                         idles[m * 9 / 10], idles[m * 99 / 100], durs[m / 10], durs[m / 2], durs[m * 9 / 10], durs[m * 99 / 100]);
        }
    }

  private:
    vstab_status launch_segment(const LkPyramid *pyr, int n_frames, const float2 *prev_pts, const void *chain_in, uint32_t parent_seq, hipStream_t st,
                                Launch &L) {
        if (n_frames < 1 || n_frames > LK_SEG_MAX) return fail(VSTAB_ERR_INVALID, "tracker: a launch covers 1 .. LK_SEG_MAX frame pairs");
        LkSegArgs a;
        std::memset(&a, 0, sizeof(a));
        L.n_frames = n_frames;
        for (int i = 0; i <= n_frames; i++) a.pyr[i] = pyr[i];
        for (int i = 0; i < n_frames; i++) {
            L.buf[i] = (int)(rec_next_++ % REC_BUFS);
            if (++seq_ == 0) ++seq_;  // tag 0 means "never written"
            L.seq[i] = seq_;
            VSTAB_TRY(hrec_[L.buf[i]].ensure((size_t)L.n_slots * 16));
            VSTAB_TRY(drec_[L.buf[i]].ensure((size_t)L.n_slots * 16));
            if (!hrec_[L.buf[i]].dev()) return fail(VSTAB_ERR_DEVICE, "hipHostGetDevicePointer failed");
            a.host_rec[i] = static_cast<uint4 *>(hrec_[L.buf[i]].dev()), a.dev_rec[i] = drec_[L.buf[i]].as<uint4>(), a.seq[i] = L.seq[i];
        }
        a.n_frames = n_frames, a.n = L.n_slots;
        a.prev_pts = prev_pts, a.chain_in = static_cast<const uint4 *>(chain_in), a.parent_seq = parent_seq;
        a.clk = static_cast<unsigned long long *>(clock_slot());
        if (L.timed && !ev_a_) (void)hipEventCreate(&ev_a_), (void)hipEventCreate(&ev_b_);
        if (L.timed) (void)hipEventRecord(ev_a_, st);
        VSTAB_TRY(launch_lk(a, st));
        if (L.timed) (void)hipEventRecord(ev_b_, st);
        return VSTAB_OK;
    }

    int w_ = 0, h_ = 0, levels_ = 1;
    int lvl_w_[LK_MAX_LEVELS] = {0}, lvl_h_[LK_MAX_LEVELS] = {0};
    DevBuf pyr_[PYR_SETS + PYR_DEV_EXTRA][LK_MAX_LEVELS], eig_, keys_, small_;  // pyramid sets: previous, current, prefetched x2
    DevBuf spec_raw_, spec_keys_, spec_small_, raw_keys_;
    bool two_pass_detector_ = false;
    const bool single_level_pyramid_ = getenv("VSTAB_PYR_SINGLE") != nullptr;  // development: one launch per pyramid level
    long fused_overflows_ = 0;
    PinnedBuf spec_host_;
    hipEvent_t spec_ev_ = nullptr;
    long spec_tag_ = -1;
    std::thread spec_thread_;
    bool spec_thread_started_ = false, spec_job_ = false, spec_quit_ = false;
    std::mutex spec_m_;
    std::condition_variable spec_cv_;
    std::atomic<int> spec_state_{0};
    std::atomic<int> spec_owner_{0};  // who runs the posted selection: 0 nobody yet, 1 the helper thread, 2 the caller (spec_poll_inline)
    const long spec_late_us_ = getenv("VSTAB_SPEC_HELPER_DELAY_US") ? atol(getenv("VSTAB_SPEC_HELPER_DELAY_US")) : 0;  // development: the helper wakes up late
    long selections_by_caller_ = 0;                 // speculative detections whose corners the caller selected itself / the helper thread selected
    std::atomic<long> selections_by_helper_{0};
    int spec_max_ = 200;
    double spec_dist_ = 30.0;
    std::vector<float> spec_xy_;
    static constexpr int CLK_N = 4096;
    DevBuf clk_;
    int clk_used_ = 0;
    PinnedBuf hsmall_, hkeys_, hpts_[PTS_BUFS], hrec_[REC_BUFS];
    DevBuf drec_[REC_BUFS];
    unsigned int cap_ = 0;
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;
    unsigned long rec_next_ = 0, pts_launches_ = 0;
    uint32_t seq_ = 0;
};

// ---------------------------------------------------------------------------------------------
// EstimateWorker: one helper thread per handle that runs guess_camera_rotation's arithmetic
// (estimate_rotation: undistortion, RANSAC, LM refit -- pure host code on <= 200 points) while the
// calling thread issues the next frame's HIP launches.  One job at a time, posted and joined by the
// calling thread inside the same vstab_pull_frame call, so results are applied in frame order.  The
// worker spins briefly for the next job (the pipeline posts one every ~60 us) and then sleeps.
// ---------------------------------------------------------------------------------------------
class EstimateWorker {
  public:
    ~EstimateWorker() {
        if (th_.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                state_.store(QUIT, std::memory_order_release);
            }
            cv_.notify_one();
            th_.join();
        }
    }
    void post(const float *prev, const float *cur, int n, const Mat3 *Kin, const Mat3 *Kout, Pcg32 *rng, bool in_fish) {
        if (!th_.joinable()) th_ = std::thread([this] { run(); });
        prev_ = prev, cur_ = cur, n_ = n, Kin_ = Kin, Kout_ = Kout, rng_ = rng, in_fish_ = in_fish;
        {
            std::lock_guard<std::mutex> lk(m_);
            state_.store(POSTED, std::memory_order_release);
        }
        cv_.notify_one();
    }
    int join(Mat3 &R) {  // blocks until the posted job is done
        for (long spins = 0; state_.load(std::memory_order_acquire) != DONE; spins++) {
            if (spins < 200000) __builtin_ia32_pause();
            else std::this_thread::yield();
        }
        state_.store(IDLE, std::memory_order_relaxed);
        R = R_;
        return inliers_;
    }

  private:
    enum { IDLE = 0, POSTED = 1, DONE = 2, QUIT = 3 };
    void run() {
        for (;;) {
            int st = state_.load(std::memory_order_acquire);
            for (int spins = 0; st != POSTED && st != QUIT && spins < 20000; spins++) {
                __builtin_ia32_pause();
                st = state_.load(std::memory_order_acquire);
            }
            if (st != POSTED && st != QUIT) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { const int s = state_.load(std::memory_order_acquire); return s == POSTED || s == QUIT; });
                st = state_.load(std::memory_order_acquire);
            }
            if (st == QUIT) return;
            inliers_ = estimate_rotation(prev_, cur_, n_, *Kin_, *Kout_, *rng_, R_, in_fish_);
            int posted = POSTED;  // a destructor that stored QUIT meanwhile must not be answered with DONE
            if (!state_.compare_exchange_strong(posted, DONE, std::memory_order_acq_rel)) return;
        }
    }
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<int> state_{IDLE};
    const float *prev_ = nullptr, *cur_ = nullptr;
    int n_ = 0;
    const Mat3 *Kin_ = nullptr, *Kout_ = nullptr;
    Pcg32 *rng_ = nullptr;
    bool in_fish_ = true;
    Mat3 R_;
    int inliers_ = 0;
};

}  // namespace vstab

using namespace vstab;

// ---------------------------------------------------------------------------------------------
// the pipeline handle
// ---------------------------------------------------------------------------------------------
struct vstab_handle {
    // every way out of vstab_create after the streams and events exist, and vstab_destroy, ends here
    ~vstab_handle() {
        for (hipStream_t s : {tstream, pstream, dstream})
            if (s) (void)hipStreamSynchronize(s);
        for (auto &pe : pending) (void)hipEventDestroy(pe.a), (void)hipEventDestroy(pe.b);
        for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
        for (auto &s : slots)
            if (s.ingested) (void)hipEventDestroy(s.ingested);
        for (auto &s : slots)
            if (s.copied) (void)hipEventDestroy(s.copied);
        for (hipEvent_t e : warp_events)
            if (e) (void)hipEventDestroy(e);
        if (epoch_tail) (void)hipEventDestroy(epoch_tail);
        for (hipStream_t s : {dstream, pstream, tstream})
            if (s) (void)hipStreamDestroy(s);
        dmabufs.clear([](hipExternalMemory_t &e) { (void)hipDestroyExternalMemory(e); });
    }
    vstab_config cfg;
    vstab_source src;
    hipStream_t stream = nullptr;   // caller-visible stream: the warp runs here, dst is complete when it drains
    hipStream_t tstream = nullptr;  // internal stream: corner detection + LK (the per-frame critical path)
    hipStream_t pstream = nullptr;  // internal stream: ingest + pyramid of the NEXT frame (prefetch, overlaps LK)
    int w = 0, h = 0, ow = 0, oh = 0;
    Mat3 Kin, Kout;
    int map_mode = VSTAB_MAP_CREATEMAP_CL;  // createMap.cl for the preset path, a projection pair in lens mode
    bool in_fish = true;
    Tracker tracker;

    struct Slot {
        DevBuf buf;  // packed NV12, pitch = w (allocated on the first copy into the slot)
        // where the frame's planes are: in buf, or still in upstream's memory when upstream promised (vstab_frame.hold)
        // that they outlive the frame's whole stay in the pipeline -- then nothing is copied at all
        const uint8_t *y = nullptr, *uv = nullptr;
        size_t pitch_y = 0, pitch_uv = 0;
        bool borrowed = false;
        std::vector<float> feats;  // vstab_config.debug: the features tracked into this frame (input pixels)
        bool have_delta = false;  // upstream supplied this frame's rotation since the previous frame (vstab_frame.delta_rotation)
        Mat3 delta;
        DevBuf buf16;  // pixel_depth 10: the frame's P010 planes (luma rows of 2w bytes, then chroma), copied on ingest ...
        const uint8_t *y16 = nullptr, *uv16 = nullptr;  // ... or left where they are when upstream keeps them for good (hold >= 1 << 29)
        size_t pitch_y16 = 0, pitch_uv16 = 0;
        bool have_readout = false;  // ... and the rotation during the frame's read-out (vstab_frame.readout_rotation): rolling-shutter warp
        Mat3 readout;
        bool queued = false, last = false;
        long freed_at = 0;               // FIFO reuse: the slot idle the longest is taken first
        hipEvent_t ingested = nullptr;   // recorded on pstream after the copy into the slot (and its pyramid, when tracking)
        hipEvent_t copied = nullptr;     // completes with the copy kernel alone (8-bit frames copied by vstab_pack_nv12 while tracking): what
        bool copied_valid = false;       //   upstream's surface has to wait for -- the pyramid behind the copy reads the ring, not the surface
        int warped = -1;                 // index into warp_events of the event recorded behind the warp that read the slot
        bool warp_pending = false;       // a warp has read the slot since it was last filled
        unsigned long ingest_serial = 0;  // which copy `ingested` was last recorded for
    };
    struct PendingCopy {
        int slot;
        unsigned long serial;
        int hold;
    };
    // A frame used in place whose vstab_frame.hold is finite: upstream counts pull callbacks, the warp that reads the
    // planes runs on the caller's stream, so the callback at which the promise runs out first waits for that warp.
    struct PendingBorrow {
        unsigned long serial;
        int hold;
        bool warp_enqueued;
        int warped;  // index into warp_events, -1 until an event is recorded behind the warp
    };
    static constexpr int HOLD_FOREVER = 1 << 29;  // promises at least this long are not tracked
    // Event operations are the expensive HIP calls here (measured on this runtime: hipEventRecord 4.4 us,
    // hipStreamWaitEvent 3.4 us, a kernel launch 2.4 us, hipEventQuery 0.08 us), so the frame loop records as
    // few as it can: one event per ingested frame (behind copy + pyramid), one event per WARP_EVENT_STRIDE
    // warps (slots freed in between share the next one), and a stream only waits on an event that a host-side
    // query says is still pending.
    static constexpr int WARP_EVENT_STRIDE = 4, WARP_EVENT_POOL = 16;
    hipEvent_t warp_events[WARP_EVENT_POOL] = {};
    int warp_event_next = 0;
    std::vector<int> uncovered;  // slots whose warp is enqueued but not yet followed by a recorded event
    vstab_status cover_warps() {  // record one event behind every warp enqueued so far
        if (uncovered.empty() && !uncovered_borrows) return VSTAB_OK;
        const int e = warp_event_next++ % WARP_EVENT_POOL;
        VSTAB_HIP_TRY(hipEventRecord(warp_events[e], stream));
        for (int sl : uncovered) slots[sl].warped = e;
        uncovered.clear();
        if (uncovered_borrows)
            for (PendingBorrow &b : borrows)
                if (b.warp_enqueued && b.warped < 0) b.warped = e;
        uncovered_borrows = 0;
        return VSTAB_OK;
    }
    // host-side wait for an event: a short query spin (0.08 us a query), then a blocking wait
    static vstab_status host_wait(hipEvent_t ev) {
        int spins = 0;
        hipError_t q;
        while ((q = hipEventQuery(ev)) == hipErrorNotReady && ++spins < 20000) __builtin_ia32_pause();
        if (q == hipErrorNotReady) q = hipEventSynchronize(ev);
        VSTAB_HIP_TRY(q);
        return VSTAB_OK;
    }
    // make stream `waiter` wait for `ev` unless the host can already see that it has completed
    static vstab_status wait_if_pending(hipStream_t waiter, hipEvent_t ev) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return VSTAB_OK;
        if (q != hipErrorNotReady) VSTAB_HIP_TRY(q);
        VSTAB_HIP_TRY(hipStreamWaitEvent(waiter, ev, 0));
        return VSTAB_OK;
    }
    // (the first frame of a stream is tracked from but never warped: its reads were over, host-visibly, when the
    // second frame's LK results came back)
    void forget_borrow(unsigned long serial) {
        for (auto it = borrows.begin(); it != borrows.end(); ++it)
            if (it->serial == serial) {
                borrows.erase(it);
                return;
            }
    }
    std::vector<PendingCopy> copies;  // device-frame copies upstream has not been promised to outlive yet
    std::vector<PendingBorrow> borrows;  // frames used in place whose promise is finite (oldest first)
    int uncovered_borrows = 0;           // of those, warps enqueued but not yet followed by a recorded event
    int src_error = 0;                   // upstream's error code once it has failed (surfaces when the frames read ahead are used up)
    unsigned long ingest_serial = 0;
    long free_counter = 0;
    std::vector<Slot> slots;
    int last_slot = -1;  // m_last_input_frame
    int last_ingest_slot = -1;
    EstimateWorker worker;            // runs estimate_rotation beside the launch calls of the next frame
    bool estimate_posted = false;
    bool threaded_estimate = true;    // VSTAB_THREADED_ESTIMATE=0: estimate on the calling thread
    bool speculate = true;            // VSTAB_SPECULATE=0 disables speculative corner detection
    int cur_pyr = 0;     // pyramid set holding the last tracked frame's pyramid (frame index mod 3)

    long frame_index = 0, last_key = -1;       // m_frame_index, m_last_key_frame_index
    std::vector<float> corners;                // m_last_input_frame_corners
    Mat3 measured = Mat3::identity();          // m_measured_rotation
    bool have_last_rot = false;
    Mat3 last_rot = Mat3::identity();          // m_last_frame_rotation
    std::unique_ptr<RotationFilterSG> sg;      // m_rotation_filter
    RotationFilterKalman kalman;
    std::deque<std::pair<int, Mat3>> queue;    // m_buffered_frames + m_buffered_rotations
    Pcg32 rng;
    std::deque<vstab_frame_log> log;
    std::deque<Mat3> warp_log;
    // the introspection logs keep the most recent LOG_KEEP entries (indices stay absolute)
    static constexpr size_t LOG_KEEP = 1 << 16;
    long log_base = 0, warp_log_base = 0;

    // profiler
    int profiling = 0;  // 0 off, 1 warp launches only (cheap), 2 every GPU stage
    vstab_profile prof{};
    enum Stage { ST_INGEST, ST_PYRAMID, ST_CORNERS, ST_LK, ST_WARP, ST_COUNT };
    struct Pending {
        hipEvent_t a, b;
        int stage;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void fold_pending() {
        if (dstream) (void)hipStreamSynchronize(dstream);
        (void)hipStreamSynchronize(pstream);
        (void)hipStreamSynchronize(tstream);
        (void)hipStreamSynchronize(stream);
        double *sums[ST_COUNT] = {&prof.gpu_ingest_ms, &prof.gpu_pyramid_ms, &prof.gpu_corners_ms, &prof.gpu_lk_ms, &prof.gpu_warp_ms};
        for (auto &p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) *sums[p.stage] += ms;
            event_pool.push_back(p.a), event_pool.push_back(p.b);
        }
        pending.clear();
    }

    // a frame whose tracking has been launched (inflight) / whose LK results have been read (ready)
    struct Tracked {
        int slot = -1;
        vstab_frame_log lg{};
        std::vector<float> prev, pp, cp;
    };
    Tracked inflight, ready, estimating;  // ... and whose rotation estimate is running (or waiting to be computed)
    bool have_inflight = false, have_ready = false, have_estimating = false, src_eof = false;
    // LK launches: the one whose results the host waits for next, and the one chained behind it for the
    // following frame (speculative: valid unless that frame turns out to be a key frame, :415)
    // Tracker launches.  One launch covers a SEGMENT of consecutive frames (k_lk_track: every feature slot runs down its own
    // chain through the segment's frames); segments are enqueued ahead of the frame the host is at, as far as the frames read
    // ahead reach, each chained on the device behind the one before it -- or started from freshly detected corners where the
    // counter half of the key-frame rule (:415) says a key frame will be.  Everything enqueued ahead is speculative: it is
    // dropped when the count half of the rule (< 150 survivors) makes a frame a key frame nobody planned for.
    struct Segment {
        long first = 0;            // frame index (frame_index numbering) of its first frame
        int n = 0;                 // frames covered
        bool key = false;          // starts from fresh corners detected on frame first - 1 (a planned key frame); else chained
        long last_key_after = -1;  // what last_key will be once the host has passed this segment
        int stream = 0;            // epoch stream it was enqueued on (vstab_handle::estream)
        std::vector<float> corners;  // key segments: the corners it was launched with
        Tracker::Launch launch;
    };
    std::deque<Segment> segs;        // launched, not yet used up; consecutive, in frame order; front covers the host's frame
    Tracker::Launch inflight_launch; // the launch, and the frame pair of it, whose results the host waits for next
    int inflight_idx = 0;
    long segs_launched = 0, seg_frames_launched = 0, seg_frames_dropped = 0;
    DevBuf host_out;                // staging buffer of vstab_pull_frame_host
    DevBuf bgr16_out;               // 16-bit BGR frame of vstab_pull_frame_p010 (converted to P010 planes behind the warp)
    // Quantised-map cache: when two consecutive frames are warped with the same 17 parameters (tracking off, or any
    // run of identical rotations) the map is written once (vstab_quantised_map) and the following warps read it
    // instead of evaluating it -- the reference recomputes an identical map per frame (FrameSourceWarp.cpp:283-304).
    DevBuf qmap;
    float qmap_params[17] = {0}, last_params[17] = {0};
    bool qmap_valid = false, have_last_params = false, map_cache = true;  // VSTAB_MAP_CACHE=0 disables
    long warps_from_cache = 0;
    PinnedBuf marker_pts;           // vstab_config.debug: rotating sets of marker centres, read by the kernel in place
    unsigned marker_set = 0;
    hipStream_t dstream = nullptr;  // speculative corner detection (137 us of kernels every 21st frame) beside everything else
    // EPOCHS IN TURN (when dstream exists): what follows a planned key frame -- its speculative detection, the tracker segment launched from
    // those corners and the segments chained behind it -- depends on nothing tracked before it, so it runs on the OTHER of the two streams
    // {tstream, dstream} than the epoch still being tracked: two dependent chains side by side for as long as the read-ahead reaches into
    // the next epoch.  The tracker's chain sets the frame period at 1080p (profiles/r05_epochs_in_turn.txt).  VSTAB_EPOCH_OVERLAP=0 (read
    // once, in vstab_create) keeps everything on tstream.
    bool epoch_overlap = false;
    int epoch_stream = 0;            // stream (0 = tstream, 1 = dstream) of the most recently launched epoch
    int spec_stream = 1;             // stream the pending speculative detection was enqueued on
    int inflight_stream = 0;         // stream of the launch whose results the host waits for next
    long epochs_on_second_stream = 0;
    hipEvent_t epoch_tail = nullptr; // a fresh start waits for what is still queued on the other epoch stream (dropped launches precede their replacement)
    hipStream_t estream(int i) const { return i && dstream ? dstream : tstream; }
    // DMA-BUF objects imported so far (vstab_frame.mem == VSTAB_MEM_DMABUF), keyed by the inode of the object
    DmaBufCache<hipExternalMemory_t> dmabufs;  // vstab_hostlogic.hpp; VSTAB_DMABUF_CACHE=n (tests) shrinks its 256 entries
    bool chain_lk = true;        // VSTAB_CHAIN_LK=0: no launches ahead of the host's frame (one frame per launch, on demand)
    int seg_max = LK_SEG_MAX;    // VSTAB_LK_SEGMENT=n: frames per tracker launch at most (1 = a launch per frame, chained one frame ahead)
    int prefetch_depth = PREFETCH_DEPTH;  // frames pulled from upstream ahead of the one being tracked
    int seg_target = 4;          // a chained segment is enqueued once this many frames are waiting (fewer only at a key frame or when the tracker would idle)
    long chained_adopted = 0, chained_discarded = 0, key_prelaunched = 0;
    // a frame that has been pulled from upstream, copied into the ring and whose pyramid is being built
    std::deque<std::pair<int, int>> prefetched;  // (ring slot, pyramid set), oldest first
    long prefetch_count = 0;

    int acquire_slot() {
        int best = -1;
        for (size_t i = 0; i < slots.size(); i++)
            if (!slots[i].queued && !slots[i].last && (best < 0 || slots[i].freed_at < slots[best].freed_at)) best = (int)i;
        return best;
    }
    const uint8_t *gray(int s) const { return slots[s].y; }
    size_t gpitch(int s) const { return slots[s].pitch_y; }
    int borrow_hold = 0;  // vstab_frame.hold from which a device frame is used in place (set in vstab_create)
    long frames_borrowed = 0, frames_copied = 0;
};

struct GpuStage {  // records an event pair around a stage when profiling is on
    vstab_handle *H;
    hipEvent_t a = nullptr, b = nullptr;
    int stage;
    hipStream_t s;
    GpuStage(vstab_handle *h, int st)
        : H(h), stage(st), s(st == vstab_handle::ST_WARP ? h->stream : (st == vstab_handle::ST_INGEST || st == vstab_handle::ST_PYRAMID) ? h->pstream : h->tstream) {
        // level 1 times every 8th warp launch: two event records cost more host time than the launch itself
        if (H->profiling >= 2 || (H->profiling == 1 && st == vstab_handle::ST_WARP && (H->prof.warp_launches & 7) == 0)) {
            a = H->get_event();
            if (st == vstab_handle::ST_WARP) {
                // the warp launcher stamps the kernel's own start and end into the pair (hipExtLaunchKernelGGL): kernel
                // time as rocprofv3 reports it, without the dispatch wait behind the other streams' kernels
                H->prof.warp_timed++;
                b = H->get_event();
                set_launch_events(a, b);
            } else {
                (void)hipEventRecord(a, s);
            }
        }
    }
    ~GpuStage() {
        if (!a) return;
        if (b && launch_events_pending()) {  // a warp path that does not take the pair (10-bit, direct gather): stream positions
            (void)take_launch_events();
            (void)hipEventRecord(a, s);  // (late: such a launch is then timed as ~0; only the fused kernel is the metric's)
            (void)hipEventRecord(b, s);
        } else if (!b) {
            b = H->get_event();
            (void)hipEventRecord(b, s);
        }
        H->pending.push_back({a, b, stage});
        if (H->pending.size() > 4096) H->fold_pending();
    }
};
// VSTAB_HOST_TIMING=1: wall time of the host-side steps of the pull loop, printed by vstab_destroy (development aid)
struct HostTimers {
    enum { PULL_CB, INGEST_SYNC, INGEST, PYRAMID, SPEC_DETECT, LK_LAUNCH, LK_CHAIN, WARP, TOTAL, N };
    double ms[N] = {0};
    long calls[N] = {0};
    bool on = getenv("VSTAB_HOST_TIMING") != nullptr;
    static const char *name(int i) {
        static const char *n[N] = {"pull_cb", "ingest_sync", "ingest", "pyramid", "spec_detect", "lk_launch", "lk_chain", "warp", "pull_frame_total"};
        return n[i];
    }
};
static HostTimers g_ht;
struct HT {
    int i;
    std::chrono::steady_clock::time_point t0;
    explicit HT(int idx) : i(idx) {
        if (g_ht.on) t0 = std::chrono::steady_clock::now();
    }
    ~HT() {
        if (g_ht.on) g_ht.ms[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), g_ht.calls[i]++;
    }
};

struct HostStage {
    double *sum;
    std::chrono::steady_clock::time_point t0;
    explicit HostStage(double *s) : sum(s), t0(std::chrono::steady_clock::now()) {}
    ~HostStage() { *sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// mem == VSTAB_MEM_DMABUF: the planes are offsets into a DMA-BUF object (an exported decoder surface).  Import the object
// once (objects are recognised by the inode of their fd: decoders hand the same pool of surfaces round and round, under
// fds that may be closed and reused) and rewrite the frame as ordinary device memory.  This is the zero-copy stand-in for
// AvFrameSourceMapOpenCl.cpp:17-66, which moves every frame VAAPI -> host -> OpenCL.
static vstab_status resolve_dmabuf(vstab_handle *H, vstab_frame &f) {
    if (f.mem != VSTAB_MEM_DMABUF) return VSTAB_OK;
    const size_t off_y = reinterpret_cast<uintptr_t>(f.y), off_uv = reinterpret_cast<uintptr_t>(f.uv);
    const size_t bps = f.bit_depth > 8 ? 2 : 1;
    if (f.dmabuf_fd < 0 || f.dmabuf_size == 0 || f.height <= 0 || off_y + f.pitch_y * (size_t)f.height > f.dmabuf_size ||
        off_uv + f.pitch_uv * (size_t)(f.height / 2) > f.dmabuf_size || f.pitch_y < (size_t)f.width * bps || f.pitch_uv < (size_t)f.width * bps)
        return fail(VSTAB_ERR_INVALID, "vstab_frame: DMA-BUF planes do not fit in the object (fd, size, offsets, pitches)");
    // AVDRMObjectDescriptor.format_modifier: the kernels address rows of `pitch` bytes, so a tiled or compressed surface would be
    // read as garbage without any error -- refuse everything but a linear layout (or "no modifier": the exporter's implicit, linear one)
    if (f.dmabuf_modifier != VSTAB_DRM_FORMAT_MOD_LINEAR && f.dmabuf_modifier != VSTAB_DRM_FORMAT_MOD_INVALID) {
        char mod[32];
        std::snprintf(mod, sizeof(mod), "0x%016llx", (unsigned long long)f.dmabuf_modifier);
        return fail(VSTAB_ERR_UNSUPPORTED, std::string("vstab_frame: DMA-BUF with format modifier ") + mod +
                                               " (tiled / compressed surface): only DRM_FORMAT_MOD_LINEAR is read; map the surface linear first");
    }
    struct stat sb;
    if (fstat(f.dmabuf_fd, &sb) != 0) return fail(VSTAB_ERR_INVALID, "vstab_frame: dmabuf_fd is not an open file descriptor");
    // ROCm's import maps the object (the kernel driver takes its own reference on the DMA-BUF) and neither consumes nor closes the
    // descriptor -- unlike CUDA's, which takes ownership.  So the caller's fd is handed over as it is: no duplicate to leak, and the
    // caller may close its fd as soon as the callback returns (test_dmabuf_import_leaves_no_descriptor_behind checks both).
    std::string err;
    auto import = [&](hipExternalMemory_t &ext, uint8_t *&base) {
        hipExternalMemoryHandleDesc hd;
        std::memset(&hd, 0, sizeof(hd));
        hd.type = hipExternalMemoryHandleTypeOpaqueFd, hd.handle.fd = f.dmabuf_fd, hd.size = f.dmabuf_size;
        hipError_t e = hipImportExternalMemory(&ext, &hd);
        if (e != hipSuccess) {
            err = std::string("hipImportExternalMemory(DMA-BUF): ") + hipGetErrorString(e);
            return false;
        }
        hipExternalMemoryBufferDesc bd;
        std::memset(&bd, 0, sizeof(bd));
        bd.offset = 0, bd.size = f.dmabuf_size;
        void *p = nullptr;
        e = hipExternalMemoryGetMappedBuffer(&p, ext, &bd);
        if (e != hipSuccess || !p) {
            (void)hipDestroyExternalMemory(ext);
            err = std::string("hipExternalMemoryGetMappedBuffer(DMA-BUF): ") + hipGetErrorString(e);
            return false;
        }
        base = static_cast<uint8_t *>(p);
        return true;
    };
    auto destroy = [&](hipExternalMemory_t &ext) {
        (void)hipStreamSynchronize(H->pstream);  // (rare: a copy out of the object may only just have been enqueued)
        (void)hipDestroyExternalMemory(ext);
    };
    // a frame stays in the pipeline for at most `slots` pulls (read-ahead + look-ahead window + warp)
    uint8_t *base = nullptr;
    if (!H->dmabufs.lookup((unsigned long long)sb.st_ino, f.dmabuf_size, (long)H->slots.size() + 2, import, destroy, base)) return fail(VSTAB_ERR_DEVICE, err);
    f.y = base + off_y, f.uv = base + off_uv, f.mem = VSTAB_MEM_DEVICE;
    return VSTAB_OK;
}

// pyr: the pyramid set the frame's levels go to; *level1_done: the copy kernel wrote level 1 as well (k_pack_pyr)
static vstab_status ingest(vstab_handle *H, const vstab_frame &f, int slot, int pyr, bool *level1_done) {
    *level1_done = false;
    GpuStage gs(H, vstab_handle::ST_INGEST);
    if (f.width != H->w || f.height != H->h) return fail(VSTAB_ERR_INVALID, "frame size changed mid-stream");
    vstab_handle::Slot &S = H->slots[slot];
    const bool wide = f.bit_depth > 8;
    if (f.bit_depth != 0 && f.bit_depth != 8 && f.bit_depth != 10 && f.bit_depth != 12 && f.bit_depth != 16)
        return fail(VSTAB_ERR_INVALID, "vstab_frame.bit_depth must be 8, 10, 12 or 16");
    if (wide && f.mem != 0) return fail(VSTAB_ERR_INVALID, "16-bit frames must be in device memory");
    if (H->cfg.pixel_depth == 10 && !wide) return fail(VSTAB_ERR_INVALID, "a pixel_depth 10 handle needs P010 device frames (vstab_frame.bit_depth > 8)");
    S.y16 = S.uv16 = nullptr;  // set again below when this frame has 16-bit planes to warp from
    S.copied_valid = false;
    if (f.mem == 0 && !wide && f.hold >= H->borrow_hold && f.pitch_y < (1u << 24) && f.pitch_uv < (1u << 24)) {  // (the kernels form row offsets with 24-bit multiplies)
        // zero copy: track, build the pyramid from and warp upstream's planes where they are
        S.y = static_cast<const uint8_t *>(f.y), S.uv = static_cast<const uint8_t *>(f.uv), S.pitch_y = f.pitch_y, S.pitch_uv = f.pitch_uv;
        S.borrowed = true, S.warp_pending = false, S.warped = -1;
        H->frames_borrowed++;
        return VSTAB_OK;  // (S.ingested keeps its completed state from the slot's previous use; tracking records it behind the pyramid)
    }
    VSTAB_TRY(S.buf.ensure((size_t)H->w * H->h * 3 / 2));
    uint8_t *dst = S.buf.as<uint8_t>();
    S.y = dst, S.uv = dst + (size_t)H->w * H->h, S.pitch_y = S.pitch_uv = (size_t)H->w, S.borrowed = false;
    H->frames_copied++;
    if (S.warp_pending) {  // the warp that last read this slot runs on another stream
        if (S.warped < 0) VSTAB_TRY(H->cover_warps());
        VSTAB_TRY(vstab_handle::wait_if_pending(H->pstream, H->warp_events[S.warped]));
        S.warp_pending = false, S.warped = -1;
    }
    if (wide) {
        VSTAB_TRY(pack_p010_planes(f.y, f.pitch_y, f.uv, f.pitch_uv, f.width, f.height, dst, H->cfg.pixel_depth == 10, H->pstream));
        if (H->cfg.pixel_depth == 10) {  // the warp reads the 16-bit planes; the tracker the narrowed luma above
            const size_t row = (size_t)H->w * 2;
            if (f.hold >= vstab_handle::HOLD_FOREVER) {
                S.y16 = static_cast<const uint8_t *>(f.y), S.uv16 = static_cast<const uint8_t *>(f.uv), S.pitch_y16 = f.pitch_y, S.pitch_uv16 = f.pitch_uv;
            } else {
                VSTAB_TRY(S.buf16.ensure(row * H->h * 3 / 2));
                uint8_t *d16 = S.buf16.as<uint8_t>();
                VSTAB_HIP_TRY(hipMemcpy2DAsync(d16, row, f.y, f.pitch_y, row, H->h, hipMemcpyDeviceToDevice, H->pstream));
                VSTAB_HIP_TRY(hipMemcpy2DAsync(d16 + row * H->h, row, f.uv, f.pitch_uv, row, H->h / 2, hipMemcpyDeviceToDevice, H->pstream));
                S.y16 = d16, S.uv16 = d16 + row * H->h, S.pitch_y16 = S.pitch_uv16 = row;
            }
        }
        // upstream's surface is free once these copies are through (not the pyramid behind them: see pack_nv12_planes) -- an event of its own,
        // but only for a frame somebody will wait for: a marker packet per frame costs the read-ahead stream what it saves the caller
        if (H->cfg.tracking && f.hold < (int)H->slots.size()) {
            VSTAB_HIP_TRY(hipEventRecord(S.copied, H->pstream));
            S.copied_valid = true;
        }
    } else if (f.mem == 0) {
        // (while tracking, `ingested` completes with the pyramid enqueued behind the copy: upstream's surface is free as soon as the copy is)
        S.copied_valid = H->cfg.tracking != 0;
        static const bool fuse = getenv("VSTAB_PACK_PYR") == nullptr || atoi(getenv("VSTAB_PACK_PYR")) != 0;  // development: =0 copies and builds level 1 in two launches
        uint8_t *l1 = H->cfg.tracking ? H->tracker.level1(pyr) : nullptr;
        if (fuse && l1 && pack_pyr_ok(f.y, f.pitch_y, f.uv, f.pitch_uv, f.width, f.height, dst, l1, H->tracker.level1_pitch())) {
            // one pass over the luma plane: the copy into the ring and the first pyramid level (k_pack_pyr)
            VSTAB_TRY(launch_pack_pyr(static_cast<const uint8_t *>(f.y), f.pitch_y, static_cast<const uint8_t *>(f.uv), f.pitch_uv, f.width, f.height, dst, l1,
                                      H->tracker.level1_pitch(), H->pstream, S.copied));
            *level1_done = true;
        } else
        VSTAB_TRY(pack_nv12_planes(f.y, f.pitch_y, f.uv, f.pitch_uv, f.width, f.height, dst, H->pstream, S.copied_valid ? S.copied : nullptr));
    } else {
        VSTAB_HIP_TRY(hipMemcpy2DAsync(dst, H->w, f.y, f.pitch_y, H->w, H->h, hipMemcpyHostToDevice, H->pstream));
        VSTAB_HIP_TRY(hipMemcpy2DAsync(dst + (size_t)H->w * H->h, H->w, f.uv, f.pitch_uv, H->w, H->h / 2, hipMemcpyHostToDevice, H->pstream));
        VSTAB_HIP_TRY(hipStreamSynchronize(H->pstream));  // the caller may reuse its host buffer on return
    }
    if (!H->cfg.tracking) VSTAB_HIP_TRY(hipEventRecord(S.ingested, H->pstream));  // tracking: recorded behind the pyramid instead
    return VSTAB_OK;
}

// ---------------------------------------------------------------------------------------------
// consume_frame (FrameSourceWarp.cpp:397-450) split into steps so that the copy + pyramid of the frames read ahead,
// the (chained) LK tracking of frames k+1 and k+2 and the speculative corner detection run on the GPU while the host
// estimates the rotation of frame k (PREFETCH_DEPTH frames of upstream read-ahead; four HIP streams ordered by events;
// DESIGN.md section 5b):
//   prefetch_next    pull the next upstream frame, copy it into the ring unless upstream holds it, build its pyramid
//                    (pstream); start the speculative corner detection when the counter says a key frame is coming
//   launch_tracking  key-frame rule (:415-419); adopt the launch chained behind the previous frame's tracker or launch
//                    one; chain the next frame's tracker (tstream)
//   finish_wait      LK results -> surviving pairs (:422-427)
//   post_estimate / finish_estimate  rotation (worker thread) + fallback + accumulation + filter.add + queue push (:429-446)
// Every step runs in frame order, so every decision, random draw and queue entry is the one the
// reference makes; only WHEN the upstream callback is called moves (up to PREFETCH_DEPTH + 1 frames earlier).
// ---------------------------------------------------------------------------------------------
// prefetch_next: upstream callback, copy into the ring, pyramid -- all on the prefetch stream, with no
// dependence on the tracking state, so it overlaps the LK kernel of the previous frame.
static vstab_status prefetch_next(vstab_handle *H) {
    vstab_frame f;
    std::memset(&f, 0, sizeof(f));
    {
        // copies of frames whose planes upstream may recycle on this call must have finished (vstab_frame.hold)
        HT t(HostTimers::INGEST_SYNC);
        for (auto it = H->copies.begin(); it != H->copies.end();) {
            if (it->hold > 0) {
                it->hold--, ++it;
                continue;
            }
            const vstab_handle::Slot &S = H->slots[it->slot];
            if (S.ingest_serial == it->serial)  // (a re-used slot's newer copy was enqueued behind this one: also done)
                VSTAB_TRY(vstab_handle::host_wait(S.copied_valid ? S.copied : S.ingested));
            it = H->copies.erase(it);
        }
        // frames used in place: the warp reading them must be through before the callback that ends upstream's promise
        for (auto it = H->borrows.begin(); it != H->borrows.end();) {
            if (it->hold > 0) {
                it->hold--, ++it;
                continue;
            }
            if (!it->warp_enqueued)
                return fail(VSTAB_ERR_INVALID, "vstab_frame.hold ran out while the frame was still waiting in the look-ahead window");
            if (it->warped < 0) VSTAB_TRY(H->cover_warps());
            VSTAB_TRY(vstab_handle::host_wait(H->warp_events[it->warped]));
            it = H->borrows.erase(it);
        }
    }
    int rc;
    {
        HT t(HostTimers::PULL_CB);
        rc = H->src.pull(H->src.user, &f);
    }
    if (rc == VSTAB_EOF) {
        H->src_eof = true;
        return VSTAB_EOF;
    }
    if (rc != 0) {
        // The reference meets this error only when it consumes the failing frame (FrameSourceWarp.cpp:453-455), after it
        // has emitted every frame whose look-ahead window was complete; the frames read ahead here are still used.
        H->src_eof = true, H->src_error = rc;
        return VSTAB_EOF;
    }
    VSTAB_TRY(resolve_dmabuf(H, f));  // a DMA-BUF frame becomes an ordinary device frame here
    const int slot = H->acquire_slot();
    if (slot < 0) return fail(VSTAB_ERR_NOMEM, "look-ahead ring exhausted");
    const int pyr = (int)(H->prefetch_count % PYR_SETS);
    bool level1_done = false;
    {
        HT t(HostTimers::INGEST);
        VSTAB_TRY(ingest(H, f, slot, pyr, &level1_done));
    }
    H->last_ingest_slot = slot;
    H->slots[slot].have_delta = f.delta_rotation != nullptr;
    if (f.delta_rotation) std::memcpy(H->slots[slot].delta.m, f.delta_rotation, sizeof(double) * 9);
    H->slots[slot].have_readout = f.readout_rotation != nullptr;
    if (f.readout_rotation) {
        if (H->cfg.pixel_depth != 10 && H->map_mode != VSTAB_MAP_CREATEMAP_CL && H->map_mode != VSTAB_MAP_FISH_TO_RECT &&
            H->map_mode != VSTAB_MAP_CREATEMAP_CL_OPENCL)
            return fail(VSTAB_ERR_INVALID, "vstab_frame.readout_rotation: the 8-bit rolling-shutter warp exists for the preset and fisheye -> rectilinear maps only");
        std::memcpy(H->slots[slot].readout.m, f.readout_rotation, sizeof(double) * 9);
    }
    H->slots[slot].ingest_serial = ++H->ingest_serial;
    // (frames promised to outlive a whole ring of pulls are not tracked: the ring slot itself is recycled sooner)
    if (f.mem == 0 && !H->slots[slot].borrowed && f.hold < (int)H->slots.size())
        H->copies.push_back({slot, H->ingest_serial, f.hold < 0 ? 0 : f.hold});
    if (H->slots[slot].borrowed && f.hold < vstab_handle::HOLD_FOREVER) H->borrows.push_back({H->ingest_serial, f.hold, false, -1});
    H->slots[slot].queued = true;  // reserved from now on (released when its warp has been enqueued)
    if (H->cfg.tracking) {
        HT t(HostTimers::PYRAMID);
        GpuStage gs(H, vstab_handle::ST_PYRAMID);
        // `ingested` = copy AND pyramid of this frame: the event completes with the pyramid's last kernel (no marker packet on the stream)
        static const bool bind_event = getenv("VSTAB_PYR_EVENT_RECORD") == nullptr;  // development: =1 records the event behind the kernels instead
        bool bound = false;
        VSTAB_TRY(H->tracker.build_pyramid(pyr, H->gray(slot), H->gpitch(slot), H->pstream, bind_event ? H->slots[slot].ingested : nullptr, &bound, level1_done));
        if (!bound) VSTAB_HIP_TRY(hipEventRecord(H->slots[slot].ingested, H->pstream));
    }
    // Key-frame rule, counter half (:415): the frame after this one re-detects corners on THIS frame when
    // (index + 1) - last_key > 20.  That is known now, so the detector runs here, on the prefetch stream,
    // a whole frame period before its result is needed; launch_tracking falls back to detecting on demand
    // if the prediction turns out wrong (an extra key frame in between) or the candidates overflow.
    // (The rule has not been evaluated yet for the frames still in the read-ahead window; "== 21" is the
    // one frame for which it is false for every pending frame and true for the next.)
    if (H->cfg.tracking && H->speculate && H->last_key != -1 && (H->prefetch_count + 1) - H->last_key == 21)
    {
        if (debug_spec()) std::fprintf(stderr, "spec launch for frame %ld (last_key %ld)\n", H->prefetch_count, H->last_key);
        HT t(HostTimers::SPEC_DETECT);
        hipStream_t ds = H->dstream ? H->dstream : H->pstream;
        if (H->epoch_overlap) H->spec_stream = 1 - H->epoch_stream, ds = H->estream(H->spec_stream);  // the epoch after the newest one enqueued
        // the detector reads the frame's luma plane, nothing else: it waits for the copy into the ring (if there was one), not for the
        // pyramid enqueued behind it -- beside a saturating warp the detection needs most of the read-ahead's lead as it is
        if (ds != H->pstream) {
            const vstab_handle::Slot &DS = H->slots[slot];
            if (!DS.borrowed) VSTAB_HIP_TRY(hipStreamWaitEvent(ds, DS.copied_valid ? DS.copied : DS.ingested, 0));
        }
        VSTAB_TRY(H->tracker.spec_launch(H->gray(slot), H->gpitch(slot), 0.01, ds, H->prefetch_count));
        H->tracker.spec_select_async(200, 30.0);
    }
    H->prefetched.emplace_back(slot, pyr);
    H->prefetch_count++;
    return VSTAB_OK;
}

// launch_tracking: key-frame rule (:415-419) and the LK launch for the prefetched frame.  Needs the
// surviving corners of the previous frame (finish_wait), i.e. runs in frame order.
static vstab_status launch_tracking(vstab_handle *H) {
    const int slot = H->prefetched.front().first, pyr = H->prefetched.front().second;
    H->prefetched.pop_front();
    const size_t pitch = H->gpitch(slot);
    const uint8_t *g = H->gray(slot);
    if (!H->cfg.tracking) {
        // undistort-only mode (BASELINE config 1): every frame gets the identity rotation
        if (H->last_key == -1) {
            H->last_key = H->frame_index;
            H->slots[slot].queued = false;  // the first frame is never emitted (:403-407)
            H->forget_borrow(H->slots[slot].ingest_serial);
        } else {
            if (H->slots[slot].have_delta) H->measured = H->slots[slot].delta * H->measured;  // :441 with the sensor's rotation
            if (H->sg) H->sg->add(H->measured);
            H->queue.emplace_back(slot, H->measured);
        }
    } else if (H->last_key == -1) {
        // :403-407 the first frame only seeds the corner set
        H->last_key = H->frame_index;
        VSTAB_HIP_TRY(hipStreamWaitEvent(H->tstream, H->slots[slot].ingested, 0));
        {
            HostStage hs(&H->prof.host_corners_ms);
            VSTAB_TRY(H->tracker.good_features(g, pitch, 200, 0.01, 30.0, H->corners, H->tstream));
        }
        H->prof.key_frames++;
        H->slots[slot].queued = false;
        H->forget_borrow(H->slots[slot].ingest_serial);
    } else {
        vstab_handle::Tracked &T = H->inflight;
        T = vstab_handle::Tracked();
        T.slot = slot;
        const long F = H->frame_index;
        const uint8_t *pg = H->gray(H->last_slot);
        const size_t ppitch = H->gpitch(H->last_slot);
        // pyramid of a frame at or after F - 1 (all of them are in the ring: F - 1 is the last tracked frame, F the one popped
        // above, the following ones wait in `prefetched`)
        auto pyr_of = [&](long fr) {
            if (fr == F - 1) return H->tracker.pyramid(H->cur_pyr, pg, ppitch);
            if (fr == F) return H->tracker.pyramid(pyr, g, pitch);
            const auto &pf = H->prefetched[(size_t)(fr - F - 1)];
            return H->tracker.pyramid(pf.second, H->gray(pf.first), H->gpitch(pf.first));
        };
        auto slot_of = [&](long fr) { return fr == F ? slot : H->prefetched[(size_t)(fr - F - 1)].first; };
        // the last frame read ahead whose copy + pyramid (prefetch stream) have COMPLETED, as far as the host can see without
        // waiting: a launch may only start once the pyramid of its last frame exists, so a segment that reached for the
        // frame pulled a moment ago would hold all its earlier frames back (measured: segments of 8 were slower than of 1)
        long reach = F;
        for (size_t j = 0; j < H->prefetched.size(); j++) {
            if (hipEventQuery(H->slots[H->prefetched[j].first].ingested) != hipSuccess) {
                (void)hipGetLastError();  // "not ready" is an answer, not an error the next launch check should find
                break;
            }
            reach = F + 1 + (long)j;
        }
        auto drop_segments = [&]() {
            for (const auto &sg : H->segs) H->seg_frames_dropped += sg.first + sg.n - std::max(sg.first, F);
            H->segs.clear();
        };
        // enqueue a segment of n frames from `first` on: from `start` (host points) or chained behind `parent`
        auto launch_segment = [&](long first, int n, bool key, const std::vector<float> *start, const Tracker::Launch *parent, long last_key_after,
                                  bool timed, int stream_idx) -> vstab_status {
            LkPyramid pyrs[LK_SEG_MAX + 1];
            for (int i = 0; i <= n; i++) pyrs[i] = pyr_of(first - 1 + i);
            const hipStream_t es = H->estream(stream_idx);
            // copy + pyramid of the segment's frames: they are enqueued in frame order on the prefetch stream, the last one covers all
            VSTAB_TRY(vstab_handle::wait_if_pending(es, H->slots[slot_of(first + n - 1)].ingested));
            vstab_handle::Segment sg;
            sg.first = first, sg.n = n, sg.key = key, sg.last_key_after = last_key_after, sg.stream = stream_idx;
            if (start) {
                if (key) sg.corners = *start;
                VSTAB_TRY(H->tracker.track_launch(pyrs, n, *start, es, timed, sg.launch));
            } else {
                VSTAB_TRY(H->tracker.track_launch_chained(pyrs, n, *parent, es, sg.launch));  // (behind its parent: the same stream)
            }
            H->segs_launched++, H->seg_frames_launched += n;
            H->segs.push_back(std::move(sg));
            return VSTAB_OK;
        };
        while (!H->segs.empty() && H->segs.front().first + H->segs.front().n <= F) H->segs.pop_front();  // used up
        // :415-419 key-frame rule; corners are found in the PREVIOUS gray frame
        const bool is_key = F - H->last_key > 20 || H->corners.size() / 2 < 150;
        const vstab_handle::Segment *front = H->segs.empty() ? nullptr : &H->segs.front();
        // what was enqueued for this frame stands if it made the same decision: a key segment starting here for a key frame,
        // the inside of a segment (or the start of a chained one) for an ordinary frame
        const bool covered = front && front->first <= F;
        const bool planned_key = covered && front->key && front->first == F;
        bool adopt = covered && planned_key == is_key;
        if (covered && !adopt) H->chained_discarded++;
        if (is_key) {
            H->last_key = F - 1;
            HostStage hs(&H->prof.host_corners_ms);
            if (adopt) {
                H->corners = front->corners;  // this key frame's tracker is already running on them
                H->key_prelaunched++;
            } else {
                // the previous frame is F - 1: use its speculative detection if there is one
                const bool spec = H->tracker.spec_tag() == F - 1 && H->tracker.spec_finish(200, 30.0, H->corners);
                if (debug_spec()) std::fprintf(stderr, "key frame at %ld: spec_tag %ld used %d\n", F, H->tracker.spec_tag(), (int)spec);
                if (!spec) VSTAB_TRY(H->tracker.good_features(pg, ppitch, 200, 0.01, 30.0, H->corners, H->tstream));
            }
            T.lg.key_frame = 1;
            H->prof.key_frames++;
        }
        T.lg.n_corners = (int)(H->corners.size() / 2);
        T.prev = H->corners;
        // how many frames a launch may cover: the whole read-ahead when launches are chained ahead, one frame otherwise
        const bool ahead = H->chain_lk && H->profiling < 2;
        const int seg_max = ahead ? H->seg_max : 1;
        if (adopt) {
            H->chained_adopted++;
        } else {
            // everything enqueued ahead assumed another course of events (or nothing was enqueued): start afresh from the host's
            // corner list.  Dropped launches may still be running; this one queues behind them on the tracker stream.
            drop_segments();
            HT t(HostTimers::LK_LAUNCH);
            const long kc = H->last_key + 21;  // the next frame the counter makes a key frame
            const int n = (int)std::max<long>(1, std::min<long>({(long)seg_max, reach - F + 1, kc - F}));
            if (H->epoch_overlap) {  // whatever is still queued on the other epoch stream was dropped a moment ago (or is long finished): behind it
                VSTAB_HIP_TRY(hipEventRecord(H->epoch_tail, H->dstream));
                VSTAB_HIP_TRY(hipStreamWaitEvent(H->tstream, H->epoch_tail, 0));
            }
            H->epoch_stream = 0;
            VSTAB_TRY(launch_segment(F, n, false, &H->corners, nullptr, H->last_key, H->profiling >= 2, 0));
        }
        H->inflight_launch = H->segs.front().launch, H->inflight_idx = (int)(F - H->segs.front().first), H->inflight_stream = H->segs.front().stream;
        H->have_inflight = true;
        // Enqueue further segments as far as the read-ahead reaches: chained behind the last one up to the next key frame the
        // counter half of the rule (:415) predicts, and -- once that key frame's corners (detected speculatively on the frame before
        // it) are selected -- a segment from those corners, which does not depend on any earlier tracking at all.  The count half
        // (< 150 survivors) is checked when a frame's turn comes; if it fires, what was enqueued beyond is dropped (above).
        while (ahead) {
            HT t(HostTimers::LK_CHAIN);
            const vstab_handle::Segment &back = H->segs.back();
            const long tail = back.first + back.n - 1, next = tail + 1, avail = reach - tail;
            if (avail <= 0 || back.launch.n_slots == 0) break;
            const long kc = back.last_key_after + 21;  // next planned key frame
            if (next == kc) {
                if (H->tracker.spec_tag() == tail) H->tracker.spec_poll_inline();
                if (H->tracker.spec_tag() != tail || H->tracker.spec_state() != 2) {
                    if (debug_spec() && H->tracker.spec_tag() == tail)
                        std::fprintf(stderr, "frame %ld: corners for key frame %ld not selected yet (state %d)\n", F, next, H->tracker.spec_state());
                    break;  // not detected / selected yet: next pull, or on demand when the host gets there
                }
                std::vector<float> fresh;
                H->tracker.spec_take(fresh);
                const int n = (int)std::min<long>({(long)seg_max, avail, 21});
                const int es = H->epoch_overlap ? H->spec_stream : 0;  // where its detection ran: not the stream of the epoch before it
                VSTAB_TRY(launch_segment(next, n, true, &fresh, nullptr, next - 1, false, es));
                H->epoch_stream = es;
                if (es == 1) H->epochs_on_second_stream++;
                continue;
            }
            const int n = (int)std::min<long>({(long)seg_max, avail, kc - next});
            // full segments; shorter ones only up to a key frame, or when nothing is enqueued beyond the host's frame
            if (n < H->seg_target && next + n != kc && tail > F) break;
            VSTAB_TRY(launch_segment(next, n, false, nullptr, &back.launch, back.last_key_after, false, back.stream));
        }
    }
    H->cur_pyr = pyr;
    H->prof.frames_consumed++;
    if (H->last_slot >= 0) {
        H->slots[H->last_slot].last = false;
        if (!H->slots[H->last_slot].queued) H->slots[H->last_slot].freed_at = ++H->free_counter;
    }
    H->slots[slot].last = true;
    H->last_slot = slot;  // :448
    ++H->frame_index;     // :449
    return VSTAB_OK;
}

static vstab_status finish_wait(vstab_handle *H) {
    if (!H->have_inflight) return VSTAB_OK;
    vstab_handle::Tracked &T = H->inflight;
    std::vector<float> nxt;
    std::vector<uint8_t> st;
    {
        HostStage hs(&H->prof.host_track_wait_ms);
        VSTAB_TRY(H->tracker.track_wait(H->inflight_launch, H->inflight_idx, T.prev.size() / 2, nxt, st, H->estream(H->inflight_stream),
                                        H->profiling >= 2 ? &H->prof.gpu_lk_ms : nullptr));
    }
    // :261-268 keep pairs with status != 0
    for (size_t i = 0; i < st.size(); i++)
        if (st[i]) {
            T.pp.push_back(T.prev[2 * i]), T.pp.push_back(T.prev[2 * i + 1]);
            T.cp.push_back(nxt[2 * i]), T.cp.push_back(nxt[2 * i + 1]);
        }
    H->corners = T.cp;  // :427
    T.lg.n_tracked = (int)(T.cp.size() / 2);
    H->ready = std::move(T);
    H->have_ready = true, H->have_inflight = false;
    return VSTAB_OK;
}

// start the rotation estimate of the ready frame on the worker thread (:429-431).  The frame moves on to `estimating`, so
// that the next frame's LK results can be read while this estimate runs: the worker gets a whole frame period (the launches
// of the following frames, the warp of the emitted one, the caller's own code) instead of the few microseconds between two
// steps of one call.  Estimates are still computed and applied strictly in frame order (one at a time: the random stream,
// the < 40 inlier fallback and the accumulation of :441 are sequential).
static void post_estimate(vstab_handle *H) {
    if (!H->have_ready || H->have_estimating) return;
    H->estimating = std::move(H->ready);
    H->have_ready = false, H->have_estimating = true;
    if (!H->threaded_estimate) return;  // computed by finish_estimate on the calling thread
    vstab_handle::Tracked &T = H->estimating;
    H->worker.post(T.pp.data(), T.cp.data(), T.lg.n_tracked, &H->Kin, &H->Kout, &H->rng, H->in_fish);
    H->estimate_posted = true;
}

static void finish_estimate(vstab_handle *H) {
    if (!H->have_estimating) return;
    vstab_handle::Tracked &T = H->estimating;
    vstab_frame_log &lg = T.lg;
    // :429-438 rotation since the last frame, with the < 40 inlier fallback
    Mat3 R;
    int inl;
    {
        HostStage hs(&H->prof.host_estimate_ms);  // threaded: only the time the caller still had to wait
        if (H->estimate_posted)
            inl = H->worker.join(R), H->estimate_posted = false;
        else
            inl = estimate_rotation(T.pp.data(), T.cp.data(), lg.n_tracked, H->Kin, H->Kout, H->rng, R, H->in_fish);
    }
    lg.n_inliers = inl;
    if (inl < 40) {
        R = H->have_last_rot ? H->last_rot : Mat3::identity();
        lg.fallback = 1;
    }
    H->last_rot = R, H->have_last_rot = true;
    H->measured = R * H->measured;  // :441 left-multiplied accumulation
    if (H->sg) H->sg->add(H->measured);
    H->queue.emplace_back(T.slot, H->measured);
    if (H->cfg.debug) H->slots[T.slot].feats = T.cp;
    std::memcpy(lg.R_frame, R.m, sizeof(R.m));
    std::memcpy(lg.R_accum, H->measured.m, sizeof(R.m));
    H->log.push_back(lg);
    if (H->log.size() > vstab_handle::LOG_KEEP) H->log.pop_front(), H->log_base++;
    H->have_estimating = false;
}

extern "C" {

int vstab_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(vstab_frame);
        case 1: return (int)sizeof(vstab_source);
        case 2: return (int)sizeof(vstab_config);
        case 3: return (int)sizeof(vstab_frame_log);
        case 4: return (int)sizeof(vstab_profile);
        default: return -1;
    }
}

int vstab_abi_version(void) { return VSTAB_ABI_VERSION; }

void vstab_config_default(vstab_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = VSTAB_ABI_VERSION;
    cfg->preset = VSTAB_GOPRO_H4B_WIDE169_MEASURED;
    cfg->scale = 1, cfg->crop_borders = 0, cfg->zoom = 1, cfg->smooth_radius = 30;  // FrameSourceWarp.hpp:86-89
    cfg->interpolation = 1, cfg->smoother = VSTAB_SMOOTHER_SG, cfg->tracking = 1, cfg->seed = 1, cfg->stream = nullptr;
    cfg->lens_mode = 0, cfg->in_projection = VSTAB_PROJ_FISH, cfg->out_projection = VSTAB_PROJ_RECT;
    cfg->in_dfov = 0, cfg->out_dfov = 0, cfg->out_width = 0, cfg->out_height = 0, cfg->out_cx = -1, cfg->out_cy = -1, cfg->debug = 0;
    cfg->pixel_depth = 8, cfg->blend = VSTAB_BLEND_EXACT;
    // the reference's map is what ITS kernel computes on this GPU (createMap.cl through ROCm's OpenCL compiler): the default
    cfg->map_precision = VSTAB_MAP_PRECISION_OPENCL;
    cfg->read_ahead = 0;  // the library's default (PREFETCH_DEPTH)
}

vstab_status vstab_preload_kernels(void) {
    VSTAB_TRY(preload_track_kernels());
    VSTAB_TRY(preload_warp_kernels());
    VSTAB_TRY(preload_fused_kernels());
    VSTAB_TRY(preload_p010_kernels());
    VSTAB_TRY(preload_planar_kernels());
    return VSTAB_OK;
}

vstab_status vstab_create(const vstab_config *cfg, const vstab_source *src, vstab_handle **out) {
    if (!cfg || !src || !out || !src->pull || !src->peek) return fail(VSTAB_ERR_INVALID, "vstab_create: null argument");
    if (cfg->abi_version != VSTAB_ABI_VERSION)
        return fail(VSTAB_ERR_INVALID, "vstab_create: vstab_config.abi_version is " + std::to_string(cfg->abi_version) + ", this library is version " +
                                           std::to_string(VSTAB_ABI_VERSION) + ": initialise the struct with vstab_config_default() of THIS library (include/vstab.h)");
    if (cfg->smooth_radius < 0 || cfg->smooth_radius > 10000) return fail(VSTAB_ERR_INVALID, "vstab_create: bad smooth_radius");
    if (cfg->interpolation != 1 && cfg->interpolation != 0)
        return fail(VSTAB_ERR_INVALID, "vstab_create: interpolation must be INTER_LINEAR (1, the only mode the reference passes) or INTER_NEAREST (0)");
    if (cfg->interpolation == 0 && (cfg->lens_mode != 0 || cfg->pixel_depth == 10))
        return fail(VSTAB_ERR_INVALID, "vstab_create: INTER_NEAREST exists for the reference's own map (lens_mode 0, 8-bit pixels)");
    if (!(cfg->scale > 0) || !(cfg->zoom > 0)) return fail(VSTAB_ERR_INVALID, "vstab_create: scale and zoom must be positive");
    if (cfg->smoother < VSTAB_SMOOTHER_SG || cfg->smoother > VSTAB_SMOOTHER_FIXED) return fail(VSTAB_ERR_INVALID, "vstab_create: unknown smoother");
    if (cfg->lens_mode != 0 && cfg->lens_mode != 1) return fail(VSTAB_ERR_INVALID, "vstab_create: lens_mode must be 0 or 1");
    if (cfg->pixel_depth != 0 && cfg->pixel_depth != 8 && cfg->pixel_depth != 10) return fail(VSTAB_ERR_INVALID, "vstab_create: pixel_depth must be 8 or 10");
    if (cfg->blend != VSTAB_BLEND_EXACT && cfg->blend != VSTAB_BLEND_FP16) return fail(VSTAB_ERR_INVALID, "vstab_create: unknown blend");
    if (cfg->map_precision != VSTAB_MAP_PRECISION_IEEE && cfg->map_precision != VSTAB_MAP_PRECISION_OPENCL)
        return fail(VSTAB_ERR_INVALID, "vstab_create: unknown map_precision");
    if (cfg->read_ahead < 0 || cfg->read_ahead > PREFETCH_MAX)
        return fail(VSTAB_ERR_INVALID, "vstab_create: read_ahead must be 0 (the default, " + std::to_string(PREFETCH_DEPTH) + ") or 1 .. " + std::to_string(PREFETCH_MAX));
    // (lens_mode 1 ignores map_precision: those maps are this library's own definitions, IEEE arithmetic throughout)
    std::unique_ptr<vstab_handle> H(new vstab_handle);
    H->cfg = *cfg, H->src = *src;
    H->rng = Pcg32(cfg->seed);
    if (const char *e = getenv("VSTAB_SPECULATE")) H->speculate = atoi(e) != 0;
    if (const char *e = getenv("VSTAB_THREADED_ESTIMATE")) H->threaded_estimate = atoi(e) != 0;
    if (const char *e = getenv("VSTAB_CHAIN_LK")) H->chain_lk = atoi(e) != 0;
    if (const char *e = getenv("VSTAB_LK_SEGMENT")) H->seg_max = std::max(1, std::min(atoi(e), LK_SEG_MAX));
    H->seg_target = std::min(H->seg_target, H->seg_max);
    if (const char *e = getenv("VSTAB_MAP_CACHE")) H->map_cache = atoi(e) != 0;
    if (const char *e = getenv("VSTAB_DMABUF_CACHE")) H->dmabufs.cap = std::max(1, atoi(e));
    H->stream = static_cast<hipStream_t>(cfg->stream);  // NULL = the default stream, as for the stateless operators
    {
        // the tracking chain is the per-frame critical path; the warp only has to finish before the
        // caller looks at dst, so the tracking stream gets the highest priority the device offers
        // every code object of the library now, not at the first launch of one of its kernels in the middle of the frame loop (and not after
        // whatever else the process has loaded and unloaded by then: vstab.h, vstab_preload_kernels)
        VSTAB_TRY(vstab_preload_kernels());
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        VSTAB_HIP_TRY(hipStreamCreateWithPriority(&H->tstream, hipStreamNonBlocking, hi));
        // copy + pyramid of the NEXT frame have a whole frame period of slack: lowest priority, so they fill
        // in behind the warp instead of taking its CUs
        VSTAB_HIP_TRY(hipStreamCreateWithPriority(&H->pstream, hipStreamNonBlocking, lo));
        // (this runtime offers two priority levels; beside a saturating warp the detection takes ~500 us at either)
        // The runtime multiplexes its streams onto four hardware queues, and two streams that share one serialise (a fifth
        // stream cost the 4K pipeline 4 k frames/s merely by existing): caller + tracker + prefetch leave ONE more: the
        // speculative corner detection's.
        // A caller that hands over a stream of its own very likely has the default stream in the process as well (any synchronous copy
        // uses it): with the detection's stream that makes five, and the fifth costs 15 % of the 4K rate and 10 % at 1080p, where queueing
        // the detection on the read-ahead stream costs 0.7 % and 3.5 % (profiles/r04_stream_count_ab.txt).  So the detection gets a stream
        // of its own only beside a caller on the default stream; VSTAB_DETECT_STREAM=1 / 0 overrides.
        const char *ds_env = getenv("VSTAB_DETECT_STREAM");
        const bool detect_stream = ds_env ? atoi(ds_env) != 0 : H->stream == nullptr;
        // (epochs in turn, below: the stream keeps the low priority -- at the tracker's priority the 4K rate lost 1.5 %, the 1080p rate gained nothing)
        if (detect_stream) VSTAB_HIP_TRY(hipStreamCreateWithPriority(&H->dstream, hipStreamNonBlocking, lo));
        for (auto &e : H->warp_events) VSTAB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // :214-219 peek the first frame for the input size, then derive both cameras
    vstab_frame f;
    std::memset(&f, 0, sizeof(f));
    const int rc = src->peek(src->user, &f);
    if (rc == VSTAB_EOF) return fail(VSTAB_EOF, "vstab_create: upstream has no frames");
    if (rc != 0) return fail(VSTAB_ERR_SOURCE, "vstab_create: upstream peek failed with " + std::to_string(rc));
    if (f.width <= 0 || f.height <= 0 || (f.width & 1) || (f.height & 1) || f.width > 32767 || f.height > 32767)
        return fail(VSTAB_ERR_INVALID, "vstab_create: frame size must be even and <= 32767");
    H->w = f.width, H->h = f.height;
    {
        // Epochs in turn (vstab_handle::epoch_overlap) pay where the tracker's dependent chain is longer than a frame's warp: 1080p + 10 - 13 %
        // (44.3 -> 49.5 k frames/s); at 4K the warp sets the period and a second tracker chain beside it costs 0.5 %.  Bit-identical either way.
        // VSTAB_EPOCH_OVERLAP=1 / 0 (development) overrides the size rule.
        const char *const eo = getenv("VSTAB_EPOCH_OVERLAP");  // (read here, on the caller's thread, like the other launch-shape switches)
        const int eo_forced = eo ? (atoi(eo) != 0 ? 1 : 0) : -1;
        const bool small_frame = (long)H->w * H->h <= 1920L * 1200;
        H->epoch_overlap = H->dstream && cfg->tracking && (eo_forced < 0 ? small_frame : eo_forced == 1);
        if (H->epoch_overlap) VSTAB_HIP_TRY(hipEventCreateWithFlags(&H->epoch_tail, hipEventDisableTiming));
    }
    if (cfg->lens_mode == 0) {
        if (!preset_camera(cfg->preset, H->w, H->h, H->Kin)) return fail(VSTAB_ERR_INVALID, "vstab_create: unknown preset");
        output_camera(H->Kin, H->w, H->h, cfg->scale, cfg->crop_borders != 0, cfg->zoom, H->Kout, H->ow, H->oh);
        if (cfg->map_precision == VSTAB_MAP_PRECISION_OPENCL) H->map_mode = VSTAB_MAP_CREATEMAP_CL_OPENCL;
    } else {
        // the libdewobble filter options as the CLI sets them (render.ts:669-683)
        H->ow = cfg->out_width > 0 ? cfg->out_width : H->w, H->oh = cfg->out_height > 0 ? cfg->out_height : H->h;
        const double out_dfov = cfg->out_dfov > 0 ? cfg->out_dfov : cfg->in_dfov;
        if (!lens_camera(cfg->in_projection, cfg->in_dfov, H->w, H->h, -1, -1, H->Kin) ||
            !lens_camera(cfg->out_projection, out_dfov, H->ow, H->oh, cfg->out_cx, cfg->out_cy, H->Kout))
            return fail(VSTAB_ERR_INVALID, "vstab_create: bad lens description (projection / field of view / size)");
        H->in_fish = cfg->in_projection == VSTAB_PROJ_FISH;
        const bool out_fish = cfg->out_projection == VSTAB_PROJ_FISH;
        H->map_mode = H->in_fish ? (out_fish ? VSTAB_MAP_FISH_TO_FISH : VSTAB_MAP_FISH_TO_RECT)
                                 : (out_fish ? VSTAB_MAP_RECT_TO_FISH : VSTAB_MAP_RECT_TO_RECT);
    }
    if (H->ow <= 0 || H->oh <= 0 || H->ow > 32767 || H->oh > 32767) return fail(VSTAB_ERR_INVALID, "vstab_create: output size out of range");
    if (cfg->smoother == VSTAB_SMOOTHER_SG) H->sg.reset(new RotationFilterSG(cfg->smooth_radius));
    if (cfg->read_ahead > 0) H->prefetch_depth = cfg->read_ahead;  // vstab_config.read_ahead: the caller's latency / throughput choice
    else if (const char *e = getenv("VSTAB_PREFETCH")) H->prefetch_depth = std::max(1, std::min(atoi(e), PREFETCH_MAX));  // (development sweeps)
    if (const char *e = getenv("VSTAB_LK_SEG_TARGET")) H->seg_target = std::max(1, std::min(atoi(e), H->seg_max));
    // queue (r + 1) + the frame whose estimate runs + the one whose results are read + the one in flight + first/last gray + 1 spare
    // + the read-ahead + the slots that wait for a shared warp event
    H->slots.resize((size_t)cfg->smooth_radius + 6 + H->prefetch_depth + vstab_handle::WARP_EVENT_STRIDE);
    for (auto &s : H->slots) {
        VSTAB_HIP_TRY(hipEventCreateWithFlags(&s.ingested, hipEventDisableTiming));
        VSTAB_HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    }
    // a frame stays in the pipeline from its pull until its warp: read-ahead + look-ahead queue + the frames in between
    H->borrow_hold = getenv("VSTAB_ALWAYS_COPY") ? (1 << 30) + 1 : cfg->smooth_radius + H->prefetch_depth + 6;
    if (cfg->tracking) VSTAB_TRY(H->tracker.init(H->w, H->h));
    *out = H.release();
    return VSTAB_OK;
}

vstab_status vstab_get_output_info(const vstab_handle *h, int *width, int *height, double K_in[9], double K_out[9]) {
    if (!h) return fail(VSTAB_ERR_INVALID, "null handle");
    if (width) *width = h->ow;
    if (height) *height = h->oh;
    if (K_in) std::memcpy(K_in, h->Kin.m, sizeof(h->Kin.m));
    if (K_out) std::memcpy(K_out, h->Kout.m, sizeof(h->Kout.m));
    return VSTAB_OK;
}

}  // extern "C"

constexpr int OUT_BGR16 = 16;        // internal: the 10-bit path's output (vstab_pull_frame_bgr16)
constexpr int OUT_P010 = 17;         // internal: the 10-bit path's frame as P010 planes (vstab_pull_frame_p010)
constexpr int OUT_P010_PLANAR = 18;  // internal: the 10-bit frame warped plane by plane (vstab_pull_frame_p010_planar)
static inline bool out_is_10bit(int f) { return f == OUT_BGR16 || f == OUT_P010 || f == OUT_P010_PLANAR; }
static inline bool out_has_chroma_plane(int f) { return f == VSTAB_OUT_NV12 || f == VSTAB_OUT_NV12_PLANAR || f == OUT_P010 || f == OUT_P010_PLANAR; }

// FrameSourceWarp::pull_frame, :452-476
static vstab_status pull_frame_impl(vstab_handle *H, int out_format, void *dst, size_t pitch_dst, void *dst_uv, size_t pitch_dst_uv) {
    if (!H || !dst || (out_has_chroma_plane(out_format) && !dst_uv)) return fail(VSTAB_ERR_INVALID, "vstab_pull_frame: null argument");
    if ((H->cfg.pixel_depth == 10) != out_is_10bit(out_format))
        return fail(VSTAB_ERR_INVALID, "vstab_pull_frame: a pixel_depth 10 handle emits through vstab_pull_frame_bgr16, an 8-bit handle through the others");
    HT t_total(HostTimers::TOTAL);
    while (H->queue.size() <= (size_t)H->cfg.smooth_radius) {  // :453
        if (!H->have_inflight && !H->have_ready && !H->have_estimating && H->prefetched.empty() && H->src_eof) {  // every frame read has been queued
            if (H->src_error) return fail(VSTAB_ERR_SOURCE, "upstream pull failed with " + std::to_string(H->src_error));
            // :456-461 pretend the camera kept its last orientation (once per call while draining)
            if (H->sg) H->sg->add(H->measured);
            break;
        }
        // 1. LK results of the frame in flight -> surviving corners
        if (H->have_inflight) VSTAB_TRY(finish_wait(H));
        // 2. the rotation estimate that was started a frame ago (it ran beside everything since) -> queue its frame; then
        //    the frame just read starts its estimate on the worker thread
        finish_estimate(H);
        post_estimate(H);
        // 3. key-frame rule + LK launch for the oldest prefetched frame, and the launches that can be enqueued ahead of it
        if (H->prefetched.empty() && !H->src_eof) {
            const vstab_status st = prefetch_next(H);
            if (st != VSTAB_OK && st != VSTAB_EOF) return st;
        }
        if (!H->prefetched.empty() && !H->have_inflight) {
            const size_t queued = H->queue.size();
            VSTAB_TRY(launch_tracking(H));
            if (H->queue.size() != queued) continue;  // (tracking off: the frame is queued at once) re-check :453 before :456
        }
        // 4. read ahead: pull + copy + pyramid of the following frames (prefetch stream)
        while ((int)H->prefetched.size() < H->prefetch_depth && !H->src_eof) {
            const vstab_status st = prefetch_next(H);
            if (st != VSTAB_OK && st != VSTAB_EOF) return st;
        }
    }
    if (H->queue.empty()) return VSTAB_EOF;  // :465-467
    const int slot = H->queue.front().first;
    const Mat3 measured = H->queue.front().second;
    H->queue.pop_front();
    Mat3 corrected, warp_R;
    {
        HostStage hs(&H->prof.host_smooth_ms);
        if (H->cfg.smoother == VSTAB_SMOOTHER_SG)
            corrected = H->sg->filter();  // :471
        else if (H->cfg.smoother == VSTAB_SMOOTHER_KALMAN)
            corrected = H->kalman.update(measured);
        else if (H->cfg.smoother == VSTAB_SMOOTHER_FIXED)
            corrected = Mat3::identity();  // hold the orientation of the first frame
        else
            corrected = measured;
        const Mat3 correction = corrected * measured.inv();  // :472
        warp_R = correction.inv();                           // :475
    }
    H->warp_log.push_back(warp_R);
    if (H->warp_log.size() > vstab_handle::LOG_KEEP) H->warp_log.pop_front(), H->warp_log_base++;
    H->prof.frames_emitted++, H->prof.warp_launches++;
    float p[17];
    map_params(H->Kin, H->Kout, warp_R, p);
    vstab_handle::Slot &S = H->slots[slot];
    // rolling shutter: the camera kept turning while the rows were read out; the last row is warped with the stabilising
    // rotation of the orientation it was exposed at (measured' = readout * measured  =>  W' = readout * W)
    float p_bottom[17];
    if (S.have_readout) map_params(H->Kin, H->Kout, S.readout * warp_R, p_bottom);
    bool cached = false;
    // (the quantised map holds no chroma positions: the plane-wise warp always evaluates its map)
    if (H->map_cache && !S.have_readout && !out_is_10bit(out_format) && out_format != VSTAB_OUT_NV12_PLANAR) {
        if (H->qmap_valid && std::memcmp(p, H->qmap_params, sizeof(p)) == 0) {
            cached = true;
        } else if (H->have_last_params && std::memcmp(p, H->last_params, sizeof(p)) == 0) {
            // second frame in a row with these parameters: write the map down now (same stream, ahead of the warp)
            VSTAB_TRY(H->qmap.ensure(vstab_quantised_map_bytes(H->ow, H->oh)));
            VSTAB_TRY(vstab_quantised_map(H->qmap.p, H->ow, H->oh, p, H->map_mode, H->stream));
            std::memcpy(H->qmap_params, p, sizeof(p));
            H->qmap_valid = cached = true;
        }
        std::memcpy(H->last_params, p, sizeof(p));
        H->have_last_params = true;
        H->warps_from_cache += cached;
    }
    HT t_warp(HostTimers::WARP);
    VSTAB_TRY(vstab_handle::wait_if_pending(H->stream, S.ingested));  // the slot was filled on the prefetch stream (long ago, as a rule)
    vstab_status st;
    {
        // the profiling events bracket the launch call and nothing else, so the interval is the kernel
        // (plus its dispatch), not host work between two API calls
        GpuStage gs(H, vstab_handle::ST_WARP);
#ifdef VSTAB_DEV
        // development builds: VSTAB_DEV_SKIP_WARP=1 launches no warp at all, so that tools/lk_timeline.py sees the tracker chain with
        // nothing but the pyramid kernels beside it (how much of an iteration is the chain, how much is contention with the warp)
        static const bool skip_warp = getenv("VSTAB_DEV_SKIP_WARP") != nullptr;
        if (skip_warp) {
            (void)take_launch_events();
            st = VSTAB_OK;
        } else
#endif
        if (out_format == OUT_BGR16)
            st = vstab_warp_p010(S.y16, S.pitch_y16, S.uv16, S.pitch_uv16, H->w, H->h, p,
                                 S.have_readout ? p_bottom + 8 : nullptr, H->map_mode, H->cfg.blend, dst, pitch_dst, H->ow, H->oh, H->stream);
        else if (out_format == OUT_P010_PLANAR)
            st = vstab_warp_p010_planar(S.y16, S.pitch_y16, S.uv16, S.pitch_uv16, H->w, H->h, p, S.have_readout ? p_bottom + 8 : nullptr, H->map_mode,
                                        H->cfg.blend, dst, pitch_dst, dst_uv, pitch_dst_uv, H->ow, H->oh, H->stream);
        else if (out_format == OUT_P010) {
            st = vstab_warp_p010_planes(S.y16, S.pitch_y16, S.uv16, S.pitch_uv16, H->w, H->h, p, S.have_readout ? p_bottom + 8 : nullptr, H->map_mode,
                                        H->cfg.blend, dst, pitch_dst, dst_uv, pitch_dst_uv, H->ow, H->oh, H->stream);
            if (st == VSTAB_ERR_UNSUPPORTED) {
                const size_t bpitch = ((size_t)H->ow * 6 + 255) & ~(size_t)255;
                st = H->bgr16_out.ensure(bpitch * H->oh);
                if (st == VSTAB_OK)
                    st = vstab_warp_p010(S.y16, S.pitch_y16, S.uv16, S.pitch_uv16, H->w, H->h, p, S.have_readout ? p_bottom + 8 : nullptr, H->map_mode,
                                         H->cfg.blend, H->bgr16_out.p, bpitch, H->ow, H->oh, H->stream);
                if (st == VSTAB_OK) st = vstab_cvt_bgr16_p010(H->bgr16_out.p, bpitch, H->ow, H->oh, dst, pitch_dst, dst_uv, pitch_dst_uv, H->stream);
            }
        }
        else if (H->cfg.interpolation == 0) {
            // (vstab_warp_nv12_nearest_ex takes the profiler's event pair like the other warp kernels; a refused request launches nothing,
            //  so the pair armed by GpuStage is taken back here instead of staying pending for somebody else's launch)
            if (out_format != VSTAB_OUT_BGR8 || S.have_readout) {
                (void)take_launch_events();
                st = fail(VSTAB_ERR_INVALID, "INTER_NEAREST emits 8-bit BGR frames without a read-out rotation");
            }
            else st = vstab_warp_nv12_nearest_ex(S.y, S.pitch_y, S.uv, S.pitch_uv, H->w, H->h, p, H->map_mode, dst, pitch_dst, H->ow, H->oh, H->stream);
        } else if (cached)
            st = vstab_warp_nv12_mapped(S.y, S.pitch_y, S.uv, S.pitch_uv, H->w, H->h, H->qmap.p, out_format, dst, pitch_dst, dst_uv, pitch_dst_uv,
                                        H->ow, H->oh, H->stream);
        else if (S.have_readout)
            st = vstab_warp_nv12_rs(S.y, S.pitch_y, S.uv, S.pitch_uv, H->w, H->h, p, p_bottom + 8, H->map_mode, out_format, dst, pitch_dst, dst_uv,
                                    pitch_dst_uv, H->ow, H->oh, H->stream);
        else
            st = vstab_warp_nv12_ex(S.y, S.pitch_y, S.uv, S.pitch_uv, H->w, H->h, p, H->map_mode, out_format, dst,
                                    pitch_dst, dst_uv, pitch_dst_uv, H->ow, H->oh, H->stream);
    }
    if (st == VSTAB_OK && H->cfg.debug && !S.feats.empty() && !out_is_10bit(out_format)) {  // (markers are drawn into 8-bit outputs)
        // where the warp sends each tracked feature: input pixel -> ray -> R^T -> output projection (the inverse of the map)
        constexpr int SETS = 16, CAP = 256;
        VSTAB_TRY(H->marker_pts.ensure(sizeof(int) * 2 * CAP * SETS));
        int *host = H->marker_pts.as<int>() + 2 * CAP * (H->marker_set % SETS);
        int *dev = static_cast<int *>(H->marker_pts.dev()) + 2 * CAP * (H->marker_set % SETS);
        H->marker_set++;
        const bool out_fish = H->map_mode == VSTAB_MAP_FISH_TO_FISH || H->map_mode == VSTAB_MAP_RECT_TO_FISH;
        int n = 0;
        for (size_t i = 0; i + 1 < S.feats.size() && n < CAP; i += 2) {
            const double a = (S.feats[i] - H->Kin(0, 2)) / H->Kin(0, 0), b = (S.feats[i + 1] - H->Kin(1, 2)) / H->Kin(1, 1);
            double rx = a, ry = b, rz = 1;
            if (H->in_fish) {
                const double th = std::hypot(a, b), sc = th > 0 ? std::sin(th) / th : 1.0;
                rx = a * sc, ry = b * sc, rz = std::cos(th);
            }
            const double ox = warp_R(0, 0) * rx + warp_R(1, 0) * ry + warp_R(2, 0) * rz, oy = warp_R(0, 1) * rx + warp_R(1, 1) * ry + warp_R(2, 1) * rz,
                         oz = warp_R(0, 2) * rx + warp_R(1, 2) * ry + warp_R(2, 2) * rz;
            if (!(oz > 0)) continue;
            double u = ox / oz, v = oy / oz;
            if (out_fish) {
                const double r = std::hypot(ox, oy), th = std::atan2(r, oz), sc = r > 0 ? th / r : 1.0;
                u = ox * sc, v = oy * sc;
            }
            host[2 * n] = (int)std::nearbyint(H->Kout(0, 2) + u * H->Kout(0, 0)), host[2 * n + 1] = (int)std::nearbyint(H->Kout(1, 2) + v * H->Kout(1, 1));
            n++;
        }
        if (out_format == VSTAB_OUT_NV12 || out_format == VSTAB_OUT_NV12_PLANAR)
            st = vstab_draw_markers(dst, pitch_dst, H->ow, H->oh, 1, dev, n, 3, 235u, H->stream);
        else
            st = vstab_draw_markers(dst, pitch_dst, H->ow, H->oh, 3, dev, n, 3, 0x0000FF00u, H->stream);
        S.feats.clear();
    }
    S.queued = false, S.freed_at = ++H->free_counter;
    if (!S.borrowed) {
        S.warp_pending = true, S.warped = -1;
        H->uncovered.push_back(slot);  // the next copy into this slot waits for an event recorded behind this warp
        if ((int)H->uncovered.size() >= vstab_handle::WARP_EVENT_STRIDE) VSTAB_TRY(H->cover_warps());
    } else if (!H->borrows.empty()) {
        for (vstab_handle::PendingBorrow &b : H->borrows)
            if (b.serial == S.ingest_serial) {
                b.warp_enqueued = true;
                if (++H->uncovered_borrows >= vstab_handle::WARP_EVENT_STRIDE) VSTAB_TRY(H->cover_warps());
                break;
            }
    }
    return st;
}

extern "C" {

vstab_status vstab_pull_frame(vstab_handle *h, void *dst, size_t pitch_dst) {
    return pull_frame_impl(h, VSTAB_OUT_BGR8, dst, pitch_dst, nullptr, 0);
}

vstab_status vstab_pull_frames(vstab_handle *h, int n, void *const *dst, const size_t *pitch_dst, int n_dst, int first, int *n_done) {
    if (n_done) *n_done = 0;
    if (!h || !dst || !pitch_dst || n < 0 || n_dst <= 0 || first < 0) return fail(VSTAB_ERR_INVALID, "vstab_pull_frames: bad argument");
    for (int i = 0; i < n; i++) {
        const int k = (int)(((long)first + i) % n_dst);
        const vstab_status st = pull_frame_impl(h, VSTAB_OUT_BGR8, dst[k], pitch_dst[k], nullptr, 0);
        if (st != VSTAB_OK) return st;
        if (n_done) *n_done = i + 1;
    }
    return VSTAB_OK;
}

vstab_status vstab_pull_frame_host(vstab_handle *h, void *dst, size_t pitch_dst) {
    if (!h || !dst || pitch_dst < (size_t)h->ow * 3) return fail(VSTAB_ERR_INVALID, "vstab_pull_frame_host: bad argument");
    const size_t dpitch = ((size_t)h->ow * 3 + 255) & ~(size_t)255;
    VSTAB_TRY(h->host_out.ensure(dpitch * h->oh));
    const vstab_status st = pull_frame_impl(h, VSTAB_OUT_BGR8, h->host_out.p, dpitch, nullptr, 0);
    if (st != VSTAB_OK) return st;
    VSTAB_HIP_TRY(hipMemcpy2DAsync(dst, pitch_dst, h->host_out.p, dpitch, (size_t)h->ow * 3, h->oh, hipMemcpyDeviceToHost, h->stream));
    VSTAB_HIP_TRY(hipStreamSynchronize(h->stream));
    return VSTAB_OK;
}

vstab_status vstab_pull_frame_bgr16(vstab_handle *h, void *dst, size_t pitch_dst) {
    return pull_frame_impl(h, OUT_BGR16, dst, pitch_dst, nullptr, 0);
}

vstab_status vstab_pull_frame_p010(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv) {
    if (!h || !dst_y || !dst_uv) return fail(VSTAB_ERR_INVALID, "vstab_pull_frame_p010: null argument");
    // the warp writes the planes itself (OUT_P010) where the frame's planes allow the tiled kernel; pull_frame_impl falls back
    // to a 16-bit BGR buffer of the handle + vstab_cvt_bgr16_p010 otherwise
    return pull_frame_impl(h, OUT_P010, dst_y, pitch_y, dst_uv, pitch_uv);
}

vstab_status vstab_pull_frame_nv12(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv) {
    return pull_frame_impl(h, VSTAB_OUT_NV12, dst_y, pitch_y, dst_uv, pitch_uv);
}

vstab_status vstab_pull_frame_nv12_planar(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv) {
    return pull_frame_impl(h, VSTAB_OUT_NV12_PLANAR, dst_y, pitch_y, dst_uv, pitch_uv);
}

vstab_status vstab_pull_frame_p010_planar(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv) {
    return pull_frame_impl(h, OUT_P010_PLANAR, dst_y, pitch_y, dst_uv, pitch_uv);
}

vstab_status vstab_peek_frame(vstab_handle *h, void *dst, size_t pitch_dst) { return vstab_pull_frame(h, dst, pitch_dst); }  // :478-480

vstab_status vstab_enable_profiling(vstab_handle *h, int enable) {
    if (!h) return fail(VSTAB_ERR_INVALID, "null handle");
    h->profiling = enable < 0 ? 0 : enable > 2 ? 2 : enable;
    return VSTAB_OK;
}

vstab_status vstab_get_profile(vstab_handle *h, vstab_profile *out) {
    if (!h || !out) return fail(VSTAB_ERR_INVALID, "vstab_get_profile: null argument");
    h->fold_pending();
    h->prof.dmabuf_imports = h->dmabufs.imports, h->prof.dmabuf_evictions = h->dmabufs.evictions, h->prof.dmabuf_cached = (long)h->dmabufs.size();
    h->prof.corner_selections_by_caller = h->tracker.selections_by_caller(), h->prof.corner_selections_by_helper = h->tracker.selections_by_helper();
    h->prof.epochs_in_turn = h->epochs_on_second_stream;
    *out = h->prof;
    return VSTAB_OK;
}

void vstab_destroy(vstab_handle *h) {
    if (!h) return;
    if (g_ht.on) {
        for (int i = 0; i < HostTimers::N; i++)
            if (g_ht.calls[i]) std::fprintf(stderr, "host %-18s %8ld calls  %8.2f us/call\n", HostTimers::name(i), g_ht.calls[i], g_ht.ms[i] / g_ht.calls[i] * 1e3);
        g_ht = HostTimers();
    }
    h->fold_pending();
    h->tracker.report_clock();
    if (debug_spec())
        std::fprintf(stderr, "frames used in place %ld, copied into the ring %ld; warps from the cached map %ld\n", h->frames_borrowed, h->frames_copied,
                     h->warps_from_cache);
    if (debug_spec())
        std::fprintf(stderr, "tracker launches: %ld segments covering %ld frames (%ld of them dropped); frames taken from a launch enqueued ahead %ld, "
                             "replaced on demand %ld, of %ld; key frames pre-launched %ld of %ld\n",
                     h->segs_launched, h->seg_frames_launched, h->seg_frames_dropped, h->chained_adopted, h->chained_discarded, h->frame_index,
                     h->key_prelaunched, h->prof.key_frames);
    if (h->estimate_posted) {
        Mat3 r;
        (void)h->worker.join(r);
    }
    h->fold_pending();  // drains the streams
    delete h;           // ~vstab_handle releases the events and the internal streams
}

int vstab_frame_log_count(const vstab_handle *h) { return h ? (int)(h->log_base + (long)h->log.size()) : 0; }

vstab_status vstab_get_frame_log(const vstab_handle *h, int index, vstab_frame_log *out) {
    if (!h || !out || index < h->log_base || index >= h->log_base + (long)h->log.size())
        return fail(VSTAB_ERR_INVALID, "vstab_get_frame_log: bad index (only the most recent 65536 entries are kept)");
    *out = h->log[(size_t)(index - h->log_base)];
    return VSTAB_OK;
}

vstab_status vstab_get_warp_rotation(const vstab_handle *h, int index, double R[9]) {
    if (!h || !R || index < h->warp_log_base || index >= h->warp_log_base + (long)h->warp_log.size())
        return fail(VSTAB_ERR_INVALID, "vstab_get_warp_rotation: bad index (only the most recent 65536 entries are kept)");
    std::memcpy(R, h->warp_log[(size_t)(index - h->warp_log_base)].m, sizeof(double) * 9);
    return VSTAB_OK;
}

// ---------------------------------------------------------------------------------------------
// ring source
// ---------------------------------------------------------------------------------------------
struct vstab_ring_source {
    std::vector<const void *> frames;
    int w, h;
    size_t pitch;
    long total, pos;
    int bit_depth = 8;
    int hold = 1 << 30;  // what the source promises: by default the caller owns the frames for the life of the source and never rewrites them
    std::vector<double> readout;  // optional: 9 doubles per ring frame (vstab_frame.readout_rotation)
};

static int ring_fill(vstab_ring_source *s, vstab_frame *out) {
    if (s->pos >= s->total) return VSTAB_EOF;
    const uint8_t *p = static_cast<const uint8_t *>(s->frames[(size_t)(s->pos % (long)s->frames.size())]);
    out->y = p, out->uv = p + s->pitch * s->h, out->pitch_y = out->pitch_uv = s->pitch;
    out->width = s->w, out->height = s->h, out->mem = 0, out->pts = s->pos;
    out->hold = s->hold;
    out->bit_depth = s->bit_depth;
    if (!s->readout.empty()) out->readout_rotation = &s->readout[9 * (size_t)(s->pos % (long)s->frames.size())];
    return 0;
}
static int ring_pull(void *user, vstab_frame *out) {
    vstab_ring_source *s = static_cast<vstab_ring_source *>(user);
    const int rc = ring_fill(s, out);
    if (rc == 0) s->pos++;
    return rc;
}
static int ring_peek(void *user, vstab_frame *out) { return ring_fill(static_cast<vstab_ring_source *>(user), out); }

void vstab_ring_source_set_hold(vstab_ring_source *s, int hold) {
    if (s) s->hold = hold < 0 ? 0 : hold;
}

vstab_status vstab_ring_source_create(const void *const *frames, int n_frames, int width, int height, size_t pitch,
                                      long total_frames, vstab_ring_source **out, vstab_source *as_source) {
    if (!frames || n_frames <= 0 || !out || !as_source || width <= 0 || height <= 0 || pitch < (size_t)width)
        return fail(VSTAB_ERR_INVALID, "vstab_ring_source_create: bad argument");
    vstab_ring_source *s = new vstab_ring_source;
    s->frames.assign(frames, frames + n_frames);
    s->w = width, s->h = height, s->pitch = pitch, s->total = total_frames, s->pos = 0;
    as_source->pull = ring_pull, as_source->peek = ring_peek, as_source->user = s;
    *out = s;
    return VSTAB_OK;
}

vstab_status vstab_ring_source_create_ex(const void *const *frames, int n_frames, int width, int height, size_t pitch, long total_frames, int bit_depth,
                                         const double *readout_rotations, vstab_ring_source **out, vstab_source *as_source) {
    if (bit_depth != 8 && bit_depth != 10 && bit_depth != 12 && bit_depth != 16) return fail(VSTAB_ERR_INVALID, "vstab_ring_source_create: bit_depth must be 8, 10, 12 or 16");
    if (pitch < (size_t)width * (bit_depth > 8 ? 2 : 1)) return fail(VSTAB_ERR_INVALID, "vstab_ring_source_create: bad argument");
    VSTAB_TRY(vstab_ring_source_create(frames, n_frames, width, height, pitch, total_frames, out, as_source));
    (*out)->bit_depth = bit_depth;
    if (readout_rotations) (*out)->readout.assign(readout_rotations, readout_rotations + 9 * (size_t)n_frames);
    return VSTAB_OK;
}

void vstab_ring_source_destroy(vstab_ring_source *s) { delete s; }

// ---------------------------------------------------------------------------------------------
// stateless tracking / motion entry points
// ---------------------------------------------------------------------------------------------
vstab_status vstab_pyr_down(const void *src, size_t pitch_src, int width, int height, void *dst, size_t pitch_dst, void *stream) {
    if (!src || !dst || width <= 0 || height <= 0 || pitch_src < (size_t)width || pitch_dst < (size_t)((width + 1) / 2))
        return fail(VSTAB_ERR_INVALID, "vstab_pyr_down: bad argument");
    return launch_pyr_down((const uint8_t *)src, pitch_src, width, height, (uint8_t *)dst, pitch_dst, static_cast<hipStream_t>(stream));
}

vstab_status vstab_pyr_down_x2(const void *src, size_t pitch_src, int width, int height, void *mid, size_t pitch_mid, void *dst, size_t pitch_dst, void *stream) {
    const int mw = (width + 1) / 2, mh = (height + 1) / 2;
    if (!src || !mid || !dst || width <= 0 || height <= 0 || pitch_src < (size_t)width || pitch_mid < (size_t)mw || pitch_dst < (size_t)((mw + 1) / 2))
        return fail(VSTAB_ERR_INVALID, "vstab_pyr_down_x2: bad argument");
    if (!pyr_down_x2_ok(width, height)) {  // tiny images: two single-level launches, the same bytes
        VSTAB_TRY(launch_pyr_down((const uint8_t *)src, pitch_src, width, height, (uint8_t *)mid, pitch_mid, static_cast<hipStream_t>(stream)));
        return launch_pyr_down((const uint8_t *)mid, pitch_mid, mw, mh, (uint8_t *)dst, pitch_dst, static_cast<hipStream_t>(stream));
    }
    return launch_pyr_down_x2((const uint8_t *)src, pitch_src, width, height, (uint8_t *)mid, pitch_mid, (uint8_t *)dst, pitch_dst, static_cast<hipStream_t>(stream));
}

vstab_status vstab_min_eig(const void *gray, size_t pitch, int width, int height, void *eig, void *stream) {
    if (!gray || !eig || width <= 0 || height <= 0 || pitch < (size_t)width) return fail(VSTAB_ERR_INVALID, "vstab_min_eig: bad argument");
    DevBuf mb;
    VSTAB_TRY(mb.ensure(16));
    VSTAB_TRY(launch_min_eig((const uint8_t *)gray, pitch, width, height, (float *)eig, mb.as<int>(), static_cast<hipStream_t>(stream)));
    VSTAB_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return VSTAB_OK;
}

vstab_status vstab_good_features_ex(const void *gray, size_t pitch, int width, int height, int max_corners, double quality,
                                    double min_distance, int detector, float *xy, int *count, int *detector_used, void *stream) {
    if (!gray || !xy || !count || width < 3 || height < 3 || pitch < (size_t)width || max_corners <= 0 ||
        (detector != VSTAB_DETECTOR_AUTO && detector != VSTAB_DETECTOR_TWO_PASS))
        return fail(VSTAB_ERR_INVALID, "vstab_good_features: bad argument");
    Tracker t;
    VSTAB_TRY(t.init(width, height));
    t.set_two_pass_detector(detector == VSTAB_DETECTOR_TWO_PASS);
    std::vector<float> out;
    VSTAB_TRY(t.good_features((const uint8_t *)gray, pitch, max_corners, quality, min_distance, out, static_cast<hipStream_t>(stream)));
    *count = (int)(out.size() / 2);
    std::memcpy(xy, out.data(), sizeof(float) * out.size());
    if (detector_used) *detector_used = (detector == VSTAB_DETECTOR_TWO_PASS || t.fused_overflows()) ? VSTAB_DETECTOR_TWO_PASS : VSTAB_DETECTOR_FUSED;
    return VSTAB_OK;
}

vstab_status vstab_good_features(const void *gray, size_t pitch, int width, int height, int max_corners, double quality,
                                 double min_distance, float *xy, int *count, void *stream) {
    return vstab_good_features_ex(gray, pitch, width, height, max_corners, quality, min_distance, VSTAB_DETECTOR_AUTO, xy, count, nullptr, stream);
}

vstab_status vstab_pyr_lk(const void *prev, size_t pitch_prev, const void *next, size_t pitch_next, int width, int height,
                          const float *prev_xy, int n, float *next_xy, unsigned char *status, void *stream) {
    if (!prev || !next || (n > 0 && (!prev_xy || !next_xy || !status)) || n < 0 || width <= 0 || height <= 0 ||
        pitch_prev < (size_t)width || pitch_next < (size_t)width)
        return fail(VSTAB_ERR_INVALID, "vstab_pyr_lk: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Tracker t;
    VSTAB_TRY(t.init(width, height));
    VSTAB_TRY(t.build_pyramid(0, (const uint8_t *)prev, pitch_prev, st));
    VSTAB_TRY(t.build_pyramid(1, (const uint8_t *)next, pitch_next, st));
    std::vector<float> p(prev_xy, prev_xy + 2 * (size_t)n), q;
    std::vector<uint8_t> s;
    VSTAB_TRY(t.track(t.pyramid(0, (const uint8_t *)prev, pitch_prev), t.pyramid(1, (const uint8_t *)next, pitch_next), p, q, s, st));
    if (n > 0) {
        std::memcpy(next_xy, q.data(), sizeof(float) * q.size());
        std::memcpy(status, s.data(), s.size());
    }
    return VSTAB_OK;
}

vstab_status vstab_estimate_rotation(const float *prev_xy, const float *cur_xy, int n, const double K_in[9], const double K_out[9],
                                     uint64_t seed, double R[9], int *inliers) {
    if ((n > 0 && (!prev_xy || !cur_xy)) || n < 0 || !K_in || !K_out || !R || !inliers)
        return fail(VSTAB_ERR_INVALID, "vstab_estimate_rotation: bad argument");
    Mat3 ki, ko, r;
    std::memcpy(ki.m, K_in, sizeof(ki.m)), std::memcpy(ko.m, K_out, sizeof(ko.m));
    Pcg32 rng(seed);
    *inliers = estimate_rotation(prev_xy, cur_xy, n, ki, ko, rng, r);
    std::memcpy(R, r.m, sizeof(r.m));
    return VSTAB_OK;
}

vstab_status vstab_sg_weights(int m, double *weights) {
    if (m < 0 || !weights) return fail(VSTAB_ERR_INVALID, "vstab_sg_weights: bad argument");
    const std::vector<double> w = sg_weights(m);
    std::memcpy(weights, w.data(), sizeof(double) * w.size());
    return VSTAB_OK;
}

struct vstab_rotation_filter {
    RotationFilterSG f;
    explicit vstab_rotation_filter(int m) : f(m) {}
};

vstab_status vstab_rotation_filter_create(int m, vstab_rotation_filter **out) {
    if (m < 0 || !out) return fail(VSTAB_ERR_INVALID, "vstab_rotation_filter_create: bad argument");
    *out = new vstab_rotation_filter(m);
    return VSTAB_OK;
}
vstab_status vstab_rotation_filter_add(vstab_rotation_filter *f, const double R[9]) {
    if (!f || !R) return fail(VSTAB_ERR_INVALID, "vstab_rotation_filter_add: null argument");
    Mat3 r;
    std::memcpy(r.m, R, sizeof(r.m));
    f->f.add(r);
    return VSTAB_OK;
}
vstab_status vstab_rotation_filter_filter(const vstab_rotation_filter *f, double R_out[9]) {
    if (!f || !R_out) return fail(VSTAB_ERR_INVALID, "vstab_rotation_filter_filter: null argument");
    const Mat3 r = f->f.filter();
    std::memcpy(R_out, r.m, sizeof(r.m));
    return VSTAB_OK;
}
void vstab_rotation_filter_destroy(vstab_rotation_filter *f) { delete f; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// Test hooks (vstabx_*: not part of the ABI of include/vstab.h, no device needed): the CPU suite -- and its sanitizer build -- drive
// the host-side bookkeeping of vstab_hostlogic.hpp with hand-made buffers (tests/test_hostlogic_cpu.py).
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" {
// Decode n hand-made tracker records as Tracker::track_wait does.  Returns the LkParse code; *n_out entries in xy / status; *next =
// the first record that was not ready (or n).
__attribute__((visibility("default"))) int vstabx_parse_records(const uint32_t *rec, int n, uint32_t seq, int expect_n, float *xy, unsigned char *status,
                                                                 int *n_out, int *next) {
    std::vector<float> pts;
    std::vector<uint8_t> st;
    int nx = 0;
    const LkParse r = lk_parse_records(rec, 0, n, seq, (size_t)expect_n, pts, st, &nx);
    for (size_t i = 0; i < st.size(); i++) xy[2 * i] = pts[2 * i], xy[2 * i + 1] = pts[2 * i + 1], status[i] = st[i];
    *n_out = (int)st.size(), *next = nx;
    return (int)r;
}
// Run a sequence of n lookups (object ids = inodes, all of one size unless sizes is given) through a DmaBufCache with fake import /
// destroy functions.  counts = {imports, evictions, mapped now, destroys seen, largest number mapped at once}; bases[i] = the base the
// i-th lookup returned (id * 4096 for the fake import: a stale mapping would show); fail_id: the import of this id fails (-1: none).
__attribute__((visibility("default"))) int vstabx_dmabuf_cache_sim(const unsigned long long *ids, const size_t *sizes, int n, int cap, long window,
                                                                    long long fail_id, long *counts, unsigned long long *bases) {
    DmaBufCache<unsigned long long> cache;
    cache.cap = cap;
    long destroys = 0, peak = 0;
    std::vector<unsigned long long> live;
    int failures = 0;
    for (int i = 0; i < n; i++) {
        uint8_t *base = nullptr;
        const unsigned long long id = ids[i];
        const bool ok = cache.lookup(id, sizes ? sizes[i] : 4096, window,
                                     [&](unsigned long long &h, uint8_t *&b) {
                                         if ((long long)id == fail_id) return false;
                                         h = id, b = reinterpret_cast<uint8_t *>(static_cast<uintptr_t>(id * 4096));
                                         live.push_back(id);
                                         return true;
                                     },
                                     [&](unsigned long long &h) {
                                         destroys++;
                                         for (size_t k = 0; k < live.size(); k++)
                                             if (live[k] == h) {
                                                 live.erase(live.begin() + (long)k);
                                                 break;
                                             }
                                     },
                                     base);
        failures += !ok;
        bases[i] = ok ? static_cast<unsigned long long>(reinterpret_cast<uintptr_t>(base)) : ~0ull;
        peak = std::max<long>(peak, (long)cache.size());
    }
    counts[0] = cache.imports, counts[1] = cache.evictions, counts[2] = (long)cache.size(), counts[3] = destroys, counts[4] = peak;
    cache.clear([&](unsigned long long &) { destroys++; });
    counts[5] = destroys;
    return failures;
}
}  // extern "C"
