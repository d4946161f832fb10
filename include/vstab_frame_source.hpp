// vstab_frame_source.hpp -- header-only C++ mirror of the reference's FrameSource interface
// (opencv/FrameSource.hpp:9-24) and FrameSourceWarp class (opencv/FrameSourceWarp.hpp:40-96) on top
// of the C ABI in vstab.h: same class names, constructor arguments, pull/peek semantics and error
// behaviour (errors and end-of-stream travel as thrown `int`, EOF == -1).
//
// Frames are plain descriptors of DEVICE memory instead of cv::UMat, so this header needs neither
// OpenCV nor OpenCL.  A maintainer wiring it into the reference's DisplayImage.cpp wraps
// cv::UMat <-> vstab::NV12Frame / BGRFrame (INTEGRATION.md shows the ten lines).
#pragma once
#include <cstdio>
#include <memory>
#include <stdexcept>

#include "vstab.h"

namespace vstab {

// What FrameSourceFfmpegOpenCl yields: one packed or two-plane NV12 frame in device memory.
struct NV12Frame {
    const void *y = nullptr, *uv = nullptr;
    size_t pitch_y = 0, pitch_uv = 0;
    int width = 0, height = 0;
    bool host = false;
    const double *delta_rotation = nullptr;  // optional external (gyro) rotation since the previous frame, 3x3 row-major
    const double *readout_rotation = nullptr;  // optional: rotation during this frame's read-out (rolling shutter), from the same sensor
    int bit_depth = 8;  // 10 / 12 / 16: P010-style 16-bit samples (device memory), narrowed on ingest
    int hold = 0;  // vstab_frame.hold: further pulls this frame's memory stays valid for (0 = until the next pull)
};

// What FrameSourceWarp yields: BGR8 in device memory (cv::UMat CV_8UC3 in the reference).
struct BGRFrame {
    void *data = nullptr;
    size_t pitch = 0;
    int width = 0, height = 0;
};

// opencv/FrameSource.hpp:9-24 -- "Raises an exception if no frames are ready"
template <typename FrameT>
class FrameSourceT {
  public:
    virtual FrameT pull_frame() = 0;
    virtual FrameT peek_frame() = 0;
    virtual ~FrameSourceT() = default;
};
using NV12FrameSource = FrameSourceT<NV12Frame>;
using FrameSource = FrameSourceT<BGRFrame>;

using CameraPreset = vstab_camera_preset;  // same enumerators with a VSTAB_ prefix

// NV12 output for the encoder hand-off: two planes in device memory.
struct NV12Out {
    void *y = nullptr, *uv = nullptr;
    size_t pitch_y = 0, pitch_uv = 0;
    int width = 0, height = 0;
};

// The option surface of the CLI's libdewobble filter (src/render.ts:611-617,669-683,711-717), field for field.
struct DewobbleOptions {
    enum Projection { rect = VSTAB_PROJ_RECT, fish = VSTAB_PROJ_FISH };
    enum Stab { none, fixed, sg };
    Projection in_p = fish, out_p = rect;
    double in_dfov = 0, out_dfov = 0;  // degrees; out_dfov 0 = in_dfov
    int out_w = 0, out_h = 0;          // 0 = input size
    double out_fx = -1, out_fy = -1;   // "focal point" = principal point; negative = out_w/2, out_h/2
    Stab stab = sg;
    int stab_r = 30;
    bool debug = false;
};

// opencv/FrameSourceWarp.hpp:40-96.  `out` is caller-provided device storage for the frame returned
// by pull_frame (the reference allocates a fresh UMat per frame; ownership rules: INTEGRATION.md).
class FrameSourceWarp : public FrameSource {
  public:
    FrameSourceWarp(std::shared_ptr<NV12FrameSource> source, CameraPreset input_camera, double scale = 1,
                    bool crop_borders = false, double zoom = 1, int smooth_radius = 30,
                    int interpolation = 1 /* cv::INTER_LINEAR */, void *hip_stream = nullptr)
        : m_source(std::move(source)) {
        vstab_config cfg;
        vstab_config_default(&cfg);
        cfg.preset = input_camera, cfg.scale = scale, cfg.crop_borders = crop_borders, cfg.zoom = zoom;
        cfg.smooth_radius = smooth_radius, cfg.interpolation = interpolation, cfg.stream = hip_stream;
        init(cfg);
    }
    // Motion from an external sensor instead of optical flow -- the gyro path the reference stubs (gpmf.cpp:5-11,
    // AvFrameSourceFileVaapi.cpp:121-123): every frame of `source` carries delta_rotation (and, for a rolling shutter,
    // readout_rotation), e.g. from vstab_gyro_integrate; smoothing and warp are the reference's (not a reference constructor).
    struct SensorMotion {};
    FrameSourceWarp(std::shared_ptr<NV12FrameSource> source, CameraPreset input_camera, SensorMotion, double scale = 1,
                    bool crop_borders = false, double zoom = 1, int smooth_radius = 30, void *hip_stream = nullptr)
        : m_source(std::move(source)) {
        vstab_config cfg;
        vstab_config_default(&cfg);
        cfg.preset = input_camera, cfg.scale = scale, cfg.crop_borders = crop_borders, cfg.zoom = zoom;
        cfg.smooth_radius = smooth_radius, cfg.stream = hip_stream, cfg.tracking = 0;
        init(cfg);
    }
    // The same object configured the way the CLI configures libdewobble (not a reference constructor).
    FrameSourceWarp(std::shared_ptr<NV12FrameSource> source, const DewobbleOptions &o, void *hip_stream = nullptr)
        : m_source(std::move(source)) {
        vstab_config cfg;
        vstab_config_default(&cfg);
        cfg.lens_mode = 1, cfg.in_projection = o.in_p, cfg.out_projection = o.out_p, cfg.in_dfov = o.in_dfov, cfg.out_dfov = o.out_dfov;
        cfg.out_width = o.out_w, cfg.out_height = o.out_h, cfg.out_cx = o.out_fx, cfg.out_cy = o.out_fy;
        cfg.tracking = o.stab != DewobbleOptions::none;
        cfg.smoother = o.stab == DewobbleOptions::fixed ? VSTAB_SMOOTHER_FIXED : o.stab == DewobbleOptions::sg ? VSTAB_SMOOTHER_SG : VSTAB_SMOOTHER_NONE;
        cfg.smooth_radius = o.stab == DewobbleOptions::sg ? o.stab_r : 0, cfg.stream = hip_stream, cfg.debug = o.debug;
        init(cfg);
    }
    ~FrameSourceWarp() override { vstab_destroy(m_handle); }
    FrameSourceWarp(const FrameSourceWarp &) = delete;
    FrameSourceWarp &operator=(const FrameSourceWarp &) = delete;

  private:
    void init(const vstab_config &cfg) {
        vstab_source src{&FrameSourceWarp::pull_cb, &FrameSourceWarp::peek_cb, this};
        const vstab_status st = vstab_create(&cfg, &src, &m_handle);
        if (st != VSTAB_OK) {
            std::fprintf(stderr, "FrameSourceWarp: %s\n", vstab_last_error());
            throw m_pending_error ? m_pending_error : (int)st;  // the reference rethrows the upstream int (:462)
        }
        vstab_get_output_info(m_handle, &m_out_w, &m_out_h, nullptr, nullptr);
    }

  public:
    int output_width() const { return m_out_w; }
    int output_height() const { return m_out_h; }
    // storage for the next returned frame: width*3 <= pitch, output_height() rows, device memory
    void set_output(void *device_bgr, size_t pitch) { m_out = device_bgr, m_pitch = pitch; }

    BGRFrame pull_frame() override {  // FrameSourceWarp.cpp:452-476
        if (!m_out) throw -1;
        const vstab_status st = vstab_pull_frame(m_handle, m_out, m_pitch);
        if (st == VSTAB_EOF) throw (int)EOF;                           // :466
        if (st == VSTAB_ERR_SOURCE && m_pending_error) throw m_pending_error;  // :462 `throw err`
        if (st != VSTAB_OK) {
            std::fprintf(stderr, "FrameSourceWarp: %s\n", vstab_last_error());
            throw (int)st;
        }
        return BGRFrame{m_out, m_pitch, m_out_w, m_out_h};
    }
    BGRFrame peek_frame() override { return pull_frame(); }  // :478-480 (destructive in the reference too)

    // the next frame in HOST memory (e.g. the data of a cv::Mat for imshow, DisplayImage.cpp:63-65)
    BGRFrame pull_frame_host(void *host_bgr, size_t pitch) {
        const vstab_status st = vstab_pull_frame_host(m_handle, host_bgr, pitch);
        if (st == VSTAB_EOF) throw (int)EOF;
        if (st == VSTAB_ERR_SOURCE && m_pending_error) throw m_pending_error;
        if (st != VSTAB_OK) {
            std::fprintf(stderr, "FrameSourceWarp: %s\n", vstab_last_error());
            throw (int)st;
        }
        return BGRFrame{host_bgr, pitch, m_out_w, m_out_h};
    }

    // the next frame as NV12 (planes provided by the caller: width bytes per luma row, 2*ceil(width/2) per chroma row).
    // plane_wise = true: the luma and chroma planes remapped as they are (vstab_pull_frame_nv12_planar: no colour conversion at all -- the
    // encoder hand-off of render.ts:606-607,664-665,688 and the fastest output); false: the NV12 conversion of the BGR frame
    NV12Out pull_frame_nv12(void *device_y, size_t pitch_y, void *device_uv, size_t pitch_uv, bool plane_wise = false) {
        const vstab_status st = plane_wise ? vstab_pull_frame_nv12_planar(m_handle, device_y, pitch_y, device_uv, pitch_uv)
                                           : vstab_pull_frame_nv12(m_handle, device_y, pitch_y, device_uv, pitch_uv);
        if (st == VSTAB_EOF) throw (int)EOF;
        if (st == VSTAB_ERR_SOURCE && m_pending_error) throw m_pending_error;
        if (st != VSTAB_OK) {
            std::fprintf(stderr, "FrameSourceWarp: %s\n", vstab_last_error());
            throw (int)st;
        }
        return NV12Out{device_y, device_uv, pitch_y, pitch_uv, m_out_w, m_out_h};
    }

  private:
    static int fill(FrameSourceWarp *self, vstab_frame *out, bool advance) {
        try {
            const NV12Frame f = advance ? self->m_source->pull_frame() : self->m_source->peek_frame();
            out->y = f.y, out->uv = f.uv, out->pitch_y = f.pitch_y, out->pitch_uv = f.pitch_uv;
            out->width = f.width, out->height = f.height, out->mem = f.host ? 1 : 0, out->pts = 0, out->delta_rotation = f.delta_rotation, out->bit_depth = f.bit_depth, out->hold = f.hold;
            out->readout_rotation = f.readout_rotation;
            return 0;
        } catch (int err) {  // upstream errors are thrown ints (AvFrameSourceFileVaapi.cpp:141)
            if (err != EOF) self->m_pending_error = err;
            return err == EOF ? (int)VSTAB_EOF : (err ? err : 1);
        }
    }
    static int pull_cb(void *user, vstab_frame *out) { return fill(static_cast<FrameSourceWarp *>(user), out, true); }
    static int peek_cb(void *user, vstab_frame *out) { return fill(static_cast<FrameSourceWarp *>(user), out, false); }

    std::shared_ptr<NV12FrameSource> m_source;
    vstab_handle *m_handle = nullptr;
    void *m_out = nullptr;
    size_t m_pitch = 0;
    int m_out_w = 0, m_out_h = 0, m_pending_error = 0;
};

}  // namespace vstab
