// examples/display_image.cpp -- the role of the reference's DisplayImage.cpp:21-75 main loop on the
// vstab C++ adapter: build the source chain, pull frames until EOF (thrown int), report fps in the
// format of the reference's Profiler (Profiler.cpp:25-34).  The decode chain upstream of
// FrameSourceWarp (VAAPI/OpenCL, out of scope) is replaced by a synthetic NV12 source.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/display_image.cpp -Lvideo-annotator_amd/lib -lvstab -o display_image
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "vstab_frame_source.hpp"

class SyntheticSource : public vstab::NV12FrameSource {  // stands in for FrameSourceFfmpegOpenCl
  public:
    // gyro: also stand in for the GPMF "GYRO" stream the reference never got to read (gpmf.cpp:5-11): 3.2 kHz samples of a
    // hand-held shake, delivered as GPMF packets, parsed, and integrated per frame into the rotation since the previous frame and the
    // rotation during read-out
    SyntheticSource(int w, int h, int n, bool gyro = false) : w_(w), h_(h), left_(n), gyro_(gyro) {
        if (gyro) {
            // The camera's metadata track as the demuxer would hand it over (AvFrameSourceFileVaapi.cpp:121-123: the GPMF packets the
            // reference leaves as a TODO): one packet per second, DEVC > STRM > {SCAL, GYRO}, 3200 int16 triples (Z, X, Y) scaled by 1000,
            // everything big-endian.  vstab_gpmf_parse_gyro turns each packet into GyroFrame records (gpmf.cpp:95-101).
            const int hz = 3200;
            const double fps = 30.0, scal = 1000.0;
            const int packets = (int)((n + 2) / fps) + 1;
            for (int pk = 0; pk < packets; pk++) {
                std::vector<unsigned char> gy;
                auto be16 = [&gy](int v) { gy.push_back((unsigned char)((v >> 8) & 255)), gy.push_back((unsigned char)(v & 255)); };
                for (int i = 0; i < hz; i++) {
                    const double t = pk + (i + 0.5) / hz;
                    be16((int)std::lround(scal * 0.5 * std::sin(23 * t + 1.1))), be16((int)std::lround(scal * 0.9 * std::sin(40 * t))), be16((int)std::lround(scal * 0.7 * std::cos(31 * t + 0.3)));
                }
                auto item = [](const char *key, unsigned char type, int size, int repeat, const std::vector<unsigned char> &data) {
                    std::vector<unsigned char> v(key, key + 4);
                    v.push_back(type), v.push_back((unsigned char)size), v.push_back((unsigned char)(repeat >> 8)), v.push_back((unsigned char)(repeat & 255));
                    v.insert(v.end(), data.begin(), data.end());
                    while (v.size() % 4) v.push_back(0);
                    return v;
                };
                std::vector<unsigned char> strm = item("SCAL", 's', 2, 1, {(unsigned char)((int)scal >> 8), (unsigned char)((int)scal & 255)});
                const std::vector<unsigned char> g = item("GYRO", 's', 6, hz, gy);
                strm.insert(strm.end(), g.begin(), g.end());
                const std::vector<unsigned char> devc = item("DEVC", 0, 4, (int)(item("STRM", 0, 4, (int)strm.size() / 4, strm).size() / 4), item("STRM", 0, 4, (int)strm.size() / 4, strm));
                std::vector<vstab_gyro_sample> got(hz);
                int found = 0;
                if (vstab_gpmf_parse_gyro(devc.data(), devc.size(), (double)pk, 1.0, got.data(), hz, &found) != VSTAB_OK || found != hz) throw -1;
                samples_.insert(samples_.end(), got.begin(), got.end());
            }
        }
        std::vector<unsigned char> host((size_t)w * h * 3 / 2);
        for (int f = 0; f < 4; f++) {
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) host[(size_t)y * w + x] = (unsigned char)(((x / 16 + y / 16 + f) & 1) ? 200 : 60);
            for (size_t i = (size_t)w * h; i < host.size(); i++) host[i] = 128;
            void *d = nullptr;
            if (hipMalloc(&d, host.size()) != hipSuccess) throw -1;
            (void)hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
            frames_.push_back(d);
        }
    }
    ~SyntheticSource() override {
        for (void *d : frames_) (void)hipFree(d);
    }
    vstab::NV12Frame peek_frame() override {
        if (left_ <= 0) throw (int)EOF;
        const unsigned char *p = static_cast<const unsigned char *>(frames_[idx_ % frames_.size()]);
        // hold: these frames live as long as the source, so the library uses them in place (no copy into its ring)
        vstab::NV12Frame f{p, p + (size_t)w_ * h_, (size_t)w_, (size_t)w_, w_, h_};
        f.hold = 1 << 20;
        if (gyro_) {
            // frame k: first row exposed at k / 30 s, last row 8 ms later; a body-rate gyro -> rate_scale -1
            const double t_first = idx_ / 30.0, t_prev = idx_ ? (idx_ - 1) / 30.0 : t_first;
            if (vstab_gyro_integrate(samples_.data(), (int)samples_.size(), -1.0, t_prev, t_first, t_first + 0.008, delta_, readout_) != VSTAB_OK) throw -1;
            f.delta_rotation = delta_, f.readout_rotation = readout_;
        }
        return f;
    }
    vstab::NV12Frame pull_frame() override {
        vstab::NV12Frame f = peek_frame();
        idx_++, left_--;
        return f;
    }

  private:
    int w_, h_, left_;
    bool gyro_;
    size_t idx_ = 0;
    std::vector<void *> frames_;
    std::vector<vstab_gyro_sample> samples_;
    double delta_[9], readout_[9];
};

int run(vstab::FrameSourceWarp &warped, int n, bool nv12_planes = false);

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 100;
    const bool gyro = argc > 2 && std::string(argv[2]) == "gyro";
    const bool nv12 = argc > 2 && std::string(argv[2]) == "nv12";  // frames leave as NV12 planes, remapped plane-wise: what an encoder takes (render.ts:275-281)
    auto source = std::make_shared<SyntheticSource>(1920, 1440, n, gyro);
    if (gyro) {  // motion from the (synthetic) gyro stream instead of optical flow
        vstab::FrameSourceWarp warped(source, VSTAB_GOPRO_H4B_WIDE43_MEASURED, vstab::FrameSourceWarp::SensorMotion{}, 0.5, false, 1.0, 30);
        return run(warped, n);
    }
    // DisplayImage.cpp:55: FrameSourceWarp(ffmpeg_source, GOPRO_H4B_WIDE43_MEASURED, 0.5, false, 1.0, 30)
    vstab::FrameSourceWarp warped(source, VSTAB_GOPRO_H4B_WIDE43_MEASURED, 0.5, false, 1.0, 30);
    return run(warped, n, nv12);
}

int run(vstab::FrameSourceWarp &warped, int n, bool nv12_planes) {
    void *out = nullptr;
    // BGR: width * 3 bytes per row; NV12: a luma plane of `width` bytes per row and, behind it, a chroma plane of 2 * ceil(width / 2)
    const int ow = warped.output_width(), oh = warped.output_height();
    const size_t pitch = nv12_planes ? (((size_t)ow + 1) & ~(size_t)1) : (size_t)ow * 3;
    const size_t rows = nv12_planes ? (size_t)oh + ((size_t)oh + 1) / 2 : (size_t)oh;
    if (hipMalloc(&out, pitch * rows) != hipSuccess) return 1;
    warped.set_output(out, pitch);
    int frames = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
        try {
            if (nv12_planes) warped.pull_frame_nv12(out, pitch, static_cast<uint8_t *>(out) + pitch * oh, pitch, /*plane_wise=*/true);
            else warped.pull_frame();
            frames++;
        } catch (int err) {
            if (err == EOF) break;
            throw;
        }
    }
    (void)hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "opencv-warped: %.3f ms/frame (%.1f fps). %d frames of %dx%d\n", ms / frames, frames * 1e3 / ms, frames,
                 warped.output_width(), warped.output_height());
    (void)hipFree(out);
    return frames == n - 1 ? 0 : 2;  // first input frame is never emitted (FrameSourceWarp.cpp:403-407)
}
