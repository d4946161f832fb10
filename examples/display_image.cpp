// examples/display_image.cpp -- the role of the reference's DisplayImage.cpp:21-75 main loop on the
// vstab C++ adapter: build the source chain, pull frames until EOF (thrown int), report fps in the
// format of the reference's Profiler (Profiler.cpp:25-34).  The decode chain upstream of
// FrameSourceWarp (VAAPI/OpenCL, out of scope) is replaced by a synthetic NV12 source.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/display_image.cpp -Lvideo-annotator_amd/lib -lvstab -o display_image
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vstab_frame_source.hpp"

class SyntheticSource : public vstab::NV12FrameSource {  // stands in for FrameSourceFfmpegOpenCl
  public:
    SyntheticSource(int w, int h, int n) : w_(w), h_(h), left_(n) {
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
        return f;
    }
    vstab::NV12Frame pull_frame() override {
        vstab::NV12Frame f = peek_frame();
        idx_++, left_--;
        return f;
    }

  private:
    int w_, h_, left_;
    size_t idx_ = 0;
    std::vector<void *> frames_;
};

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 100;
    auto source = std::make_shared<SyntheticSource>(1920, 1440, n);
    // DisplayImage.cpp:55: FrameSourceWarp(ffmpeg_source, GOPRO_H4B_WIDE43_MEASURED, 0.5, false, 1.0, 30)
    vstab::FrameSourceWarp warped(source, VSTAB_GOPRO_H4B_WIDE43_MEASURED, 0.5, false, 1.0, 30);
    void *out = nullptr;
    const size_t pitch = (size_t)warped.output_width() * 3;
    if (hipMalloc(&out, pitch * warped.output_height()) != hipSuccess) return 1;
    warped.set_output(out, pitch);
    int frames = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
        try {
            warped.pull_frame();
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
