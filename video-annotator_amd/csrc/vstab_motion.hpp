// vstab_motion.hpp -- host-side camera-motion estimation and trajectory smoothing (fp64).
#pragma once
#include <cstdint>
#include <vector>

#include "vstab_geometry.hpp"

namespace vstab {

// PCG32 (O'Neill): the seeded replacement for the reference's un-seeded libc rand()
// (FrameSourceWarp.cpp:345, SURVEY.md F6 / Appendix C "FIX").
struct Pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    explicit Pcg32(uint64_t seed = 42, uint64_t seq = 54) {
        state = 0, inc = (seq << 1u) | 1u;
        next();
        state += seed;
        next();
    }
    uint32_t next() {
        const uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
    double uniform() { return next() * (1.0 / 4294967295.0); }  // [0,1], like rand()*1./RAND_MAX
    uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)next() * n) >> 32); }
};

Mat3 rodrigues(const double rvec[3]);
void rodrigues_inv(const Mat3 &R, double rvec[3]);

// guess_camera_rotation, FrameSourceWarp.cpp:316-375.  prev/cur: n (x,y) float pairs in input
// (fisheye) pixels.  Returns the number of RANSAC inliers; R = rotation since the last frame.
// in_fish = false: pinhole input lens (libdewobble in_p=rect); the reference only has fisheye input.
int estimate_rotation(const float *prev, const float *cur, int n, const Mat3 &Kin, const Mat3 &Kout, Pcg32 &rng,
                      Mat3 &R, bool in_fish = true);

// Savitzky-Golay weights for (m, t=0, n=2, s=0) -- gram_sg::SavitzkyGolayFilterConfig(r,0,2,0),
// FrameSourceWarp.cpp:212.
std::vector<double> sg_weights(int m);

// polar factor U*V^T of a 3x3 matrix (what RotationFilter::filter() returns after the convolution)
Mat3 polar_orthogonal(const Mat3 &M);

// gram_sg::RotationFilter: ring of 2m+1 matrices initialised to zero
class RotationFilterSG {
  public:
    explicit RotationFilterSG(int m);
    void add(const Mat3 &R);
    Mat3 filter() const;

  private:
    int m_;
    std::vector<double> w_;
    std::vector<Mat3> ring_;
    size_t head_ = 0;  // index of the oldest entry
};

// Optional alternative smoother named by north_star: three scalar constant-velocity Kalman
// filters over the rotation vector, constants of init_filter (FrameSourceWarp.cpp:167-175 ==
// kalman/kalman.cpp:43-48): F = [[1,1],[0,1]], H = [1 0], Q = 1e-5 I, R = 1e-1, P0 = I.
// The reference never calls init_filter, so this mode has no reference output.
class RotationFilterKalman {
  public:
    RotationFilterKalman();
    Mat3 update(const Mat3 &measured);

  private:
    struct Axis {
        double x[2] = {0, 0}, P[4] = {1, 0, 0, 1};
        double step(double z);
    } ax_[3];
};

}  // namespace vstab
