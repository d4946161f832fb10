/*
 * vstab.h -- C ABI of the MI355X-native stabilisation / fisheye-undistort hot path.
 *
 * Drop-in boundary for the reference's FrameSource -> FrameSourceWarp stage
 * (reference paths are relative to /root/reference/opencv/).  Plain pointers and sizes only;
 * no C++, OpenCV, HIP or torch types cross this interface.  `stream` arguments are a
 * hipStream_t passed as void* (NULL = the default stream); every device pointer must belong
 * to the device that is current on the calling thread.  Stateless entry points enqueue work
 * on `stream` and return without synchronising unless they say otherwise.
 *
 * Error model: the reference throws `int` (EOF == -1 for end of stream,
 * FrameSourceWarp.cpp:466; other values for failures, :181,:303).  Nothing is thrown across
 * this ABI; the same information travels as vstab_status return codes and
 * vstab_last_error().  include/vstab_frame_source.hpp re-throws them as `int` so C++ callers
 * see the reference's behaviour.
 */
#ifndef VSTAB_H_
#define VSTAB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAB_API __attribute__((visibility("default")))

typedef enum vstab_status {
    VSTAB_OK = 0,
    VSTAB_EOF = -1,          /* == EOF thrown at FrameSourceWarp.cpp:466 */
    VSTAB_ERR_INVALID = -2,  /* bad argument / unsupported geometry */
    VSTAB_ERR_DEVICE = -3,   /* HIP runtime error (details in vstab_last_error) */
    VSTAB_ERR_NOMEM = -4,
    VSTAB_ERR_SOURCE = -5    /* upstream callback failed with a code other than EOF */
} vstab_status;

/* CameraPreset, FrameSourceWarp.hpp:14-21 (same order, same values) */
typedef enum vstab_camera_preset {
    VSTAB_GOPRO_H4B_WIDE43_PUBLISHED = 0,
    VSTAB_GOPRO_H4B_WIDE43_MEASURED = 1,
    VSTAB_GOPRO_H4B_WIDE43_MEASURED_STABILISATION = 2,
    VSTAB_GOPRO_H4B_WIDE169_PUBLISHED = 3,
    VSTAB_GOPRO_H4B_WIDE169_MEASURED = 4,
    VSTAB_GOPRO_H4B_WIDE169_MEASURED_STABILISATION = 5
} vstab_camera_preset;

/* Thread-local text of the last failure of any call on this thread. */
VSTAB_API const char *vstab_last_error(void);
/* "vstab <version> gfx950" -- also proves the HIP code object is embedded. */
VSTAB_API const char *vstab_version(void);
/* Number of HIP devices visible, or a negative vstab_status. */
VSTAB_API int vstab_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Cameras (host, fp64).  Replaces get_preset_camera / get_output_camera,
 * FrameSourceWarp.cpp:27-86 and :88-165, including their integer quirks.
 * Matrices are row-major 3x3 doubles.
 * ------------------------------------------------------------------------------------------ */
VSTAB_API vstab_status vstab_get_preset_camera(int preset, int width, int height, double K[9]);
VSTAB_API vstab_status vstab_get_output_camera(const double K_in[9], int width, int height,
                                               double scale, int crop_borders, double zoom,
                                               double K_out[9], int *out_width, int *out_height);
/* cv::fisheye::undistortPoints with zero distortion (calls at :93, :322, :333).
 * R and P may be NULL (identity).  pts/out are n (x,y) double pairs. */
VSTAB_API vstab_status vstab_fisheye_undistort_points(const double *pts, int n, const double K[9],
                                                      const double *R, const double *P,
                                                      double *out);
/* The 17 cl_float kernel arguments of createMap in argument order (:283-299):
 * src cx,cy,fx,fy ; out cx,cy,fx,fy ; rot00..rot22 (double -> float casts). */
VSTAB_API void vstab_map_params(const double K_in[9], const double K_out[9], const double R[9],
                                float params[17]);

/* ------------------------------------------------------------------------------------------
 * Stateless device operators (one HIP kernel each; all pointers are DEVICE pointers).
 * ------------------------------------------------------------------------------------------ */

/* Replaces convert_ocl_images_to_nv12_umat, FrameSourceFfmpegOpenCl.cpp:12-93: packs a pitched
 * luma plane (w x h) and a pitched interleaved chroma plane (w/2 x h/2 texels of 2 bytes) into
 * one contiguous (h*3/2) x w buffer.  w and h must be even ("Mismatched image dimensions"). */
VSTAB_API vstab_status vstab_pack_nv12(const void *y, size_t pitch_y, const void *uv,
                                       size_t pitch_uv, int width, int height, void *dst_nv12,
                                       void *stream);

/* Replaces cvtColor(COLOR_YUV2BGR_NV12), FrameSourceWarp.cpp:401.  dst is BGR8, pitch_dst bytes
 * per row (>= 3*width). */
VSTAB_API vstab_status vstab_cvt_nv12_bgr(const void *y, size_t pitch_y, const void *uv,
                                          size_t pitch_uv, int width, int height, void *dst_bgr,
                                          size_t pitch_dst, void *stream);

/* Replaces the createMap OpenCL kernel, createMap.cl:1-51 + launch at FrameSourceWarp.cpp:275-304.
 * map_x / map_y: float planes, pitch in bytes.  cols, rows <= 32767 (createMap.cl:10-11 uses
 * short indices). */
VSTAB_API vstab_status vstab_create_map(void *map_x, size_t pitch_x, void *map_y, size_t pitch_y,
                                        int cols, int rows, const float params[17], void *stream);

/* Replaces cv::remap(INTER_LINEAR, BORDER_CONSTANT 0), FrameSourceWarp.cpp:306-312, for an
 * 8-bit source of `channels` (1 or 3) interleaved channels. */
VSTAB_API vstab_status vstab_remap_bilinear(const void *src, size_t pitch_src, int src_width,
                                            int src_height, int channels, const void *map_x,
                                            size_t pitch_x, const void *map_y, size_t pitch_y,
                                            void *dst, size_t pitch_dst, int dst_width,
                                            int dst_height, void *stream);

/* The fused hot kernel: cvtColor (:401) + createMap (createMap.cl) + remap (:306-312) in one
 * pass, NV12 in -> BGR8 out, map never written to memory.  Bit-identical to running the three
 * operators above in sequence. */
VSTAB_API vstab_status vstab_warp_nv12_bgr(const void *y, size_t pitch_y, const void *uv,
                                           size_t pitch_uv, int src_width, int src_height,
                                           const float params[17], void *dst_bgr, size_t pitch_dst,
                                           int dst_width, int dst_height, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VSTAB_H_ */
